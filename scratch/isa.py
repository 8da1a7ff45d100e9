import re, sys, subprocess, collections
src=sys.argv[1]; kern=sys.argv[2]
subprocess.run(f"/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fno-fast-math -Iinclude -Isyzygy_amd/csrc -S --cuda-device-only {src} -o /tmp/k.s 2>/dev/null", shell=True, check=True)
s=open('/tmp/k.s').read()
m=re.search(r'^(_ZN3szg\d+'+kern+r'\S*):[^\n]*\n(.*?)\n\s*s_endpgm', s, re.S|re.M)
lines=m.group(2).split('\n')
def isinstr(l):
    l=l.strip(); return bool(l) and not l.startswith(('.',';','//')) and not l.endswith(':') and not re.match(r'^\.?L?BB\d+_\d+:',l)
total=sum(isinstr(l) for l in lines)
print(kern,'total static instrs',total)
# loops: find "Loop Header" labels and the branch back to them
lab={}
for i,l in enumerate(lines):
    mm=re.match(r'^(\.LBB\d+_\d+):\s*(;.*)?',l)
    if mm: lab[mm.group(1)]=(i,mm.group(2) or '')
for i,l in enumerate(lines):
    mm=re.search(r's_c?branch\S*\s+(\.LBB\d+_\d+)',l)
    if mm and mm.group(1) in lab and lab[mm.group(1)][0]<i:
        j=lab[mm.group(1)][0]
        blk=[x.strip() for x in lines[j:i+1] if isinstr(x)]
        c=collections.Counter(x.split()[0] for x in blk)
        grp=collections.Counter()
        for op,n in c.items():
            if op.startswith('v_div') or op in('v_rcp_f32',): grp['div*']+=n
            elif op.startswith(('v_sqrt','v_rsq')): grp['sqrt*']+=n
            elif op.startswith('v_fma') or op.startswith('v_pk_fma'): grp['fma']+=n
            elif op.startswith(('v_mul','v_pk_mul')): grp['mul']+=n
            elif op.startswith(('v_add','v_sub','v_pk_add')): grp['add']+=n
            elif op.startswith(('global_','buffer_','flat_','scratch_')): grp['vmem']+=n
            elif op.startswith('ds_'): grp['lds']+=n
            elif op.startswith('s_'): grp['salu']+=n
            elif op.startswith('v_cndmask'): grp['cndmask']+=n
            elif op.startswith('v_cmp'): grp['cmp']+=n
            else: grp['other_v']+=n
        print(f' loop {mm.group(1)} {lab[mm.group(1)][1][:40]}: {len(blk)} instrs', dict(grp))
