"""Round 3: the contraction rule measured class by class (include/szg/contraction.h).

    python tools/contraction_classes.py build [mask ...]   # here (no GPU): variants of libszg_hip.so with -DSZG_CONTRACT=<mask>
    python tools/contraction_classes.py run OUTDIR [mask ...]   # on the GPU box: distance from the reference-SPIR-V vectors + times
    python tools/contraction_classes.py table OUTDIR       # the markdown table kept under profiles/

Without masks: every class alone (its distance), everything but that class (what un-fusing it costs), all and none.
A variant is the product's own sources compiled with another mask; nothing else differs. Variants live in
syzygy_amd/csrc/variants/ (git-ignored .so files; they travel to the GPU box with the snapshot)."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "syzygy_amd", "csrc")
VAR = os.path.join(CSRC, "variants")


def classes():
    text = open(os.path.join(ROOT, "include", "szg", "contraction.h")).read()
    out = {}
    for name, value in re.findall(r"#define SZG_C_([A-Z]+) (0x[0-9A-Fa-f]+)u", text):
        out[name] = int(value, 16)
    return out


def default_masks():
    cls = classes()
    every = 0
    for v in cls.values():
        every |= v
    masks = [0, every]
    masks += list(cls.values())
    masks += [every & ~v for v in cls.values()]
    return sorted(set(masks))


def build(masks):
    os.makedirs(VAR, exist_ok=True)
    mk = open(os.path.join(CSRC, "Makefile")).read()
    flags = re.search(r"^HIPFLAGS \?= (.*)$", mk, re.M).group(1).replace("$(ARCH)", "gfx950")
    sched = dict(re.findall(r"^SCHED_(\w+) = (.*)$", mk, re.M))
    shared = ["kernels_raster_sort.o", "host_scene.o", "host_assets.o", "szg_comm.o"]  # no contraction site in these
    for m in masks:
        so = os.path.join(VAR, f"libszg_hip_c{m:04x}.so")
        objs = []
        procs = []
        for tu in ("kernels_lut", "kernels_deferred", "kernels_composite", "kernels_raster", "szg_api"):
            src = tu + (".cpp" if tu == "szg_api" else ".hip")
            obj = os.path.join(VAR, f"{tu}_c{m:04x}.o")
            cmd = f"/opt/rocm/bin/hipcc {flags} {sched.get(tu, '')} -DSZG_CONTRACT=0x{m:04x}u -I../../include -I. -x hip -c {src} -o {obj}"
            procs.append(subprocess.Popen(cmd, shell=True, cwd=CSRC))
            objs.append(obj)
        for p in procs:
            if p.wait() != 0:
                raise SystemExit(f"build of mask {m:#x} failed")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs +
                       [os.path.join(CSRC, o) for o in shared] + ["-ldl"], check=True)
        for o in objs:
            os.remove(o)
        print("built", so, flush=True)


def run(outdir, masks):
    os.makedirs(outdir, exist_ok=True)
    for m in masks:
        so = os.path.join(VAR, f"libszg_hip_c{m:04x}.so")
        if not os.path.exists(so):
            print("missing", so, flush=True)
            continue
        env = dict(os.environ, SZG_HIP_LIBRARY=so)
        pin = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "gpu_spirv_pin_child.py")], env=env, capture_output=True,
                             text=True, timeout=900)
        if pin.returncode != 0:
            print(f"mask {m:#x}: pin child failed: {pin.stderr[-500:]}", flush=True)
            continue
        rec = {"mask": m, "pin": json.loads(pin.stdout.strip().split("\n")[-1])}
        for wl in os.environ.get("SZG_CC_WORKLOADS", "c3,c2").split(","):
            b = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", wl, "--steps", "30", "--warmup", "5",
                                "--no-cpu-baseline", "--no-extras"], env=env, capture_output=True, text=True, timeout=900)
            if b.returncode != 0:
                print(f"mask {m:#x}: bench {wl} failed: {b.stderr[-500:]}", flush=True)
                continue
            line = json.loads(b.stdout.strip().split("\n")[-1])
            rec[wl] = {"ms_per_step": line["ms_per_step"], "passes_ms": line.get("pass_ms_rank0"), "image_checksum": line.get("image_checksum")}
        with open(os.path.join(outdir, f"c{m:04x}.json"), "w") as f:
            json.dump(rec, f)
        p = rec["pin"]
        print(f"mask {m:#05x}: camera rel {p['camera_rel_max']:.2e} lsb {p['camera_unorm_max_step']} sky {p['skyview_rel_max']:.2e} "
              f"T {p['transmittance_rel_max']:.2e} lights {p['lights_rel_max']:.2e} c3 {rec.get('c3', {}).get('ms_per_step')}", flush=True)


def table(outdir):
    cls = classes()
    every = 0
    for v in cls.values():
        every |= v
    recs = {}
    for f in os.listdir(outdir):
        if re.fullmatch(r"c[0-9a-f]{4}\.json", f):
            r = json.load(open(os.path.join(outdir, f)))
            recs[r["mask"]] = r

    def row(label, m):
        r = recs.get(m)
        if r is None:
            return f"| {label} | `{m:#05x}` | not run |"
        p = r["pin"]
        c3 = r.get("c3", {})
        passes = c3.get("passes_ms") or {}
        return (f"| {label} | `{m:#05x}` | {p['camera_rel_max']:.1e} | {p['camera_unorm_max_step']} | {p['skyview_rel_max']:.1e} | "
                f"{p['transmittance_rel_max']:.1e} | {p['lights_rel_max']:.1e} / {p['lights_unorm_max_step']} | "
                f"{c3.get('ms_per_step', float('nan')):.3f} | {passes.get('composite', float('nan')):.3f} | {passes.get('skyview', float('nan')):.3f} | "
                f"{(r.get('c5', {}).get('passes_ms') or {}).get('lights', float('nan')):.3f} |")

    head = ("| build | mask | camera.comp rel max | UNORM16 steps | sky-view rel max | transmittance rel max | lights rel / steps | "
            "C3 frame ms | composite ms | sky-view ms | C5 lights ms |\n|---|---|---|---|---|---|---|---|---|---|---|")
    print(head)
    print(row("literal (nothing fused)", 0))
    print(row("everything fused (round 2's product)", every))
    for name, v in cls.items():
        print(row(f"only {name} fused", v))
    for name, v in cls.items():
        print(row(f"all but {name} fused", every & ~v))
    print(row("**the product's rule** (MATVEC, MIX, TEXCOORD, STEP, ACCUM, PBRDOT, LDOT)", 0x3616))
    print(row("the product's rule + LUTMAP (inside the bar with a margin of 1.5; not taken)", 0x3636))
    for m in sorted(recs):
        if m in (0x3616, 0x3636):
            continue
        if m not in (0, every) and m not in cls.values() and m not in [every & ~v for v in cls.values()]:
            print(row("combination", m))


if __name__ == "__main__":
    cmd = sys.argv[1]
    if cmd == "build":
        build([int(a, 0) for a in sys.argv[2:]] or default_masks())
    elif cmd == "run":
        run(sys.argv[2], [int(a, 0) for a in sys.argv[3:]] or default_masks())
    elif cmd == "table":
        table(sys.argv[2])
