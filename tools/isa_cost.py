#!/usr/bin/env python3
"""Static issue-cost model of a gfx950 kernel from its assembly (hipcc -S --cuda-device-only).

Prices every VALU instruction at the per-SIMD issue cost measured by tools/opcost.hip on MI355X (profiles/r02_opcost.txt):
    full rate, 32-bit encoding (v_add/mul/sub/mov/fmac_f32, v_add/sub_u32, v_xor/and/or_b32 ...)        2.04 cycles
    full rate, 64-bit encoding (v_fma_f32, VOP3 forms of the above)                                       2.32
    half rate (v_max/min/med3, v_floor/fract/cvt/ldexp/frexp, v_lshl*, v_bfi/and_or, v_mul_lo/mad_u24,
               v_cmp*, v_cndmask, v_div_scale/fmas/fixup, packed f32: 2 results)                          4.08 - 4.6
    transcendental (v_rcp/rsq/sqrt/exp/log/sin/cos)                                                       8.06
and lists, for every loop (backward branch) and for the whole kernel, the instruction count and cycles per class.

    python tools/isa_cost.py file.s [kernel-name-substring] [--loops] [--top N]
"""
import collections
import re
import sys

TRANS = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32",
         "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64"}
FULL32 = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_mov_b32", "v_fmac_f32", "v_mac_f32", "v_add_u32", "v_sub_u32",
          "v_subrev_u32", "v_xor_b32", "v_and_b32", "v_or_b32", "v_not_b32", "v_add_co_u32", "v_sub_co_u32", "v_addc_co_u32",
          "v_subb_co_u32", "v_subrev_co_u32", "v_accvgpr_read_b32", "v_accvgpr_write_b32", "v_nop"}
FULL64 = {"v_fma_f32", "v_mad_f32", "v_add3_u32", "v_xad_u32", "v_or3_b32", "v_and_or_b32_FULL?"}


def classify(op, text):
    base = op
    for suffix in ("_e32", "_e64", "_dpp", "_sdwa"):
        if base.endswith(suffix):
            base = base[: -len(suffix)]
    if base in TRANS:
        return "trans", 8.06
    if base.startswith("v_pk_"):
        return "packed", 4.35
    if base in FULL32:
        # a VOP3 encoding (explicit _e64, modifiers, an SGPR/literal in a slot VOP2 cannot hold) costs the 64-bit fetch
        vop3 = op.endswith("_e64") or "|" in text or " neg(" in text or "clamp" in text or " mul:" in text or " div:" in text
        return ("full64", 2.32) if vop3 else ("full32", 2.04)
    if base in FULL64:
        return "full64", 2.32
    if base.startswith("v_cmp") or base.startswith("v_cndmask"):
        return "cmp/select", 4.08
    if base.startswith("v_readlane") or base.startswith("v_readfirstlane") or base.startswith("v_writelane"):
        return "lane", 4.5
    if base.startswith("v_mfma") or base.startswith("v_smfma"):
        return "mfma", 0.0
    if base.startswith("v_"):
        return "half", 4.3 if base in {"v_med3_f32", "v_bfi_b32", "v_and_or_b32", "v_mul_lo_u32", "v_mad_u32_u24", "v_add_lshl_u32",
                                       "v_lshl_add_u32", "v_div_fixup_f32", "v_div_fmas_f32", "v_div_scale_f32", "v_ldexp_f32",
                                       "v_fma_mix_f32", "v_mad_u64_u32", "v_perm_b32", "v_alignbit_b32", "v_max3_f32",
                                       "v_min3_f32", "v_lshl_or_b32"} else 4.08
    if base.startswith("s_"):
        return "scalar", 0.0
    if base.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem", 0.0
    if base.startswith("ds_"):
        return "lds", 0.0
    return "other", 0.0


def parse(path, want):
    kernels = collections.OrderedDict()
    name = None
    for raw in open(path):
        line = raw.rstrip("\n")
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m:
            name = m.group(1)
            kernels[name] = []
            continue
        if name is None:
            continue
        s = line.strip()
        if s.startswith(".Lfunc_end") or s.startswith(".section") or s.startswith(".rodata"):
            name = None
            continue
        kernels[name].append(line)
    return collections.OrderedDict((k, v) for k, v in kernels.items() if want in k and any("s_endpgm" in x for x in v))


def blocks_of(lines):
    """[(label, [(op, text)])] in program order; instructions before the first label belong to 'entry'."""
    out = [("entry", [])]
    for line in lines:
        s = line.split(";")[0].strip()
        if not s or s.startswith(".") and not s.endswith(":"):
            continue
        m = re.match(r"^(\.LBB\w+):", s)
        if m:
            out.append((m.group(1), []))
            continue
        if s.startswith(";") or s.endswith(":"):
            continue
        op = s.split()[0]
        out[-1][1].append((op, s))
    return out


def summarize(instrs):
    counts, cycles = collections.Counter(), collections.Counter()
    ops = collections.Counter()
    for op, text in instrs:
        cls, c = classify(op, text)
        counts[cls] += 1
        cycles[cls] += c
        ops[(cls, op)] += 1
    return counts, cycles, ops


def report(title, instrs, top):
    counts, cycles, ops = summarize(instrs)
    valu = sum(n for c, n in counts.items() if c in ("full32", "full64", "half", "trans", "packed", "cmp/select", "lane"))
    total = sum(cycles.values())
    print(f"{title}: {len(instrs)} instructions, {valu} VALU, {total:.0f} issue cycles"
          + (f" ({total / valu:.2f} per VALU instruction)" if valu else ""))
    for cls in ("full32", "full64", "half", "cmp/select", "trans", "packed", "lane", "scalar", "vmem", "lds", "other"):
        if counts[cls]:
            print(f"    {cls:11s} {counts[cls]:6d}  {cycles[cls]:8.0f} cycles")
    if top:
        priced = sorted(((n * classify(op, "")[1], n, cls, op) for (cls, op), n in ops.items()), reverse=True)[:top]
        print("    costliest opcodes: " + ", ".join(f"{op} x{n} = {c:.0f}" for c, n, cls, op in priced if c > 0))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    path = args[0]
    want = args[1] if len(args) > 1 else ""
    loops = "--loops" in sys.argv
    top = 0
    if "--top" in sys.argv:
        top = int(sys.argv[sys.argv.index("--top") + 1])
    for name, lines in parse(path, want).items():
        blocks = blocks_of(lines)
        index = {label: i for i, (label, _) in enumerate(blocks)}
        everything = [x for _, ins in blocks for x in ins]
        print("=" * 100)
        report(name[:90], everything, top)
        if not loops:
            continue
        # loops = backward branches; body = blocks from the target to the branching block (program order)
        found = []
        for i, (label, ins) in enumerate(blocks):
            for op, text in ins:
                if op.startswith("s_cbranch") or op == "s_branch":
                    target = text.split()[-1]
                    if target in index and index[target] <= i:
                        found.append((index[target], i))
        for begin, end in sorted(set(found)):
            body = [x for _, ins in blocks[begin : end + 1] for x in ins]
            inner = [f for f in found if begin <= f[0] and f[1] <= end and f != (begin, end)]
            report(f"  loop {blocks[begin][0]} .. {blocks[end][0]}" + (f" (contains {len(inner)} inner loops)" if inner else ""), body, top)


if __name__ == "__main__":
    main()
