#!/usr/bin/env python3
"""Collects the rocprofv3 evidence bench.py's `roofline` / `roofline_valu` objects refer to, for ONE workload, on the GPU box.

    python tools/collect_pmc.py --workload c3 --out gpurun_out/r02_pmc_c3.json [--stats-out gpurun_out/r02_kernel_stats_c3.csv]

Passes (each a separate rocprofv3 run of `python3 bench.py --workload W --no-extras --no-cpu-baseline`, the program directly
behind `--`; counters never share a run with a trace, MI355X_MICROARCH.md "HBM / rocprofv3"):
    1. --kernel-trace --stats                         -> average launch duration per kernel
    2. --pmc FETCH_SIZE                               -> KB fetched from HBM per launch (gfx950 under-reports wide reads: x2)
    3. --pmc WRITE_SIZE                               -> KB written per launch (exact)
    4. --pmc SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F32    -> fp32 instruction classes
    5. --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SALU
    6. --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
    7. --pmc GRBM_GUI_ACTIVE                          -> effective clock = value / 8 XCDs / duration
The JSON holds, per szg:: kernel, the mean of every counter over its launches, the corrected HBM bytes per launch and the VALU
issue cycles per launch with every class priced at the cost tools/opcost.hip measured (profiles/r02_opcost.txt).
Copy the outputs into profiles/ (gpurun_out/ is scratch)."""
import argparse
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = [
    ["FETCH_SIZE"],
    ["WRITE_SIZE"],
    ["SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32"],
    ["SQ_INSTS_VALU", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_CVT", "SQ_INSTS_SALU"],
    ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU"],
    ["GRBM_GUI_ACTIVE"],
]
# SIMD-cycles per wave64 instruction, profiles/r02_opcost.txt (tools/opcost.hip: in-kernel shader clock, >= 2 waves per SIMD):
# add / mul 2.04 (32-bit encodings), fma 2.32 (64-bit encoding), transcendentals 8.06, conversions 4.08, int32 a mixture of
# full-rate (add, xor, and: 2.04) and half-rate (shifts, mul_lo, mad: 4.1 - 4.6) opcodes priced at 3.3, everything else that
# is VALU (min / max / med3, floor / fract, compare, select, bfi, moves) at the half rate 4.08 except the moves.
CLASS_COST = {"add_f32": 2.04, "mul_f32": 2.04, "fma_f32": 2.32, "trans_f32": 8.06, "cvt": 4.08, "int32": 3.3, "other": 4.08}


def run(cmd, log):
    log.write("$ " + " ".join(cmd) + "\n")
    log.flush()
    subprocess.run(cmd, stdout=log, stderr=subprocess.STDOUT, check=True, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), timeout=600)


def kernel_key(name):
    # "void szg::k_composite<false>(...)" -> "k_composite"
    base = name.split("(")[0].strip()
    base = base.split("szg::")[-1]
    return base.split("<")[0].strip()


def _source_hash():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry

    return entry.source_hash("hip")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--stats-out", default=None)
    ap.add_argument("--steps", type=int, default=6)
    args = ap.parse_args()
    bench = ["python3", os.path.join(ROOT, "bench.py"), "--workload", args.workload, "--steps", str(args.steps), "--warmup", "2",
             "--no-cpu-baseline", "--no-extras"]
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    log = open(os.path.abspath(args.out) + ".log", "w")
    scratch = f"/tmp/szg_pmc_{args.workload}"
    subprocess.run(["rm", "-rf", scratch])

    run(["rocprofv3", "--output-format", "csv", "--kernel-trace", "--stats", "-d", scratch + "/stats", "-o", "s", "--"] + bench, log)
    stats = {}
    for f in glob.glob(scratch + "/stats/**/*kernel_stats.csv", recursive=True):
        if args.stats_out:
            subprocess.run(["cp", f, args.stats_out])
        for r in csv.DictReader(open(f)):
            if "szg::" in r["Name"]:
                stats[kernel_key(r["Name"])] = {"calls": int(r["Calls"]), "avg_launch_ms": float(r["AverageNs"]) / 1e6}

    counters = collections.defaultdict(lambda: collections.defaultdict(list))
    for i, names in enumerate(PASSES):
        d = f"{scratch}/pmc{i}"
        run(["rocprofv3", "--output-format", "csv", "--pmc"] + names + ["-d", d, "-o", "p", "--"] + bench, log)
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "szg::" in r["Kernel_Name"]:
                    counters[kernel_key(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))

    kernels = {}
    for k, cs in sorted(counters.items()):
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        m["launches_counted"] = max(len(v) for v in cs.values())
        if k in stats:
            m.update(stats[k])
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            m["hbm_bytes_per_launch_raw"] = (m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0
            m["hbm_bytes_per_launch"] = (2.0 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0  # gfx950 wide-read correction
        if "SQ_INSTS_VALU" in m and "SQ_INSTS_VALU_ADD_F32" in m:
            classes = {"add_f32": m["SQ_INSTS_VALU_ADD_F32"], "mul_f32": m["SQ_INSTS_VALU_MUL_F32"], "fma_f32": m["SQ_INSTS_VALU_FMA_F32"],
                       "trans_f32": m["SQ_INSTS_VALU_TRANS_F32"], "cvt": m.get("SQ_INSTS_VALU_CVT", 0.0), "int32": m.get("SQ_INSTS_VALU_INT32", 0.0)}
            classes["other"] = max(0.0, m["SQ_INSTS_VALU"] - sum(classes.values()))
            m["valu_classes_per_launch"] = classes
            m["valu_issue_cycles_per_launch"] = sum(n * CLASS_COST[c] for c, n in classes.items())
        if "GRBM_GUI_ACTIVE" in m and "avg_launch_ms" in m:
            m["effective_clock_GHz"] = m["GRBM_GUI_ACTIVE"] / 8.0 / (m["avg_launch_ms"] * 1e-3) / 1e9
        kernels[k] = m
    out = {
        "workload": args.workload,
        "source_hash": _source_hash(),  # the tree the profiled library was built from (__graft_entry__.source_hash("hip"))
        "command": " ".join(bench),
        "tool": "rocprofv3: --kernel-trace --stats alone, then one --pmc pass per counter group (tools/collect_pmc.py)",
        "correction": "gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads: doubled; WRITE_SIZE exact; both in KB",
        "class_issue_cycles": CLASS_COST,
        "class_issue_cycles_source": "profiles/r02_opcost.txt (tools/opcost.hip)",
        "kernels": kernels,
    }
    json.dump(out, open(args.out, "w"), indent=1, sort_keys=True)
    print(f"wrote {args.out}: {sorted(kernels)}")


if __name__ == "__main__":
    sys.exit(main())
