#!/usr/bin/env python3
"""Condense a tools/profile_round.sh output directory into profiles/<tag>_* (small, committed)."""
import collections
import csv
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "gpurun_out", f"profiles_{tag}")
os.makedirs(dst, exist_ok=True)


def short(name):
    """kernel name without its argument list (template arguments kept)"""
    depth = 0
    for i, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            name = name[:i]
            break
    return name.replace("void ", "").strip()


# 1. kernel stats of the bench command
stats = os.path.join(src, "trace", "bench_kernel_stats.csv")
shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
rows = list(csv.DictReader(open(stats)))

# 2. PMC passes: per (kernel, grid) average counter value and duration
pmc = collections.defaultdict(dict)
for which, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    path = os.path.join(src, which, "bench_counter_collection.csv")
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        key = (short(r["Kernel_Name"]), int(r["Grid_Size"]))
        agg[key].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                         int(r["VGPR_Count"]), int(r["Accum_VGPR_Count"]), int(r["SGPR_Count"])))
    for key, v in agg.items():
        d = pmc[key]
        d[counter + "_KiB_avg"] = sum(x[0] for x in v) / len(v)
        d["calls_" + which] = len(v)
        d["avg_dur_us_" + which] = sum(x[1] for x in v) / len(v) / 1e3
        d["vgpr"], d["agpr"], d["sgpr"] = v[0][2], v[0][3], v[0][4]

out = {}
for (name, grid), d in sorted(pmc.items()):
    if not name.startswith("rq::"):
        continue
    f = d.get("FETCH_SIZE_KiB_avg")
    w = d.get("WRITE_SIZE_KiB_avg")
    # gfx950: FETCH_SIZE reports 1/2 of the bytes of a coalesced streaming read (MI355X_MICROARCH.md
    # "HBM"; calibrated here on k_observe, whose reads are exactly 92 B/env) -> double it; WRITE_SIZE
    # matched the byte count of k_observe's 104 B/env stores exactly -> used as is.  Units: KiB.
    traffic = None if f is None or w is None else (2.0 * f + w) * 1024.0
    out[f"{name}@{grid}"] = {**{k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.items()},
                             "hbm_bytes_per_launch_corrected": traffic,
                             "hbm_bytes_per_env": None if traffic is None else round(traffic / grid, 2)}
json.dump(out, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1, sort_keys=True)

for fn in ("bench_trace.json", "bench_fetch.json", "bench_write.json"):
    p = os.path.join(src, fn)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, f"{tag}_{fn}"))

with open(os.path.join(dst, f"{tag}_summary.md"), "w") as f:
    f.write(f"# rocprofv3 summary {tag}\n\nCommand: `python bench.py --no-cpu-baseline` (kernel trace); "
            "PMC passes add `--steps 1000 --warmup 0`.\n\n## kernel-trace --stats\n\n")
    f.write("| kernel | calls | avg us | total % |\n|---|---|---|---|\n")
    for r in rows:
        f.write(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.2f} | {r['Percentage']} |\n")
    # the dominant kernel, call by call: --stats averages every launch of the process, including the first
    # ones at ramping clocks and the probes' launches; the timed region of bench.py is a known slice of them
    trace = os.path.join(src, "trace", "bench_kernel_trace.csv")
    bench_json = os.path.join(src, "bench_trace.json")
    if os.path.exists(trace) and os.path.exists(bench_json):
        b = json.loads(open(bench_json).read().strip().splitlines()[-1])
        durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(trace))
                if "k_rollout_fused<false, true, false" in r["Kernel_Name"]]
        untimed = 1 + -(-b["config"]["untimed_steps_before_timing"] // 500)     # the 1-step launch + warm-up chunks
        timed = -(-b["steps"] // 500)
        region = durs[untimed:untimed + timed]
        if region:
            f.write(f"\nDominant kernel `k_rollout_fused`, launch by launch (us): first {untimed} launches are "
                    f"untimed (1 step, then {untimed - 1} x 500 steps while the clocks ramp): "
                    f"{', '.join(str(round(x)) for x in durs[:untimed])}; the {timed} launches of the timed region: "
                    f"mean **{sum(region) / len(region):.1f}**, min {min(region):.0f}, max {max(region):.0f} "
                    f"(bench.py's HIP events in the same run: {b['roofline']['avg_launch_ms'] * 1e3:.1f} per launch incl. the "
                    f"per-chunk copy of the returns); later launches belong to the probes.\n")
    f.write("\n## PMC (separate passes; FETCH_SIZE doubled per the gfx950 correction)\n\n")
    f.write("| kernel@grid | VGPR/AGPR/SGPR | avg us | FETCH KiB | WRITE KiB | HBM bytes/env (corrected) |\n|---|---|---|---|---|---|\n")
    for k, d in out.items():
        f.write(f"| `{k}` | {d.get('vgpr')}/{d.get('agpr')}/{d.get('sgpr')} | {d.get('avg_dur_us_fetch', 0):.2f} | "
                f"{d.get('FETCH_SIZE_KiB_avg', 0):.1f} | {d.get('WRITE_SIZE_KiB_avg', 0):.1f} | {d.get('hbm_bytes_per_env')} |\n")
print("wrote", dst, os.listdir(dst))
