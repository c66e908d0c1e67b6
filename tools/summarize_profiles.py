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
        # the MEDIAN launch: one kernel instantiation is launched with different step counts (the region's
        # launches and a couple of warm-up ones); the median is the region's
        d[counter + "_KiB_avg"] = sorted(x[0] for x in v)[len(v) // 2]
        d["median_dur_us"] = sorted(x[1] for x in v)[len(v) // 2] / 1e3
        d["calls_" + which] = len(v)
        d["avg_dur_us_" + which] = sum(x[1] for x in v) / len(v) / 1e3
        d["vgpr"], d["agpr"], d["sgpr"] = v[0][2], v[0][3], v[0][4]

sys.path.insert(0, root)
from bench import launch_grid                      # noqa: E402


from bench import pmc_key                          # noqa: E402


def envs_of(name, grid):
    """env count of a dispatch: the grid for most kernels (rounded up to the workgroup); the streaming instantiation of
    k_actor_stream packs several 64-env groups per wave, so its grid is matched against the batches bench.py launches it at"""
    if name.startswith("rq::k_actor_stream"):
        for n in (2097152, 1048576, 524288, 262144):
            if launch_grid(name, n) == grid:
                return n
    return grid


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
    out[pmc_key(name, envs_of(name, grid))] = {**{k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.items()},
                             "grid": grid,
                             "hbm_bytes_per_launch_corrected": traffic,
                             "envs": envs_of(name, grid),
                             "hbm_bytes_per_env": None if traffic is None else round(traffic / envs_of(name, grid), 2)}
# which build ran under the profiler (round 6): bench.py uses a committed summary only for the build it was taken from
from bench import library_sha256                   # noqa: E402
out["_library_sha256"] = library_sha256()
json.dump(out, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1, sort_keys=True)
out.pop("_library_sha256")

for fn in ("bench_trace.json", "bench_fetch.json", "bench_write.json"):
    p = os.path.join(src, fn)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, f"{tag}_{fn}"))

with open(os.path.join(dst, f"{tag}_summary.md"), "w") as f:
    cmd = open(os.path.join(src, "command.txt")).read().strip() if os.path.exists(os.path.join(src, "command.txt")) else "python bench.py"
    f.write(f"# rocprofv3 summary {tag}\n\nCommand (all three passes): `{cmd}`\n\n## kernel-trace --stats\n\n")
    f.write("| kernel | calls | avg us | total % |\n|---|---|---|---|\n")
    for r in rows:
        f.write(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.2f} | {r['Percentage']} |\n")
    # the dominant kernel, instantiation by instantiation, from the per-dispatch trace: --stats averages every
    # launch of the process; bench.py launches the short-launch build for its timed regions when --steps < 48
    # and the 256-register build for the 500-step launches of `steady_state` and of the probes
    trace = os.path.join(src, "trace", "bench_kernel_trace.csv")
    bench_json = os.path.join(src, "bench_trace.json")
    if os.path.exists(trace) and os.path.exists(bench_json):
        b = json.loads(open(bench_json).read().strip().splitlines()[-1])
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(trace)):
            if "k_rollout_fused" in r["Kernel_Name"]:
                per[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        f.write("\n## `k_rollout_fused`, per instantiation (us per launch, from the kernel trace)\n\n"
                "| instantiation | launches | mean | median | min | max |\n|---|---|---|---|---|---|\n")
        launch_stats = {"command": cmd, "steps": b["steps"], "envs_per_gpu": b.get("config", {}).get("envs_per_gpu"), "source": "rocprofv3 --kernel-trace, End - Start per dispatch (us)",
                        # the build that ran under the profiler, as the profiled run itself reported it; bench.py takes its headline
                        # fraction from this file only when the library it has loaded hashes to the same value
                        "library_sha256": b.get("roofline", {}).get("library_sha256") or library_sha256()}
        for name, d in sorted(per.items()):
            ds = sorted(d)
            f.write(f"| `{name}` | {len(d)} | {sum(d) / len(d):.2f} | {ds[len(ds) // 2]:.2f} | {ds[0]:.2f} | {ds[-1]:.2f} |\n")
            # one instantiation is launched with different step counts (the timed regions' launches, the 500-step
            # launches of steady_state and of the probes, two warm-up launches): split them by duration
            med = ds[len(ds) // 2]
            short = [x for x in ds if 0.5 * med <= x <= 2.0 * med]
            long_ = [x for x in ds if x > 5.0 * med]
            if long_ and len(short) > 10:
                launch_stats[name.replace(" ", "")] = {
                    "timed_region_launches": {"launches": len(short), "mean_us": round(sum(short) / len(short), 2),
                                              "median_us": round(short[len(short) // 2], 2)},
                    "launches_of_500_steps": {"launches": len(long_), "mean_us": round(sum(long_) / len(long_), 2),
                                              "median_us": round(long_[len(long_) // 2], 2)}}
                f.write(f"| &nbsp;&nbsp;of which the timed regions' launches ({b['steps']} steps) | {len(short)} | "
                        f"{sum(short) / len(short):.2f} | {short[len(short) // 2]:.2f} | {short[0]:.2f} | {short[-1]:.2f} |\n")
                f.write(f"| &nbsp;&nbsp;of which 500-step launches (steady_state, probes) | {len(long_)} | "
                        f"{sum(long_) / len(long_):.2f} | {long_[len(long_) // 2]:.2f} | {long_[0]:.2f} | {long_[-1]:.2f} |\n")
        rl, ss, tm = b.get("roofline", {}), b.get("steady_state", {}), b.get("timing", {})
        f.write(f"\nbench.py in the same (profiled) run: timed regions of {b['steps']} steps x {tm.get('repetitions')} repetitions, "
                f"`roofline.kernel` = `{rl.get('kernel')}`, `roofline.avg_launch_ms` = {rl.get('avg_launch_ms')}; "
                f"`steady_state.kernel` = `{ss.get('kernel')}`, `steady_state.avg_launch_ms` = {ss.get('avg_launch_ms')}.  "
                "bench.py takes these from the kernels' own first-wave-in / last-wave-out spans (per-wave wall-clock records, "
                "rq_device_set_rollout_timing) of launches inside regions of the timed regions' own cadence.  The profiler's "
                "per-dispatch duration of the SAME launches reads ~2.6 us more (a 20-step launch; ~10 us of a 500-step one's 1 420): "
                "it begins when the command processor takes the dispatch and ends when the kernel's writes have been released, "
                "the waves' own clocks begin with their first instruction and end with their last.  `roofline.rocprofv3` in "
                "bench.py's line carries the figures of this table (from " + f"{tag}_fused_launch_stats.json" + ") and the fraction "
                "they give; the clock of these profiled launches is `roofline.clock_ghz_under_load` of the profiled run's line "
                f"({rl.get('clock_ghz_under_load')} GHz; the same command un-profiled: DESIGN.md section 6).\n")
        json.dump(launch_stats, open(os.path.join(dst, f"{tag}_fused_launch_stats.json"), "w"), indent=1, sort_keys=True)
    f.write("\n## PMC (separate passes; FETCH_SIZE doubled per the gfx950 correction)\n\n")
    f.write("| kernel#n<envs> | envs | VGPR/AGPR/SGPR | avg us | FETCH KiB | WRITE KiB | HBM bytes/env (corrected) |\n|---|---|---|---|---|---|---|\n")
    for k, d in out.items():
        f.write(f"| `{k}` | {d.get('envs')} | {d.get('vgpr')}/{d.get('agpr')}/{d.get('sgpr')} | {d.get('avg_dur_us_fetch', 0):.2f} | "
                f"{d.get('FETCH_SIZE_KiB_avg', 0):.1f} | {d.get('WRITE_SIZE_KiB_avg', 0):.1f} | {d.get('hbm_bytes_per_env')} |\n")
print("wrote", dst, os.listdir(dst))
