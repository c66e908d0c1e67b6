#!/bin/bash
# Run ON THE GPU BOX from the repo root:  bash tools/teacher_traffic.sh r06
# HBM traffic of the dense-stack teacher kernel (k_teacher_relabel_layers) against its algorithmic bytes: FETCH_SIZE and WRITE_SIZE in
# separate rocprofv3 --pmc passes (MI355X_MICROARCH.md: they cannot share a pass; FETCH_SIZE doubled, the gfx950 correction) over
# tools/teacher_rate.py --only-layers (65 536 envs x 500 steps, 1 000 teachers, contiguous assignment) -> gpurun_out/profiles_<tag>/<tag>_teacher_pmc.json
set -u
TAG=${1:-r06}
R=$PWD
DST=$R/gpurun_out/profiles_$TAG
OUT=$R/gpurun_out/teacher_pmc_$TAG
mkdir -p $DST $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/teacher_rate.py --teachers 1000 --assignment contiguous --only-layers"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o p -- $CMD > $OUT/fetch.log 2>&1; echo "fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o p -- $CMD > $OUT/write.log 2>&1; echo "write rc=$?"
cd $R
python - "$OUT" "$DST" "$TAG" <<'PY'
import collections, csv, json, os, sys
src, dst, tag = sys.argv[1:4]
sys.path.insert(0, os.getcwd())
from bench import library_sha256
N, T, TEACHERS = 65536, 500, 1000
per = collections.defaultdict(dict)
for which, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    rows = collections.defaultdict(list)
    for r in csv.DictReader(open(os.path.join(src, which, "p_counter_collection.csv"))):
        if "k_teacher_relabel_layers" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            rows[r["Dispatch_Id"]].append(r)
    order = sorted(rows, key=int)
    # teacher_rate.py labels four topologies in turn, six launches each (2 warm-up + 4 timed)
    for i, disp in enumerate(order):
        topo = ("22-128-128-128-4", "22-128-128-4", "22-64-64-64-4", "22-128-4")[min(i // 6, 3)]
        v = sum(float(r["Counter_Value"]) for r in rows[disp])
        d = per[topo]
        d.setdefault(counter, []).append(v)
        d.setdefault("dur_us", []).append((int(rows[disp][0]["End_Timestamp"]) - int(rows[disp][0]["Start_Timestamp"])) / 1e3)
out = {"_library_sha256": library_sha256(), "command": "tools/teacher_rate.py --teachers 1000 --assignment contiguous --only-layers",
       "envs": N, "steps": T, "teachers": TEACHERS}
for topo, d in per.items():
    widths = [int(x) for x in topo.split("-")]
    params = sum(widths[i + 1] * widths[i] + widths[i + 1] for i in range(len(widths) - 1))
    algorithmic = N * T * (88 + 16) + TEACHERS * params * 4                      # observations in + actions out + every teacher's parameters once
    f = sorted(d["FETCH_SIZE"])[len(d["FETCH_SIZE"]) // 2] * 1024.0 * 2.0      # KiB; doubled: gfx950 reports half of coalesced reads
    w = sorted(d["WRITE_SIZE"])[len(d["WRITE_SIZE"]) // 2] * 1024.0
    out[topo] = {"launches": len(d["FETCH_SIZE"]), "median_dur_us": sorted(d["dur_us"])[len(d["dur_us"]) // 2],
                 "fetch_bytes_corrected": f, "write_bytes": w, "hbm_bytes_per_launch_corrected": f + w,
                 "algorithmic_bytes": algorithmic, "traffic_over_algorithmic": round((f + w) / algorithmic, 3)}
json.dump(out, open(os.path.join(dst, f"{tag}_teacher_pmc.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
PY
