#!/bin/bash
# Run ON THE GPU BOX from the repo root:  bash tools/profile_extras.sh r02
# Side configurations of bench.py (BASELINE configs 3, 5, the larger batches, the API-granular chain) and the SQ
# counters of the teacher-bank kernels -> gpurun_out/profiles_<tag>/
set -u
TAG=${1:-r02}
R=$PWD
DST=$R/gpurun_out/profiles_$TAG
mkdir -p $DST
python bench.py > $DST/${TAG}_bench_default.json 2> $DST/default.err; echo "default rc=$?"
python bench.py --precision bf16 --no-cpu-baseline --no-kernel-probe > $DST/${TAG}_bench_bf16.json 2>/dev/null; echo "bf16 rc=$?"
python bench.py --precision f16x2 --no-cpu-baseline --no-kernel-probe > $DST/${TAG}_bench_f16x2.json 2>/dev/null; echo "f16x2 rc=$?"
python bench.py --envs-per-gpu 262144 --no-cpu-baseline --no-kernel-probe > $DST/${TAG}_bench_262144.json 2>/dev/null; echo "262144 rc=$?"
python bench.py --envs-per-gpu 1048576 --no-cpu-baseline --no-kernel-probe > $DST/${TAG}_bench_1048576.json 2>/dev/null; echo "1048576 rc=$?"
python bench.py --mode chained --steps 2000 --warmup 500 --no-cpu-baseline --no-kernel-probe > $DST/${TAG}_bench_chained.json 2>/dev/null; echo "chained rc=$?"
python tools/teacher_rate.py --teachers 1024 > $DST/${TAG}_teacher_rate.json 2>/dev/null
python tools/teacher_rate.py --teachers 1000 >> $DST/${TAG}_teacher_rate.json 2>/dev/null
python tools/teacher_rate.py --teachers 1000 --assignment contiguous >> $DST/${TAG}_teacher_rate.json 2>/dev/null
OUT=$R/gpurun_out/sq_teacher_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
A="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA"
timeout 600 rocprofv3 --pmc $A --kernel-trace --output-format csv -d $OUT/a -o p -- python $R/tools/teacher_rate.py --teachers 1024 > $OUT/a.log 2>&1
echo "teacher sq rc=$?"
cd $R
python - "$OUT" "$DST" "$TAG" <<'PY'
import collections, csv, json, os, sys
src, dst, tag = sys.argv[1:4]
per = collections.defaultdict(lambda: collections.defaultdict(dict))
for r in csv.DictReader(open(os.path.join(src, "a", "p_counter_collection.csv"))):
    if "k_teacher_relabel" not in r["Kernel_Name"]:
        continue
    name = "f32" if "relabel_f32" in r["Kernel_Name"] else "bf16"
    d = per[name][r["Dispatch_Id"]]
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    d["dur_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
out = {}
for name, disp in per.items():
    sel = list(disp.values())
    rec = {k: sum(v[k] for v in sel) / len(sel) for k in sel[0]}
    rec["launches"] = len(sel)
    rec["mfma_busy_frac_of_wave_cycles"] = rec["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * rec["SQ_WAVE_CYCLES"])
    rec["mfma_valu_coexec_frac_of_busy"] = rec["SQ_VALU_MFMA_COEXEC_CYCLES"] / max(rec["SQ_VALU_MFMA_BUSY_CYCLES"], 1.0)
    out[name] = rec
json.dump(out, open(os.path.join(dst, f"{tag}_teacher_sq_counters.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
PY
