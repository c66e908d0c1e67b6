#!/bin/bash
# Run ON THE GPU BOX from the repo root:  bash tools/sq_profile.sh r01
# SQ counters of the fused rollout kernel (fp32 and bf16 actor): issue/wait split, MFMA busy and MFMA+VALU
# co-execution cycles, instruction counts.  Two passes per precision (8 SQ slots per pass).
set -u
TAG=${1:-r01}
R=$PWD
OUT=$R/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
A="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES"
B="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE"
for prec in fp32 bf16; do
  timeout 600 rocprofv3 --pmc $A --kernel-trace --output-format csv -d $OUT/${prec}_a -o p -- python $R/tools/launch_fit.py --precision $prec > $OUT/${prec}_a.log 2>&1
  echo "$prec pass A rc=$?"
  timeout 600 rocprofv3 --pmc $B --kernel-trace --output-format csv -d $OUT/${prec}_b -o p -- python $R/tools/launch_fit.py --precision $prec > $OUT/${prec}_b.log 2>&1
  echo "$prec pass B rc=$?"
done
cd $R
python - "$OUT" "$TAG" <<'PY'
import collections, csv, json, os, sys
src, tag = sys.argv[1], sys.argv[2]
out = {}
for prec in ("fp32", "bf16"):
    rec = {}
    for p in ("a", "b"):
        path = os.path.join(src, f"{prec}_{p}", "p_counter_collection.csv")
        if not os.path.exists(path):
            continue
        per = collections.defaultdict(dict)
        for r in csv.DictReader(open(path)):
            if "k_rollout_fused" not in r["Kernel_Name"]:
                continue
            d = per[r["Dispatch_Id"]]
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            d["_dur_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        # the longest dispatch is tools/launch_fit.py's 2000-step warm-up launch of 65 536 envs (1 024 waves): it is what
        # the counters below describe, `wave_steps` = 1 024 x 2 000 turns them into per-wave-step figures
        longest = max(v["_dur_us"] for v in per.values())
        sel = [v for v in per.values() if v["_dur_us"] > 0.8 * longest]
        for k in sel[0]:
            rec[k if k != "_dur_us" else f"dur_us_pass_{p}"] = sum(v[k] for v in sel) / len(sel)
        rec[f"launches_pass_{p}"] = len(sel)
    rec["wave_steps"] = 1024 * 2000
    out[prec] = rec
dst = os.path.join(os.path.dirname(src), f"profiles_{tag}")
os.makedirs(dst, exist_ok=True)
json.dump(out, open(os.path.join(dst, f"{tag}_sq_counters.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
PY
