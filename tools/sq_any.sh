#!/bin/bash
# Run ON THE GPU BOX:  bash tools/sq_any.sh <tag> <kernel substring> -- <command...>
# SQ issue/wait counters of the dispatches of one kernel under any command -> gpurun_out/sq_<tag>.json
# (SQ_A / SQ_B in the environment replace the two counter sets, 8 SQ slots per pass)
set -u
TAG=$1; KERN=$2; shift 3
R=$PWD
OUT=$R/gpurun_out/sqany_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
A=${SQ_A:-"SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES"}
B=${SQ_B:-"SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"}
(cd $R && timeout 600 rocprofv3 --pmc $A --kernel-trace --output-format csv -d $OUT/a -o p -- "$@" > $OUT/a.log 2>&1); echo "pass A rc=$?"
(cd $R && timeout 600 rocprofv3 --pmc $B --kernel-trace --output-format csv -d $OUT/b -o p -- "$@" > $OUT/b.log 2>&1); echo "pass B rc=$?"
cd $R
python - "$OUT" "$TAG" "$KERN" <<'PY'
import collections, csv, json, os, sys
src, tag, kern = sys.argv[1:4]
rec = {}
for p in ("a", "b"):
    path = os.path.join(src, p, "p_counter_collection.csv")
    if not os.path.exists(path):
        continue
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(path)):
        if kern not in r["Kernel_Name"]:
            continue
        d = per[r["Dispatch_Id"]]
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        d["_dur_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if not per:
        continue
    longest = max(v["_dur_us"] for v in per.values())
    sel = [v for v in per.values() if v["_dur_us"] > 0.8 * longest]
    for k in sel[0]:
        rec[k if k != "_dur_us" else f"dur_us_pass_{p}"] = sum(v[k] for v in sel) / len(sel)
    rec[f"launches_pass_{p}"] = len(sel)
json.dump(rec, open(os.path.join(os.path.dirname(src), f"sq_{tag}.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(rec, indent=1, sort_keys=True))
PY
