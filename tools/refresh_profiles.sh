#!/bin/bash
# Run ON THE GPU BOX from the repo root:  bash tools/refresh_profiles.sh r02
# Everything profiles/<tag>_* is made from, in one call.  Order matters since round 6: bench.py takes its headline fraction from a
# committed rocprofv3 trace only when that trace is of the very build it has loaded (library sha256), so the trace and the counters are
# taken FIRST and put where bench.py looks (profiles/ of this copy of the tree), and the un-profiled record of the driver's command is
# made after them, with the same build.  Then the SQ counters of the fused kernel, the side configurations, the teacher bank and its
# traffic, the side rates, the soaks.  Results -> gpurun_out/profiles_<tag>/.
set -u
TAG=${1:-r02}
R=$PWD
DST=$R/gpurun_out/profiles_$TAG
mkdir -p $DST
bash tools/profile_round.sh $TAG > $DST/profile_round.log 2>&1; echo "profile_round rc=$?"
cp $DST/${TAG}_fused_launch_stats.json $DST/${TAG}_pmc.json $R/profiles/ 2>/dev/null
bash tools/sq_profile.sh $TAG > $DST/sq_profile.log 2>&1; echo "sq_profile rc=$?"
cp $DST/${TAG}_sq_counters.json $R/profiles/ 2>/dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 > $DST/${TAG}_bench_driver_cmd.json 2> $DST/driver_cmd.err; echo "driver cmd rc=$?"
bash tools/profile_extras.sh $TAG > $DST/profile_extras.log 2>&1; echo "profile_extras rc=$?"
bash tools/teacher_traffic.sh $TAG > $DST/teacher_traffic.log 2>&1; echo "teacher_traffic rc=$?"
{
  python tools/record_rate.py --precision fp32
  python tools/record_rate.py --precision bf16
  python tools/sequence_rate.py --precision fp32
  python tools/sequence_rate.py --precision bf16
  python tools/sequence_rate.py --precision fp32 --steps 2000
  python tools/kernel_time.py
  python tools/readme_loop_split.py
  RQ_NO_RESIDENT=1 python tools/readme_loop_split.py
  RQ_RESIDENT_TIMING=1 python tools/resident_check.py --envs 8 --iters 2000
  python tools/resident_check.py --envs 100 --iters 1000
  ./tools/bar_probe
} > $DST/${TAG}_side_rates.txt 2>&1
echo "side rates rc=$?"
python tools/determinism_soak.py --steps 1500 > $DST/${TAG}_determinism_soak.txt 2>&1; echo "soak rc=$?"
python tools/foreign_soak.py --reps 30 --json $DST/${TAG}_foreign_soak.json > $DST/${TAG}_foreign_soak.txt 2>&1; echo "foreign soak rc=$?"
python tools/resident_soak.py --iters 100000 --json $DST/${TAG}_resident_soak.json > $DST/resident_soak.log 2>&1; echo "resident soak rc=$?"
python tools/cross_stream_soak.py --aggressor f16x2 --reps 30 > $DST/${TAG}_cross_stream_soak.txt 2>&1; echo "cross-stream soak rc=$?"
ls -la $DST
