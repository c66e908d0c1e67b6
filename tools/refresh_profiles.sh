#!/bin/bash
# Run ON THE GPU BOX from the repo root:  bash tools/refresh_profiles.sh r02
# Everything profiles/<tag>_* is made from, in one call: the driver's bench command un-profiled, then under rocprofv3
# (kernel trace, FETCH_SIZE / WRITE_SIZE passes), the SQ counters of the fused kernel, the side configurations, the
# teacher bank, and the side rates (recorded rollout, evaluate_sequence).  Results -> gpurun_out/profiles_<tag>/.
set -u
TAG=${1:-r02}
R=$PWD
DST=$R/gpurun_out/profiles_$TAG
mkdir -p $DST
python bench.py --gpus 1 --steps 20 --warmup 5 > $DST/${TAG}_bench_driver_cmd.json 2> $DST/driver_cmd.err; echo "driver cmd rc=$?"
bash tools/profile_round.sh $TAG > $DST/profile_round.log 2>&1; echo "profile_round rc=$?"
bash tools/sq_profile.sh $TAG > $DST/sq_profile.log 2>&1; echo "sq_profile rc=$?"
bash tools/profile_extras.sh $TAG > $DST/profile_extras.log 2>&1; echo "profile_extras rc=$?"
{
  python tools/record_rate.py --precision fp32
  python tools/record_rate.py --precision bf16
  python tools/sequence_rate.py --precision fp32
  python tools/sequence_rate.py --precision bf16
  python tools/kernel_time.py
} > $DST/${TAG}_side_rates.txt 2>&1
echo "side rates rc=$?"
python tools/determinism_soak.py --steps 1500 > $DST/${TAG}_determinism_soak.txt 2>&1; echo "soak rc=$?"
ls -la $DST
