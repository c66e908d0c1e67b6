#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root:  bash tools/profile_round.sh r02
# Collects, for the bench command, (1) rocprofv3 --kernel-trace --stats and (2) two separate PMC
# passes (FETCH_SIZE, WRITE_SIZE: they cannot share a pass, MI355X_MICROARCH.md "rocprofv3 PMC slots"),
# then writes the per-kernel summary the repo commits under profiles/.
set -u
TAG=${1:-r01}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# the command the driver runs at round end, unless one is given: bash tools/profile_round.sh r02 "--steps 10000 --warmup 5000"
ARGS=${2:---gpus 1 --steps 20 --warmup 5}
BENCH="python $R/bench.py $ARGS --no-cpu-baseline"
echo "python bench.py $ARGS --no-cpu-baseline" > $OUT/command.txt
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $BENCH > $OUT/bench_trace.json 2> $OUT/trace.err
echo "trace rc=$?"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o bench -- $BENCH > $OUT/bench_fetch.json 2> $OUT/fetch.err
echo "fetch rc=$?"
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o bench -- $BENCH > $OUT/bench_write.json 2> $OUT/write.err
echo "write rc=$?"
cd $R
python tools/summarize_profiles.py $OUT $TAG
