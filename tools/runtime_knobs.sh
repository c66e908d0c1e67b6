#!/bin/bash
# Run ON THE GPU BOX from the repo root:  bash tools/runtime_knobs.sh > gpurun_out/runtime_knobs.txt
# The 20-step timed region of bench.py (launch call + kernel + torch.cuda.synchronize) under ROCm runtime settings that
# act on the way a dispatch is handed to the GPU and on how its completion reaches the host.  One process per setting
# (the runtime reads them once, at initialisation).
set -u
run() {
  echo "== $*"
  env "$@" timeout 300 python tools/region_split.py --variants "torch only" --reps 3000 2>&1 | grep -v "amdgpu.ids" | grep -E "region|kernel|idle"
}
run RQ_NONE=1
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run HSA_ENABLE_INTERRUPT=0
run ROC_ACTIVE_WAIT_TIMEOUT=200
run ROC_SYSTEM_SCOPE_SIGNAL=0
run AMD_DIRECT_DISPATCH=0
run ROC_SKIP_KERNEL_ARG_COPY=1
run GPU_MAX_HW_QUEUES=1
run HSA_ENABLE_INTERRUPT=0 HIP_FORCE_DEV_KERNARG=1 ROC_ACTIVE_WAIT_TIMEOUT=200
