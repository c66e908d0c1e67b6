#!/bin/bash
# The experiment builds of round 5's hunt for the two-waves-per-SIMD bf16 build's run-to-run differences (profiles/r05_bf16_two_wave_hunt.md):
# scratch/variants/libraptor_quad_<X>.so, each the whole library with the round-4 routing of bf16 batches > 65 536 envs to ActorBF16Lean
# (-DRQ_BF16_FUSED_LEAN).  On the GPU box:  RAPTOR_QUAD_LIB=scratch/variants/libraptor_quad_A.so python tools/hazard_diag.py --ref ...
#   C  default instruction scheduler                      (sound: the reference results come from here)
#   A  -amdgpu-sched-strategy=max-ilp                      (differs from run to run)
#   B  A without -amdgpu-mfma-vgpr-form=1                  (still differs: not that flag)
#   D  A + -amdgpu-waitcnt-forcezero                       (differs more: every wait hipcc emits already waits for everything)
#   F  A + the packed-math inline asm as plain C++         (still differs: not the asm statements' hidden hazards)
#   G  A + two wait states on either side inside every asm (still differs)
#   H  A + scalar spills to memory instead of VGPR lanes   (still differs)
#   K / L  F / A with v8 .. v247 zeroed at kernel entry    (still differs: not a read of an uninitialised register)
#   FH F + -DRQ_DEBUG_HWID (every wave records HW_REG_HW_ID: tools/hazard_diag.py --hwid)
# Listing-level experiments on top of F / A: tools/asm_edit.py, tools/asm_instrument.py, tools/opsel_rewrite.py + tools/asm_build.sh.
# RQ_DEBUG_FUSED_LDS=<bytes> (any of these builds) gives every workgroup that much LDS and so bounds the waves per CU.
set -e
cd "$(dirname "$0")/.."
mkdir -p scratch/variants
M="-mllvm -amdgpu-sched-strategy=max-ilp"
build() { name=$1; shift; ( RQ_NO_OPSEL_REWRITE=1 "$@" python -m raptor_amd.build --variant "$name" --patch tools/variants/hunt_experiments.patch -DRQ_BF16_FUSED_LEAN ${FLAGS} > "scratch/variants/$name.log" 2>&1 && echo "built $name" || { echo "FAILED $name"; tail -n 5 "scratch/variants/$name.log"; } ) & }
FLAGS="" build C env
FLAGS="$M" build A env
FLAGS="$M" build B env RQ_NO_MFMA_VGPR_FORM=1
FLAGS="$M -mllvm -amdgpu-waitcnt-forcezero=1" build D env
wait
FLAGS="$M -DRQ_PK_PLAIN_C" build F env
FLAGS="$M -DRQ_PK_PADDED_ASM" build G env
FLAGS="$M -mllvm -amdgpu-spill-sgpr-to-vgpr=0" build H env
FLAGS="$M -DRQ_PK_PLAIN_C -DRQ_DEBUG_ZERO_VGPRS" build K env
wait
FLAGS="$M -DRQ_DEBUG_ZERO_VGPRS" build L env
FLAGS="$M -DRQ_PK_PLAIN_C -DRQ_DEBUG_HWID" build FH env
wait
for t in opsel_repro hazard_probe hazard_probe2 hazard_probe3 hazard_probe4 hazard_probe5 hazard_probe6 hazard_probe7 scratch_probe; do hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/$t.hip -o tools/$t 2>/dev/null && echo "built tools/$t"; done
