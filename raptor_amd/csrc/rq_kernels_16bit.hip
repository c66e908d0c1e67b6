// rq_kernels_16bit.hip - the fused rollout kernel instantiated for the 16-bit actors (bf16 operands: BASELINE config 5;
// split-f16 operands), in a translation unit of its own because it wants another instruction scheduler than the
// hand-ordered fp32 build: see rq_rollout.hpp (and raptor_amd/build.py SOURCE_FLAGS for what was tried and taken back).
#include "rq_rollout.hpp"

namespace rq {

hipError_t launch_rollout_fused_16bit(hipStream_t s, const FusedArgs& a, bool noise, bool ar, int precision) {
    // The bf16 actor runs its one-wave-per-SIMD (512-register) build at EVERY batch size (round 4): the two-waves-per-SIMD build
    // spills (124 dwords of scratch per lane) and was 4 % slower per env at 262 144 envs than the one-wave build is at 65 536
    // (5.69 us against 4 x 1.36 us per step); only the SampleAndSquash stage still rides on the 256-register build.
    if (a.sas.mode != RQ_SAS_OFF) {
        if (precision == RQ_POLICY_F16X2_MFMA) launch_fused_actor<true, ActorF16X2>(s, a, noise, ar);
        else                                   launch_fused_actor<true, ActorBF16Lean>(s, a, noise, ar);
    } else if (precision == RQ_POLICY_F16X2_MFMA) {
        launch_fused_actor<false, ActorF16X2>(s, a, noise, ar);
    } else {
        launch_fused_actor<false, ActorBF16>(s, a, noise, ar);
    }
    return hipGetLastError();
}

}  // namespace rq
