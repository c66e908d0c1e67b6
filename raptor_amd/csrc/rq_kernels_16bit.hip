// rq_kernels_16bit.hip - the fused rollout kernel instantiated for the 16-bit actors (bf16 operands: BASELINE config 5;
// split-f16 operands), in a translation unit of its own because it wants another instruction scheduler than the
// hand-ordered fp32 build: see rq_rollout.hpp (and raptor_amd/build.py SOURCE_FLAGS for what was tried and taken back).
#include "rq_rollout.hpp"

namespace rq {

hipError_t launch_rollout_fused_16bit(hipStream_t s, const FusedArgs& a, bool noise, bool ar, int precision) {
    // one wave per SIMD up to 65 536 envs, the two-waves-per-SIMD (256-register) build beyond; the SampleAndSquash
    // stage rides on the 256-register bf16 build only
    const bool lean = a.b.n > 65536u;
    if (a.sas.mode != RQ_SAS_OFF) {
        if (precision == RQ_POLICY_F16X2_MFMA) launch_fused_actor<true, ActorF16X2>(s, a, noise, ar);
        else                                   launch_fused_actor<true, ActorBF16Lean>(s, a, noise, ar);
    } else if (precision == RQ_POLICY_F16X2_MFMA) {
        launch_fused_actor<false, ActorF16X2>(s, a, noise, ar);
    } else if (lean) {
        launch_fused_actor<false, ActorBF16Lean>(s, a, noise, ar);
    } else {
        launch_fused_actor<false, ActorBF16>(s, a, noise, ar);
    }
    return hipGetLastError();
}

}  // namespace rq
