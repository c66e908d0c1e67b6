"""raptor_amd — MI355X-native vectorised quadrotor rollout engine.

Drop-in for the rollout path of rl-tools/raptor: ``raptor_amd.l2f`` mirrors the ``l2f``
vector API (initialize_environment, sample_initial_parameters, sample_initial_state, observe,
step) and ``raptor_amd.foundation_policy.Raptor`` mirrors ``foundation_policy.Raptor``
(reset, evaluate_step).  All compute runs in libraptor_quad.so (hand-written HIP for gfx950
behind the C ABI of include/raptor_quad.h); this package is the thin binding.
"""
from ._lib import RaptorQuadError, EnvConfig, LIB_PATH  # noqa: F401

__version__ = "0.1.0"
