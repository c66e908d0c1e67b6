"""Drop-in for ``foundation_policy.Raptor`` (README.md:16-24,46-48,94,97).

    from raptor_amd.foundation_policy import Raptor
    policy = Raptor()
    policy.reset()
    action = policy.evaluate_step(observation[:, :22])     # [B,22] float32 -> [B,4]

The policy is the published RAPTOR checkpoint: Dense(22->16, ReLU) -> GRU(16) -> Dense(16->4)
(checkpoint.h:39-65,75-139,149-175; chain checkpoint.h:185), 2 084 float32 parameters shipped
as ``raptor_amd/data/raptor_policy.bin`` (extracted by tests/golden/make_golden.py).  The
output is the raw Dense output — the shipped actor has no squashing layer (checkpoint.h:170),
clipping to [-1,1] happens in the consumer (``vector.step``).  The GRU hidden state is kept
per batch element between calls; ``reset()`` restores ``initial_hidden_state``.
"""
import ctypes as C
import os
import weakref

import numpy as np

from . import _lib
from ._lib import POLICY_INPUT_DIM, POLICY_NUM_WEIGHTS, POLICY_OUTPUT_DIM, POLICY_HIDDEN_DIM

WEIGHTS_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "raptor_policy.bin")


def load_weights(path=WEIGHTS_PATH):
    w = np.fromfile(path, dtype="<f4")
    if w.size != POLICY_NUM_WEIGHTS:
        raise ValueError(f"{path}: expected {POLICY_NUM_WEIGHTS} float32 values, found {w.size}")
    return np.ascontiguousarray(w, np.float32)


_default_device = None


def _get_default_device():
    global _default_device
    if _default_device is None:
        from .l2f import Device
        _default_device = Device(0)
    return _default_device


PRECISIONS = {"fp32": _lib.POLICY_FP32, "bf16": _lib.POLICY_BF16_MFMA, "f16x2": _lib.POLICY_F16X2_MFMA}


class Raptor:
    """``precision``: "fp32" (exact-f32 MFMA, matches the reference KATs to < 1e-5) or "bf16"
    (BASELINE config 5: bf16 operands on v_mfma_f32_16x16x32_bf16, fp32 accumulate and gates;
    ~2e-2 max abs action deviation on the KATs) or "f16x2" (every operand as two f16 pieces on
    v_mfma_f32_16x16x32_f16: fp32-grade results, KATs to ~1e-6, at close to the bf16 actor's speed; not fp32
    arithmetic, so not the default)."""

    def __init__(self, device=None, weights=None, precision="fp32"):
        if precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(PRECISIONS)}")
        self._precision = precision
        self._weights = load_weights() if weights is None else np.ascontiguousarray(weights, np.float32)
        if self._weights.size != POLICY_NUM_WEIGHTS:
            raise ValueError(f"expected {POLICY_NUM_WEIGHTS} weights")
        self._device = device
        self._h = None
        self._fin = None
        self.example = None
        self.observation_spec = None

    @classmethod
    def from_checkpoint(cls, path, device=None, precision="fp32", check_observation=True):
        """Load a policy checkpoint written by rl-tools: the C++ code export (``checkpoint.h``) or its
        HDF5 twin (``checkpoint.h5``), same topology.  The embedded known-answer example, if any, is
        kept as ``policy.example``.  An HDF5 checkpoint states the observation it was trained on
        (``/actor@meta``); one that names another layout than this engine's ``observe`` assembles is
        refused (``check_observation=False`` loads it anyway); ``policy.observation_spec`` keeps the string."""
        from . import checkpoint as ck
        weights, example, meta = ck.load_checkpoint(path, with_meta=True)
        spec = ck.check_observation(meta, path) if check_observation else ck.observation_of_meta(meta)
        pol = cls(device=device, weights=weights, precision=precision)
        pol.example = example
        pol.observation_spec = spec
        return pol

    def save_checkpoint(self, path, example=None):
        """Write the policy back in an rl-tools format chosen by the extension: ``.h5`` (HDF5 layout of
        ``checkpoint.h5``) or anything else (C++ byte-array export like ``checkpoint.h``).  ``example``
        defaults to the known-answer pair the policy was loaded with."""
        from .checkpoint import write_checkpoint_h5, write_checkpoint_header
        example = self.example if example is None else example
        if str(path).endswith((".h5", ".hdf5")):
            write_checkpoint_h5(path, self._weights, example)
        else:
            write_checkpoint_header(path, self._weights, example)

    # the C object is created on first use so that ``Raptor()`` itself needs no device argument
    def _handle(self, device=None):
        if self._h is None:
            if self._device is None:
                self._device = device if device is not None else _get_default_device()
            h = C.c_void_p()
            _lib.call("rq_policy_create", self._device._h, _lib.fptr(self._weights), self._weights.size, C.byref(h))
            self._h = h
            self._fin = weakref.finalize(self, _lib.load().rq_policy_destroy, h)
            _lib.call("rq_policy_set_precision", h, PRECISIONS[self._precision])
        elif device is not None and device is not self._device:
            raise _lib.RaptorQuadError(-5, "policy was created on another device")
        return self._h

    @property
    def weights(self):
        return self._weights

    @property
    def precision(self):
        return self._precision

    def set_precision(self, precision):
        if precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(PRECISIONS)}")
        self._precision = precision
        if self._h is not None:
            _lib.call("rq_policy_set_precision", self._h, PRECISIONS[precision])

    def set_standardize(self, mean=None, std=None):
        """Optional Standardize input stage (x - mean) / std (not part of the shipped checkpoint;
        folded into layer_0 on the host).  ``None`` disables."""
        if mean is None:
            _lib.call("rq_policy_set_standardize", self._handle(), None, None)
        else:
            m, s_ = np.ascontiguousarray(mean, np.float32), np.ascontiguousarray(std, np.float32)
            assert m.shape == (POLICY_INPUT_DIM,) and s_.shape == (POLICY_INPUT_DIM,)
            _lib.call("rq_policy_set_standardize", self._handle(), _lib.fptr(m), _lib.fptr(s_))

    def set_squash(self, enable):
        """Optional tanh output stage (SampleAndSquash evaluated deterministically; not in the
        shipped checkpoint)."""
        _lib.call("rq_policy_set_squash", self._handle(), 1 if enable else 0)

    def set_sample_and_squash(self, mode, log_std_weights=None, log_std_bias=None, seed=0):
        """The SampleAndSquash output layer (rl-tools ``nn/layers/sample_and_squash``, README.md:116; not in the shipped
        checkpoint, semantics [UPSTREAM-UNVERIFIED]): the last dense layer has 8 outputs [mean | log_std]; the mean rows
        are the policy's layer_2, the log-std rows are given here (``log_std_weights`` [4, 16] or None for a state-
        independent log-std, ``log_std_bias`` [4]).  ``mode``: "off" (raw output), "mean" (tanh(mean)) or "sample"
        (tanh(mean + exp(clamp(log_std, -20, 2)) * eps), eps from the Philox stream keyed by ``seed``)."""
        m = {"off": 0, "mean": 1, "sample": 2}[mode]
        w = None if log_std_weights is None else np.ascontiguousarray(log_std_weights, np.float32)
        b = None if log_std_bias is None else np.ascontiguousarray(log_std_bias, np.float32)
        assert w is None or w.shape == (POLICY_OUTPUT_DIM, POLICY_HIDDEN_DIM)
        assert b is None or b.shape == (POLICY_OUTPUT_DIM,)
        _lib.call("rq_policy_set_sample_and_squash", self._handle(), m, None if w is None else _lib.fptr(w),
                  None if b is None else _lib.fptr(b), int(seed))

    def reset(self):
        """README.md:21,94 — hidden state <- initial_hidden_state (checkpoint.h:123, zeros)."""
        _lib.call("rq_policy_reset", self._handle())

    def evaluate_step(self, observation):
        """README.md:24,97 — one recurrent step; ``observation`` [B, >=22] -> action [B,4]."""
        fast = _lib.fast
        if type(observation) is not np.ndarray or observation.dtype != np.float32:
            observation = np.asarray(observation, np.float32)      # README.md:23 builds a float64 array; lists work too
        if fast is not None and self._h is not None and observation.ndim == 2:
            act = np.empty((observation.shape[0], POLICY_OUTPUT_DIM), np.float32)      # (no ctypes in between: csrc/rq_pyfast.c)
            status = fast.evaluate_step(_lib.fn_addr("rq_policy_evaluate_step"), self._h, observation, act, POLICY_INPUT_DIM)
            if status == 0:
                return act
            if status != 1:
                _lib.check(status)
        obs = observation
        if obs.ndim != 2 or obs.shape[1] < POLICY_INPUT_DIM:
            raise ValueError("observation must be [batch, >=22]")
        if not obs.flags.c_contiguous:
            # rows may be strided (e.g. observation[:, :22] of a wider array): keep the row stride
            if obs.strides[1] == 4 and obs.strides[0] % 4 == 0 and obs.strides[0] >= 4 * POLICY_INPUT_DIM:
                stride = obs.strides[0] // 4
            else:
                obs = np.ascontiguousarray(obs)
                stride = obs.shape[1]
        else:
            stride = obs.shape[1]
        batch = obs.shape[0]
        act = np.empty((batch, POLICY_OUTPUT_DIM), np.float32)
        _lib.call("rq_policy_evaluate_step", self._handle(), None, _lib.fptr(obs),
                  batch, stride, _lib.fptr(act))
        return act

    def evaluate_sequence(self, observation):
        """rl-tools' evaluate on a sequence tensor (the layout of the checkpoint's example, checkpoint.h:197-215):
        ``observation`` [T, B, >=22] -> action [T, B, 4], the same result as T calls of ``evaluate_step`` (hidden
        state carried from the current one and kept afterwards) in ONE kernel launch.  NumPy in -> NumPy out;
        a CUDA/HIP torch tensor in (contiguous float32) -> torch tensor out, nothing leaves the device."""
        if hasattr(observation, "data_ptr"):                # torch tensor on the device
            import torch
            obs = observation
            if obs.dim() != 3 or obs.shape[2] < POLICY_INPUT_DIM or obs.dtype != torch.float32 or not obs.is_cuda \
                    or not obs.is_contiguous():
                raise ValueError("observation must be a contiguous float32 device tensor [T, B, >=22]")
            act = torch.empty((obs.shape[0], obs.shape[1], POLICY_OUTPUT_DIM), dtype=torch.float32, device=obs.device)
            torch.cuda.current_stream(obs.device).synchronize()       # the engine runs on its own stream
            _lib.call("rq_policy_evaluate_sequence", self._handle(), C.c_void_p(obs.data_ptr()), obs.shape[0],
                      obs.shape[1], obs.shape[2], C.c_void_p(act.data_ptr()), 1)
            return act
        obs = np.ascontiguousarray(observation, np.float32)
        if obs.ndim != 3 or obs.shape[2] < POLICY_INPUT_DIM:
            raise ValueError("observation must be [T, B, >=22]")
        act = np.empty((obs.shape[0], obs.shape[1], POLICY_OUTPUT_DIM), np.float32)
        _lib.call("rq_policy_evaluate_sequence", self._handle(), obs.ctypes.data, obs.shape[0],
                  obs.shape[1], obs.shape[2], act.ctypes.data, 0)
        return act

    def evaluate_step_device(self, env):
        """Device-resident variant: reads the env's observation buffer (``observe(..., None, ...)``)
        and writes its action buffer (consumed by ``step(..., action=None, ...)``)."""
        _lib.call("rq_policy_evaluate_step", self._handle(env._device), env._require("environment"), None,
                  env.N_ENVIRONMENTS, 0, None)

    def hidden_state(self, batch):
        out = np.empty((batch, POLICY_HIDDEN_DIM), np.float32)
        _lib.call("rq_policy_get_hidden", self._handle(), _lib.fptr(out), batch)
        return out

    def set_hidden_state(self, hidden):
        h = np.ascontiguousarray(hidden, np.float32)
        assert h.ndim == 2 and h.shape[1] == POLICY_HIDDEN_DIM
        _lib.call("rq_policy_set_hidden", self._handle(), _lib.fptr(h), h.shape[0])

    def selftest(self, input_seq, expected, tolerance=1e-5):
        """Known-answer self-test (the embedded backend's boot test, README.md:136-139).
        ``input_seq`` [T,B,22], ``expected`` [T,B,4]; returns max |out - expected|."""
        x = np.ascontiguousarray(input_seq, np.float32)
        y = np.ascontiguousarray(expected, np.float32)
        err = C.c_float()
        _lib.call("rq_policy_selftest", self._handle(), _lib.fptr(x), _lib.fptr(y), x.shape[0], x.shape[1],
                  float(tolerance), C.byref(err))
        return float(err.value)
