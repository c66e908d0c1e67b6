"""Teacher bank: many MLP teacher policies queried on student-visited states in one launch.

The distillation step of rl-tools/raptor (/root/reference/README.md:208-216) labels the states the student
visits with the action of the teacher that was trained for that quadrotor - about 1000 MLP teachers, one per
sampled dynamics.  ``TeacherBank`` holds such a set on the device and ``Trajectory.relabel_teachers`` evaluates
teacher ``teacher_ids[i]`` on every recorded step of env ``i``.  The teachers'
architecture is not in the reference tree: this is the plain MLP family input -> h1 -> h2 -> 4
[UPSTREAM-UNVERIFIED], h1, h2 in {16, 32, 64}.
"""
import ctypes as C
import weakref

import numpy as np

from . import _lib

ACTIVATIONS = {"identity": _lib.ACT_IDENTITY, "relu": _lib.ACT_RELU, "tanh": _lib.ACT_TANH}
PRECISIONS = {"fp32": _lib.POLICY_FP32, "bf16": _lib.POLICY_BF16_MFMA, "f16x2": _lib.POLICY_F16X2_MFMA}


def parameter_count(in_dim, h1, h2):
    """floats per teacher: W1 [h1, in_dim], b1 [h1], W2 [h2, h1], b2 [h2], W3 [4, h2], b3 [4]"""
    return h1 * in_dim + h1 + h2 * h1 + h2 + 4 * h2 + 4


def flatten_teacher(W1, b1, W2, b2, W3, b3):
    """One teacher's layers (matrices [out, in], as rl-tools dense layers store them) -> its flat block."""
    parts = [np.asarray(a, np.float32).ravel() for a in (W1, b1, W2, b2, W3, b3)]
    return np.concatenate(parts)


class TeacherBank:
    def __init__(self, device, weights, in_dim=22, h1=64, h2=64, hidden_activation="relu",
                 output_activation="identity", precision="fp32"):
        w = np.ascontiguousarray(weights, np.float32)
        per = parameter_count(in_dim, h1, h2)
        if w.ndim != 2 or w.shape[1] != per:
            raise ValueError(f"weights must be [n_teachers, {per}] for {in_dim}-{h1}-{h2}-4")
        self.n_teachers, self.in_dim, self.h1, self.h2 = int(w.shape[0]), int(in_dim), int(h1), int(h2)
        self.hidden_activation, self.output_activation = hidden_activation, output_activation
        self._device = device
        h = C.c_void_p()
        _lib.call("rq_teacher_bank_create", device._h, _lib.fptr(w), self.n_teachers, self.in_dim, self.h1, self.h2,
                  ACTIVATIONS[hidden_activation], ACTIVATIONS[output_activation], C.byref(h))
        self._h = h
        self._fin = weakref.finalize(self, _lib.load().rq_teacher_bank_destroy, h)
        self.set_precision(precision)

    def set_precision(self, precision):
        if precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(PRECISIONS)}")
        self.precision = precision
        _lib.call("rq_teacher_bank_set_precision", self._h, PRECISIONS[precision])


def balanced_teacher_assignment(n_envs, n_teachers):
    """teacher id of every env (uint32 [n_envs], contiguous groups) with WHOLE 16-env tiles per teacher.

    The relabel kernels give one wave a tile of up to 16 envs of ONE teacher; a teacher with 66 envs (65 536 envs over
    the reference's 1 000 teachers, README.md:207-216, split evenly) fills 5 tiles of which the last holds 2 envs: 18 %
    of the matrix work is padding (0.54 of the f32 MFMA peak against 0.66 at 1 024 teachers, round 2).  Here the
    n_envs / 16 tiles are dealt out instead: 65 536 / 1 000 -> 96 teachers x 80 envs + 904 teachers x 64 envs, no
    padding at all.  Which env is flown by which teacher's quadrotor is the caller's choice when it samples the
    parameters (one teacher = one set of dynamics); this only fixes the group sizes."""
    n_envs, n_teachers = int(n_envs), int(n_teachers)
    if n_envs <= 0 or n_teachers <= 0:
        raise ValueError("n_envs and n_teachers must be positive")
    tiles = (n_envs + 15) // 16
    base, extra = divmod(tiles, n_teachers)
    per_teacher = np.full(n_teachers, base, np.int64)
    per_teacher[:extra] += 1                                  # tiles of each teacher
    ids = np.repeat(np.arange(n_teachers, dtype=np.uint32), per_teacher * 16)[:n_envs]
    return np.ascontiguousarray(ids)
