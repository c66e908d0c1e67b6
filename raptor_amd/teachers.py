"""Teacher bank: many MLP teacher policies queried on student-visited states in one launch.

The distillation step of rl-tools/raptor (/root/reference/README.md:208-216) labels the states the student
visits with the action of the teacher that was trained for that quadrotor - about 1000 MLP teachers, one per
sampled dynamics.  ``TeacherBank`` holds such a set on the device and ``Trajectory.relabel_teachers`` evaluates
teacher ``teacher_ids[i]`` on every recorded step of env ``i``.  The teachers'
architecture is not in the reference tree [UPSTREAM-UNVERIFIED]: what is supported is any stack of dense layers
input -> w1 [-> w2 [-> w3]] -> 4 with widths that are multiples of 16 up to 128 - two hidden layers of 16 / 32 / 64 units
take register-stationary kernels in three precisions, everything else a streaming fp32 kernel - and
``TeacherBank.from_checkpoints`` loads such teachers from files in the reference's HDF5 layout (``h5:/actor/layers/*``).
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


def layers_parameter_count(in_dim, widths):
    """floats per teacher of the stack in_dim -> widths[0] -> ... -> 4: [W1 | b1 | ... | W_out | b_out]"""
    n, prev = 0, int(in_dim)
    for h in list(widths) + [4]:
        n += int(h) * prev + int(h)
        prev = int(h)
    return n


class TeacherBank:
    @classmethod
    def from_layers(cls, device, weights, in_dim, widths, hidden_activation="relu", output_activation="identity", precision="fp32"):
        """Teachers of the stack in_dim -> widths[0] -> ... -> widths[-1] -> 4 (1 to 3 hidden layers, widths multiples of 16 up
        to 128); ``weights`` [n_teachers, layers_parameter_count(in_dim, widths)].  Two hidden layers of 16 / 32 / 64 units are the
        fast family (``TeacherBank(...)``, three precisions); anything else is evaluated in fp32 by the streaming kernel."""
        widths = [int(h) for h in widths]
        if len(widths) == 2 and all(h in (16, 32, 64) for h in widths):
            return cls(device, weights, in_dim, widths[0], widths[1], hidden_activation, output_activation, precision)
        if not 1 <= len(widths) <= 3 or any(h % 16 or not 16 <= h <= 128 for h in widths):
            raise ValueError(f"unsupported teacher topology {in_dim}-{'-'.join(map(str, widths))}-4: one to three hidden layers, "
                             "widths multiples of 16 from 16 to 128")
        w = np.ascontiguousarray(weights, np.float32)
        per = layers_parameter_count(in_dim, widths)
        if w.ndim != 2 or w.shape[1] != per:
            raise ValueError(f"weights must be [n_teachers, {per}] for {in_dim}-{'-'.join(map(str, widths))}-4")
        self = cls.__new__(cls)
        self.n_teachers, self.in_dim, self.widths = int(w.shape[0]), int(in_dim), widths
        self.h1, self.h2 = widths[0], widths[1] if len(widths) > 1 else 0
        self.hidden_activation, self.output_activation = hidden_activation, output_activation
        self._device = device
        wd = np.asarray(widths, np.uint32)
        h = C.c_void_p()
        _lib.call("rq_teacher_bank_create_layers", device._h, _lib.fptr(w), self.n_teachers, self.in_dim, len(widths),
                  wd.ctypes.data_as(C.POINTER(C.c_uint32)), ACTIVATIONS[hidden_activation], ACTIVATIONS[output_activation], C.byref(h))
        self._h = h
        self._fin = weakref.finalize(self, _lib.load().rq_teacher_bank_destroy, h)
        self.set_precision(precision)
        return self

    @classmethod
    def from_checkpoints(cls, device, paths, in_dim=None, precision="fp32", group="actor"):
        """One teacher per file, each a `sequential` of `dense` layers in the reference's HDF5 layout (what
        ``extract_checkpoints.sh`` gathers, README.md:211-216; ``raptor_amd.checkpoint.load_mlp_checkpoint_h5``).  All files must
        describe the same topology: 1 to 3 hidden layers with one activation, 4 outputs; the first ``in_dim`` recorded observation
        features are the input (default: the first layer's input width, at most 22).  -> TeacherBank, teacher k = paths[k]."""
        from .checkpoint import load_mlp_checkpoint_h5
        paths = list(paths)
        if not paths:
            raise ValueError("no checkpoint files")
        blocks, topo = [], None
        for path in paths:
            layers, acts = load_mlp_checkpoint_h5(path, group)
            shape = ([lay[0].shape for lay in layers], acts)
            if topo is None:
                topo = shape
                if len(layers) < 2 or layers[-1][0].shape[0] != 4:
                    raise ValueError(f"{path}: a teacher ends in a dense layer with 4 outputs (found {shape[0]})")
                if len(set(acts[:-1])) != 1:
                    raise ValueError(f"{path}: the hidden layers use different activations {acts[:-1]}")
            elif shape != topo:
                raise ValueError(f"{path}: topology {shape} differs from the first file's {topo}")
            blocks.append(np.concatenate([np.concatenate([W.ravel(), b.ravel()]) for W, b in layers]))
        first_in = topo[0][0][1]
        in_dim = first_in if in_dim is None else int(in_dim)
        if in_dim != first_in or in_dim > 22:
            raise ValueError(f"the teachers read {first_in} inputs; the recorded observation offers its first {min(in_dim, 22)}")
        widths = [sh[0] for sh in topo[0][:-1]]
        return cls.from_layers(device, np.stack(blocks).astype(np.float32), in_dim, widths, topo[1][0], topo[1][-1], precision)

    def __init__(self, device, weights, in_dim=22, h1=64, h2=64, hidden_activation="relu",
                 output_activation="identity", precision="fp32"):
        w = np.ascontiguousarray(weights, np.float32)
        per = parameter_count(in_dim, h1, h2)
        if w.ndim != 2 or w.shape[1] != per:
            raise ValueError(f"weights must be [n_teachers, {per}] for {in_dim}-{h1}-{h2}-4")
        self.n_teachers, self.in_dim, self.h1, self.h2 = int(w.shape[0]), int(in_dim), int(h1), int(h2)
        self.widths = [int(h1), int(h2)]
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
