"""Drop-in for the ``l2f`` Python surface used by the rollout loop of rl-tools/raptor.

The names, argument order and in-place semantics follow /root/reference/README.md:41-101:

    import raptor_amd.l2f as l2f
    from raptor_amd.l2f import vector8 as vector      # or: vector = l2f.vector(65536)
    device = l2f.Device(); rng = vector.VectorRng(); env = vector.VectorEnvironment()
    params = vector.VectorParameters(); state = vector.VectorState(); next_state = vector.VectorState()
    observation = np.zeros((env.N_ENVIRONMENTS, env.OBSERVATION_DIM), dtype=np.float32)
    vector.initialize_rng(device, rng, 0)
    vector.initialize_environment(device, env)
    vector.sample_initial_parameters(device, env, params, rng)
    vector.sample_initial_state(device, env, params, state, rng)
    vector.observe(device, env, params, state, observation, rng)
    dts = vector.step(device, env, params, state, action, next_state, rng)
    state.assign(next_state)

Differences that are deliberate (DESIGN.md "Boundary"):
  * the batch size is a runtime value: ``l2f.vector(N)`` builds the module-like object the
    reference pre-compiles as ``vector1 ... vectorN``; ``vector8`` is ``vector(8)``;
  * passing ``None`` for ``observation`` / ``action`` keeps the data in HBM (zero PCIe
    traffic); ``vector.rollout`` runs the whole loop body on the device;
  * everything executes on an MI355X through libraptor_quad.so — no CPU path exists here.

The UI / JSON message helpers of l2f (``set_ui_message``, ``set_parameters_message``,
``set_state_action_message``, README.md:63-92) are emitted by the host side here (plain JSON from device
snapshots); their schema beyond ``namespace`` / ``channel`` / ``data[i]`` is not in the reference tree
[UPSTREAM-UNVERIFIED], see the functions.
"""
import ctypes as C
import json
import weakref

import numpy as np

from . import _lib
from ._lib import (ACTION_DIM, OBSERVATION_DIM, PARAM_DIM, STATE_DIM, ROLLOUT_AUTORESET, ROLLOUT_CHAINED,
                   ROLLOUT_FUSED, EnvConfig, RaptorQuadError)

__all__ = ["Device", "UI", "vector", "vector8", "EnvConfig", "RaptorQuadError"]


class UI:
    """``l2f.UI()`` (README.md:52): carries the namespace the ui-server hands out in its handshake
    (``ui.ns = handshake["data"]["namespace"]``, README.md:82-85)."""

    def __init__(self, ns=""):
        self.ns = ns


class Device:
    """``l2f.Device()`` (README.md:49): one HIP device + one stream."""

    def __init__(self, ordinal=0):
        h = C.c_void_p()
        _lib.call("rq_device_create", int(ordinal), C.byref(h))
        self._h = h
        self.ordinal = int(ordinal)
        self._fin = weakref.finalize(self, _lib.load().rq_device_destroy, h)

    def synchronize(self):
        _lib.call("rq_device_synchronize", self._h)

    def timer_start(self):
        """HIP-event stopwatch on this device's stream."""
        _lib.call("rq_device_timer_start", self._h)

    def timer_stop(self):
        ms = C.c_float()
        _lib.call("rq_device_timer_stop", self._h, C.byref(ms))
        return float(ms.value)

    def set_rollout_timing(self, enable):
        """Kernel-level timing of fused rollouts: every wave records the wall-clock ticks at which it came in and went out."""
        _lib.call("rq_device_set_rollout_timing", self._h, 1 if enable else 0)

    def last_rollout_ms(self):
        """Duration (ms) of the most recent fused rollout kernel launched while ``set_rollout_timing(True)``."""
        ms = C.c_float()
        _lib.call("rq_device_last_rollout_ms", self._h, C.byref(ms))
        return float(ms.value)

    def last_rollout_waves(self):
        """(t_in, t_out, xcd, t_first_step, t_last_step_done) per wave of the most recent timed fused rollout; 100 MHz
        ticks, comparable within a die."""
        n = C.c_uint32()
        _lib.call("rq_device_last_rollout_waves", self._h, None, 0, C.byref(n))
        rec = np.zeros((n.value, 4), np.uint64)
        _lib.call("rq_device_last_rollout_waves", self._h, rec.ctypes.data, n.value, C.byref(n))
        mask = np.uint64(0x0FFFFFFFFFFFFFFF)
        return (rec[:, 0] & mask, rec[:, 1] & mask, (rec[:, 1] >> np.uint64(60)).astype(np.int64) & 7,
                rec[:, 2] & mask, rec[:, 3] & mask)

    def last_rollout_clock_ghz(self):
        """Core clock (GHz) the most recent timed fused rollout ran its steps at (median wave: shader-clock cycles over
        constant-rate ticks between its first step's start and its last step's end)."""
        ghz = C.c_float()
        _lib.call("rq_device_last_rollout_clock", self._h, C.byref(ghz))
        return float(ghz.value)

    def launch_floor(self, n, reps=200):
        """Average us per launch of back-to-back near-empty kernels on an n-thread grid (diagnostic)."""
        us = C.c_float()
        _lib.call("rq_device_launch_floor", self._h, int(n), int(reps), C.byref(us))
        return float(us.value)

    def set_speculation(self, enable):
        """The small-batch loop's speculative policy step (rq_device_set_speculation, include/raptor_quad.h): off / on
        for this device.  Left on, the device suspends it by itself after four speculated steps in a row nobody took."""
        _lib.call("rq_device_set_speculation", self._h, 1 if enable else 0)

    def speculation(self):
        """-> {"enabled", "suspended", "consecutive_misses"}"""
        e, s, m = C.c_int(), C.c_int(), C.c_uint32()
        _lib.call("rq_device_get_speculation", self._h, C.byref(e), C.byref(s), C.byref(m))
        return {"enabled": bool(e.value), "suspended": bool(s.value), "consecutive_misses": int(m.value)}

    def set_resident(self, enable):
        """The small-batch loop's resident executor (rq_device_set_resident, include/raptor_quad.h): off / on for this device."""
        _lib.call("rq_device_set_resident", self._h, 1 if enable else 0)

    def resident(self):
        """-> {"enabled", "running", "starts", "commands", "replays"}: is a resident kernel on the device now; kernels started,
        commands posted to them, commands replayed as launches since the device was created."""
        e, r = C.c_int(), C.c_int()
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _lib.call("rq_device_get_resident", self._h, C.byref(e), C.byref(r), C.byref(a), C.byref(b), C.byref(c))
        return {"enabled": bool(e.value), "running": bool(r.value), "starts": int(a.value), "commands": int(b.value), "replays": int(c.value)}

    def resident_timing_us(self):
        """Device-side timeline of the last command the resident kernel finished, in microseconds from the moment it saw the command:
        {"rows_read", "stepped", "first_flag", "acted", "done"} (rq_device_get_resident_timing)."""
        t = (C.c_uint64 * 6)()
        _lib.call("rq_device_get_resident_timing", self._h, t)
        return {k: (int(t[i + 1]) - int(t[0])) / 100.0 for i, k in enumerate(("rows_read", "stepped", "first_flag", "acted", "done"))}

    @property
    def stream(self):
        s = C.c_void_p()
        _lib.call("rq_device_stream", self._h, C.byref(s))
        return s.value

    @staticmethod
    def count():
        n = C.c_int()
        lib = _lib.load()
        lib.rq_device_count(C.byref(n))
        return n.value


class _Handle:
    """Lazily created C object (the reference's constructors take no device argument)."""
    _destroy = None

    def __init__(self):
        self._h = None
        self._fin = None

    def _adopt(self, h):
        self._h = h
        self._fin = weakref.finalize(self, getattr(_lib.load(), self._destroy), h)

    def _require(self, what):
        if self._h is None:
            raise RaptorQuadError(-6, f"{what} is not initialised yet")
        return self._h


class _DeviceArray:
    """A device buffer of the engine described by ``__cuda_array_interface__`` (version 3): PyTorch (and CuPy,
    Numba) wrap it WITHOUT a copy - ``torch.as_tensor(x, device="cuda")`` - so a learner reads rollouts where
    the engine wrote them.  ``owner`` keeps the C object alive as long as a tensor made from this exists."""

    def __init__(self, ptr, shape, typestr, owner):
        self._owner = owner
        self.__cuda_array_interface__ = {"shape": tuple(int(d) for d in shape), "typestr": typestr,
                                         "data": (int(ptr), False), "version": 3, "strides": None}


def _ordinal_of(owner):
    """HIP device ordinal of the engine object a buffer belongs to (env, container or trajectory)."""
    for obj in (owner, getattr(owner, "_env", None)):
        dev = getattr(obj, "_device", None)
        if dev is not None:
            return dev.ordinal
    raise RaptorQuadError(-6, "the object is not bound to a device yet")


def _as_torch(ptr, shape, typestr, owner):
    """Zero-copy torch view of an engine buffer ON THE ENGINE'S DEVICE (not torch's current one: as_tensor with a
    bare "cuda" would silently copy to the current device and the view would be a detached copy)."""
    import torch
    t = torch.as_tensor(_DeviceArray(ptr, shape, typestr, owner), device=f"cuda:{_ordinal_of(owner)}")
    if t.data_ptr() != int(ptr):
        raise RaptorQuadError(-3, "torch copied the buffer instead of wrapping it (device mismatch)")
    return t


class _StateView:
    """One element of ``VectorState.states`` (README.md:73-75): a host-side snapshot."""
    __slots__ = ("_row",)

    def __init__(self, row):
        self._row = row

    position = property(lambda s: s._row[0:3])
    orientation = property(lambda s: s._row[3:7])
    linear_velocity = property(lambda s: s._row[7:10])
    angular_velocity = property(lambda s: s._row[10:13])
    rpm = property(lambda s: s._row[13:17])
    last_action = property(lambda s: s._row[17:21])
    force = property(lambda s: s._row[21:24])
    torque = property(lambda s: s._row[24:27])


class VectorModule:
    """What ``from l2f import vector8 as vector`` gives, for a runtime batch size.

    ``global_env_offset``: global id of env 0 when one logical batch is sharded over several
    devices/processes — the RNG is keyed by global id, so results do not depend on sharding.
    """

    def __init__(self, n_environments, global_env_offset=0):
        if n_environments <= 0:
            raise ValueError("n_environments must be positive")
        self.N_ENVIRONMENTS = int(n_environments)
        self.OBSERVATION_DIM = OBSERVATION_DIM
        self.ACTION_DIM = ACTION_DIM
        self.global_env_offset = int(global_env_offset)
        mod = self

        class VectorRng(_Handle):
            _destroy = "rq_rng_destroy"

            def _ensure(self, device):
                if self._h is None:
                    h = C.c_void_p()
                    _lib.call("rq_rng_create", device._h, C.byref(h))
                    self._adopt(h)
                    self._device = device
                return self._h

            @property
            def epoch(self):
                e = C.c_uint32()
                _lib.call("rq_rng_get", self._require("rng"), None, C.byref(e))
                return e.value

            @property
            def seed(self):
                s = C.c_uint64()
                _lib.call("rq_rng_get", self._require("rng"), C.byref(s), None)
                return s.value

        class VectorEnvironment(_Handle):
            _destroy = "rq_env_destroy"
            N_ENVIRONMENTS = mod.N_ENVIRONMENTS
            OBSERVATION_DIM = OBSERVATION_DIM
            ACTION_DIM = ACTION_DIM

            def _ensure(self, device):
                if self._h is None:
                    h = C.c_void_p()
                    _lib.call("rq_env_create", device._h, mod.N_ENVIRONMENTS, mod.global_env_offset, C.byref(h))
                    self._adopt(h)
                    self._device = device
                return self._h

            # --- configuration (the MDP constants initialize_environment fills) ---
            @property
            def config(self):
                cfg = _lib.env_config_type()()
                _lib.call("rq_env_get_config", self._require("environment"), C.byref(cfg))
                return cfg

            @config.setter
            def config(self, cfg):
                _lib.call("rq_env_set_config", self._require("environment"), C.byref(cfg))
                self._dt = None

            def _step_dt(self):
                """dt of the configuration in force (what ``step`` returns per env), asked of the library once per configuration."""
                dt = getattr(self, "_dt", None)
                if dt is None:
                    dt = self._dt = float(self.config.dt)
                return dt

            # --- device-resident observation / action buffers ---
            def observation(self):
                out = np.empty((mod.N_ENVIRONMENTS, OBSERVATION_DIM), np.float32)
                _lib.call("rq_env_get_observation", self._require("environment"), _lib.fptr(out))
                return out

            def action(self):
                out = np.empty((mod.N_ENVIRONMENTS, ACTION_DIM), np.float32)
                _lib.call("rq_env_get_action", self._require("environment"), _lib.fptr(out))
                return out

            def _ld(self):
                ld = C.c_uint32()
                _lib.call("rq_env_leading_dim", self._require("environment"), C.byref(ld))
                return ld.value

            def observation_tensor(self):
                """The env's device observation buffer as a torch tensor view, field-major [OBSERVATION_DIM, ld]
                (env i = column i < N_ENVIRONMENTS); zero copy.  Order your torch work behind the engine's stream
                (``device.synchronize()`` or an event on ``device.stream``)."""
                p = C.c_void_p()
                _lib.call("rq_env_observation_device_ptr", self._require("environment"), C.byref(p))
                return _as_torch(p.value, (OBSERVATION_DIM, self._ld()), "<f4", self)

            def action_tensor(self):
                """The env's device action buffer, field-major [4, ld]; zero copy (a learner may write it)."""
                p = C.c_void_p()
                _lib.call("rq_env_action_device_ptr", self._require("environment"), C.byref(p))
                return _as_torch(p.value, (ACTION_DIM, self._ld()), "<f4", self)

            def set_action(self, action):
                a = np.ascontiguousarray(action, np.float32)
                assert a.shape == (mod.N_ENVIRONMENTS, ACTION_DIM)
                _lib.call("rq_env_set_action", self._require("environment"), _lib.fptr(a))

            # --- episode statistics ---
            def _stat(self, fn, dtype, out=None, wait=True):
                h = self._require("environment")
                if out is not None:   # a torch tensor on the same HIP device; wait=False: only enqueued on
                    import torch                                                     # the engine's stream
                    want = {np.float32: torch.float32, np.uint32: torch.int32, np.uint8: torch.uint8}[dtype]
                    if (not out.is_cuda or out.device.index != self._device.ordinal or not out.is_contiguous() or
                            out.numel() != mod.N_ENVIRONMENTS or (out.dtype != want and not
                                                                 (dtype is np.uint32 and out.dtype == getattr(torch, "uint32", None)))):
                        raise ValueError(f"out must be a contiguous {want} tensor of {mod.N_ENVIRONMENTS} elements on "
                                         f"cuda:{self._device.ordinal}")
                    _lib.call(fn, h, C.c_void_p(out.data_ptr()), 1 if wait else 2)
                    return out
                a = np.empty(mod.N_ENVIRONMENTS, dtype)
                _lib.call(fn, h, a.ctypes.data, 0)
                return a

            def rewards(self, out=None): return self._stat("rq_env_get_rewards", np.float32, out)
            def terminated(self, out=None): return self._stat("rq_env_get_terminated", np.uint8, out)
            def done_codes(self, out=None): return self._stat("rq_env_get_done_codes", np.uint8, out)
            def frozen(self, out=None): return self._stat("rq_env_get_frozen", np.uint8, out)
            def episode_index(self, out=None): return self._stat("rq_env_get_episode_index", np.uint32, out)
            def returns(self, out=None): return self._stat("rq_env_get_returns", np.float32, out)
            def episode_steps(self, out=None): return self._stat("rq_env_get_episode_steps", np.uint32, out)
            def finished_returns(self, out=None, wait=True):
                return self._stat("rq_env_get_finished_returns", np.float32, out, wait)
            def finished_lengths(self, out=None): return self._stat("rq_env_get_finished_lengths", np.uint32, out)
            def finished_counts(self, out=None): return self._stat("rq_env_get_finished_counts", np.uint32, out)
            def finished_terminated(self, out=None): return self._stat("rq_env_get_finished_terminated", np.uint32, out)

            def reset_statistics(self):
                _lib.call("rq_env_reset_statistics", self._require("environment"))

        class _Container(_Handle):
            _create = None
            _get = None
            _set = None
            _dim = 0

            def _ensure(self, env):
                if self._h is None:
                    h = C.c_void_p()
                    _lib.call(self._create, env._require("environment"), C.byref(h))
                    self._adopt(h)
                    self._env = env
                return self._h

            def numpy(self):
                out = np.empty((mod.N_ENVIRONMENTS, self._dim), np.float32)
                _lib.call(self._get, self._require(type(self).__name__), _lib.fptr(out))
                return out

            def set(self, array):
                a = np.ascontiguousarray(array, np.float32)
                assert a.shape == (mod.N_ENVIRONMENTS, self._dim), a.shape
                _lib.call(self._set, self._require(type(self).__name__), _lib.fptr(a))

            def tensor(self):
                """The container's device buffer as a torch tensor view, field-major [dim, ld]; zero copy."""
                p = C.c_void_p()
                _lib.call(self._ptr, self._require(type(self).__name__), C.byref(p))
                return _as_torch(p.value, (self._dim, self._env._ld()), "<f4", self)

        class VectorParameters(_Container):
            _destroy, _create, _get, _set, _dim = ("rq_params_destroy", "rq_params_create", "rq_params_get",
                                                   "rq_params_set", PARAM_DIM)
            _ptr = "rq_params_device_ptr"

        class VectorState(_Container):
            _destroy, _create, _get, _set, _dim = ("rq_state_destroy", "rq_state_create", "rq_state_get",
                                                   "rq_state_set", STATE_DIM)
            _ptr = "rq_state_device_ptr"

            _mirror = None      # host copy handed out by ``.states``; written back before the next device use

            def _flush(self):
                m, self._mirror = self._mirror, None
                if m is not None and self._h is not None:
                    _lib.call(self._set, self._h, _lib.fptr(m))

            def _require(self, what):
                if self._mirror is not None:
                    self._flush()
                return super()._require(what)

            def _ensure(self, env):
                if self._mirror is not None:
                    self._flush()
                return super()._ensure(env)

            def assign(self, other):
                """``state.assign(next_state)`` (README.md:99)."""
                if self._h is None:
                    self._ensure(other._env)
                self._mirror = None                       # overwritten anyway
                fast = _lib.fast
                if fast is not None and other._mirror is None:
                    status = fast.assign(_lib.fn_addr("rq_state_assign"), self._h, other._h)
                    if status == 0:
                        return
                    if status != 1:
                        _lib.check(status)
                _lib.call("rq_state_assign", self._h, other._require("VectorState"))

            @property
            def states(self):
                """One view per env (``.position`` etc., README.md:73-75).  Like the reference's, the views are
                writable - ``s.position[0] += 0.1`` (README.md:74) changes this VectorState: they alias a host
                copy that is written back to the device before the next call that uses the state (views kept
                beyond that call are stale)."""
                self._mirror = m = self.numpy()
                return [_StateView(r) for r in m]

            def __copy__(self):
                c = VectorState()
                if self._h is not None:
                    c._ensure(self._env)
                    c.assign(self)
                return c

        class Trajectory(_Handle):
            """Rollout recording buffer: ``vector.rollout(..., trajectory=t)``
            appends per step and env the 22 policy inputs, the raw action, the reward and a done
            code (0 running, 1 terminated, 2 step limit, 4 frozen/not stepped)."""
            _destroy = "rq_trajectory_destroy"

            def __init__(self, env, capacity_steps):
                super().__init__()
                h = C.c_void_p()
                _lib.call("rq_trajectory_create", env._require("environment"), int(capacity_steps), C.byref(h))
                self._adopt(h)
                self._env = env

            def __len__(self):
                n = C.c_uint32()
                _lib.call("rq_trajectory_length", self._h, C.byref(n), None)
                return n.value

            def reset(self):
                _lib.call("rq_trajectory_reset", self._h)

            def numpy(self):
                """-> dict(obs [T,N,22], act [T,N,4], rew [T,N], done [T,N] uint8)"""
                T, N = len(self), mod.N_ENVIRONMENTS
                out = dict(obs=np.empty((T, N, 22), np.float32), act=np.empty((T, N, 4), np.float32),
                           rew=np.empty((T, N), np.float32), done=np.empty((T, N), np.uint8))
                _lib.call("rq_trajectory_get", self._h, _lib.fptr(out["obs"]), _lib.fptr(out["act"]),
                          _lib.fptr(out["rew"]), out["done"].ctypes.data)
                return out

            def tensors(self):
                """The recorded steps as torch tensor VIEWS of the device buffers (zero copy), in the device layout:
                obs [T, 22, ld], act [T, 4, ld], rew [T, ld], done [T, ld] uint8 - env i is index i < N of the
                last axis.  ``obs.permute(0, 2, 1)[:, :N]`` is the learner layout [T, N, 22]."""
                o, a, r, d, ld = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint32()
                _lib.call("rq_trajectory_device_ptrs", self._require("trajectory"), C.byref(o), C.byref(a), C.byref(r),
                          C.byref(d), C.byref(ld))
                T, L = len(self), ld.value
                return dict(obs=_as_torch(o.value, (T, 22, L), "<f4", self), act=_as_torch(a.value, (T, 4, L), "<f4", self),
                            rew=_as_torch(r.value, (T, L), "<f4", self), done=_as_torch(d.value, (T, L), "|u1", self))

            def relabel(self, policy, overwrite=False, fetch=True):
                """Actions of ``policy`` (a teacher, a newer student, ...) on the recorded observations, following
                the recorded episode structure (GRU state reset after episode ends, held on frozen steps; it
                starts from the policy's current hidden state: ``policy.reset()`` for episode starts).
                -> [T, N, 4] (``fetch=False``: only on the device); ``overwrite=True`` also replaces the stored
                actions.  With the policy that recorded the trajectory the result equals the stored actions."""
                T, N = len(self), mod.N_ENVIRONMENTS
                out = np.empty((T, N, 4), np.float32) if fetch else None
                _lib.call("rq_trajectory_relabel", self._h, policy._handle(self._env._device),
                          _lib.fptr(out) if fetch else None, 1 if overwrite else 0)
                return out

            def relabel_teachers(self, bank, teacher_ids, overwrite=False, fetch=True):
                """Actions of MLP teacher ``teacher_ids[i]`` of ``bank`` (raptor_amd.teachers.TeacherBank) on every
                recorded step of env i, in one launch on the matrix cores -> [T, N, 4] (``fetch=False``: only on
                the device); ``overwrite=True`` replaces the stored actions with the teachers' (the regression
                targets of the distillation step, README.md:208-216)."""
                T, N = len(self), mod.N_ENVIRONMENTS
                ids = np.ascontiguousarray(teacher_ids, np.uint32)
                if ids.shape != (N,):
                    raise ValueError("teacher_ids must hold one id per env")
                out = np.empty((T, N, 4), np.float32) if fetch else None
                _lib.call("rq_trajectory_relabel_teachers", self._h, bank._h, ids.ctypes.data,
                          _lib.fptr(out) if fetch else None, 1 if overwrite else 0)
                return out

        self.Trajectory = Trajectory
        self.VectorRng = VectorRng
        self.VectorEnvironment = VectorEnvironment
        self.VectorParameters = VectorParameters
        self.VectorState = VectorState

    # ------------------------------------------------------------------ l2f vector:: functions
    def initialize_rng(self, device, rng, seed):
        """README.md:58"""
        _lib.call("rq_initialize_rng", device._h, rng._ensure(device), int(seed))

    def initialize_environment(self, device, env):
        """README.md:59 — default MDP configuration."""
        _lib.call("rq_initialize_environment", device._h, env._ensure(device))
        env._dt = None

    def sample_initial_parameters(self, device, env, params, rng):
        """README.md:60 — per-env dynamics parameters (domain randomisation)."""
        _lib.call("rq_sample_initial_parameters", device._h, env._require("environment"), params._ensure(env),
                  rng._require("rng"))

    def sample_initial_state(self, device, env, params, state, rng):
        """README.md:61"""
        _lib.call("rq_sample_initial_state", device._h, env._require("environment"),
                  params._require("VectorParameters"), state._ensure(env), rng._require("rng"))

    def observe(self, device, env, params, state, observation, rng):
        """README.md:96 — fills ``observation`` [N, OBSERVATION_DIM] float32 in place
        (``None``: keep it in the env's device buffer)."""
        fast = _lib.fast
        if fast is not None and observation is not None and state._mirror is None:      # the README loop's call: no ctypes in between
            # (csrc/rq_pyfast.c; a state with views handed out goes the ordinary way, which writes them back first)
            status = fast.observe(_lib.fn_addr("rq_observe"), device._h, env._h, params._h, state._h, observation, rng._h,
                                  self.N_ENVIRONMENTS, OBSERVATION_DIM)
            if status == 0:
                return
            if status != 1:
                _lib.check(status)
        ptr = None
        if observation is not None:
            if (observation.dtype != np.float32 or not observation.flags.c_contiguous or
                    observation.shape != (self.N_ENVIRONMENTS, OBSERVATION_DIM)):
                raise ValueError("observation must be C-contiguous float32 [N_ENVIRONMENTS, OBSERVATION_DIM]")
            ptr = _lib.fptr_cached(observation)
        _lib.call("rq_observe", device._h, env._require("environment"), params._require("VectorParameters"),
                  state._require("VectorState"), ptr, rng._require("rng"))

    def step(self, device, env, params, state, action, next_state, rng):
        """README.md:98 — returns the list of per-env dt in seconds.
        ``action`` [N,4] (``None``: the env's device action buffer, e.g. written by
        ``Raptor.evaluate_step_device``)."""
        fast = _lib.fast
        if fast is not None and action is not None and next_state._h is not None and state._mirror is None and next_state._mirror is None:
            status = fast.step(_lib.fn_addr("rq_step"), device._h, env._h, params._h, state._h, action, next_state._h, rng._h,
                               self.N_ENVIRONMENTS)
            if status == 0:
                return [env._step_dt()] * self.N_ENVIRONMENTS
            if status != 1:
                _lib.check(status)
        aptr = None
        if action is not None:
            a = np.ascontiguousarray(action, np.float32)
            if a.shape != (self.N_ENVIRONMENTS, ACTION_DIM):
                raise ValueError("action must be [N_ENVIRONMENTS, 4]")
            aptr = _lib.fptr(a)
        _lib.call("rq_step", device._h, env._require("environment"), params._require("VectorParameters"),
                  state._require("VectorState"), aptr, next_state._ensure(env), rng._require("rng"), None)
        # every env advances by the configured dt (rq_step's dts output is that constant N times); turning 65 536
        # float32 into Python floats one by one took longer than the step itself (numpy tolist: ~1 ms)
        # (the dt itself is remembered per env: asking the library for the whole configuration cost the README loop 2 us per step)
        return [env._step_dt()] * self.N_ENVIRONMENTS

    def step_device(self, device, env, params, state, next_state, rng):
        """``step`` without host traffic: action from the env's device buffer, no dt list."""
        _lib.call("rq_step", device._h, env._require("environment"), params._require("VectorParameters"),
                  state._require("VectorState"), None, next_state._ensure(env), rng._require("rng"), None)

    def rollout(self, device, env, params, state, policy, rng, n_steps, mode="fused", autoreset=False,
                trajectory=None):
        """The loop body README.md:95-99, ``n_steps`` times, entirely on the device; with
        ``trajectory`` every transition is also appended to that buffer."""
        m = {"fused": ROLLOUT_FUSED, "chained": ROLLOUT_CHAINED}[mode]
        fast = _lib.fast
        if fast is not None and trajectory is None and state._mirror is None and hasattr(fast, "rollout"):
            status = fast.rollout(_lib.fn_addr("rq_rollout"), device._h, env._h, params._h, state._h, policy._handle(device), rng._h,
                                  int(n_steps), m, ROLLOUT_AUTORESET if autoreset else 0)
            if status == 0:
                return
            if status != 1:
                _lib.check(status)
        args = (device._h, env._require("environment"), params._require("VectorParameters"),
                state._require("VectorState"), policy._handle(device), rng._require("rng"), int(n_steps), m,
                ROLLOUT_AUTORESET if autoreset else 0)
        if trajectory is None:
            _lib.call("rq_rollout", *args)
        else:
            _lib.call("rq_rollout_record", *args, trajectory._require("trajectory"))


    # ------------------------------------------------------------------ ui-server messages (README.md:63-92)
    # Only three things about the wire format are visible in the reference tree: every message is a JSON object
    # with "namespace" (from the handshake) and "channel", and the parameters message carries "data", a list with
    # one entry per env that a client may extend (README.md:63-70 adds d["ui"] = {"model", "name"}).  Channel
    # names and the field names inside data[i] follow the l2f parameter / state structure as recalled
    # [UPSTREAM-UNVERIFIED: no file in /root/reference states them]; they are here so that GPU rollouts can be
    # inspected with a ui-server-like client, not as a parity claim.
    @staticmethod
    def _num(v):
        """JSON has no NaN / Infinity (a browser's JSON.parse rejects the tokens Python would emit): null instead."""
        v = float(v)
        return v if np.isfinite(v) else None

    def set_ui_message(self, device, env, ui):
        """README.md:86 - announces what is going to be rendered: N quadrotors."""
        return json.dumps({"namespace": ui.ns, "channel": "setUI",
                           "data": {"type": "l2f-vector", "n_environments": self.N_ENVIRONMENTS}})

    def set_parameters_message(self, device, env, params, ui):
        """README.md:87 - one entry of dynamics parameters per env in ``data``."""
        P = params.numpy()
        data = []
        for p in P:
            data.append({"dynamics": {
                "mass": float(p[0]), "J": [[float(p[1]), 0.0, 0.0], [0.0, float(p[2]), 0.0], [0.0, 0.0, float(p[3])]],
                "rotor_positions": [[float(v) for v in p[4 + 3 * i:7 + 3 * i]] for i in range(4)],
                "rotor_thrust_directions": [[0.0, 0.0, 1.0]] * 4,
                "rotor_torque_directions": [[0.0, 0.0, d] for d in (-1.0, 1.0, -1.0, 1.0)],
                "rotor_thrust_coefficients": [[float(p[16]), float(p[17]), float(p[18])]] * 4,
                "rotor_torque_constants": [float(p[19])] * 4,
                "rotor_time_constants_rising": [float(p[20])] * 4, "rotor_time_constants_falling": [float(p[21])] * 4,
                "action_limit": {"min": float(p[22]), "max": float(p[23])}, "hovering_rpm": float(p[24])}})
        return json.dumps({"namespace": ui.ns, "channel": "setParameters", "data": data}, allow_nan=False)

    def set_state_action_message(self, device, env, params, ui, state, action):
        """README.md:76 - the states (a ``copy(state)`` whose ``.states[i].position`` was shifted works, README.md:73-75)
        and the actions about to be applied, one entry per env."""
        S = state.numpy()
        A = np.asarray(action, np.float32).reshape(self.N_ENVIRONMENTS, ACTION_DIM)
        f = self._num
        data = [{"state": {"position": [f(v) for v in s[0:3]], "orientation": [f(v) for v in s[3:7]],
                           "linear_velocity": [f(v) for v in s[7:10]], "angular_velocity": [f(v) for v in s[10:13]],
                           "rpm": [f(v) for v in s[13:17]]},
                 "action": [f(v) for v in a]} for s, a in zip(S, A)]
        return json.dumps({"namespace": ui.ns, "channel": "setStateAction", "data": data}, allow_nan=False)


_modules = {}


def vector(n_environments, global_env_offset=0):
    """Runtime-sized equivalent of the reference's ``l2f.vectorN`` modules."""
    key = (int(n_environments), int(global_env_offset))
    if key not in _modules:
        _modules[key] = VectorModule(*key)
    return _modules[key]


vector8 = vector(8)   # README.md:45


def __getattr__(name):
    """``from l2f import vectorN`` for any N (the reference ships a fixed set of pre-compiled ``vectorN`` modules, README.md:45
    imports ``vector8``; here N is a run-time value, so every positive N resolves)."""
    if name.startswith("vector") and name[6:].isdigit() and int(name[6:]) > 0 and not name[6:].startswith("0"):
        return vector(int(name[6:]))
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
