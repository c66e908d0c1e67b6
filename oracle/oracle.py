"""ctypes front-end of oracle/libraptor_oracle.so — TEST INFRASTRUCTURE ONLY.

Importable from tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
only; nothing under ``raptor_amd/`` may import it (tests/test_no_oracle_in_product.py checks).

All arrays are array-of-structs, float32, C-contiguous: params [n, 26], state [n, 27],
observation [n, 26], action [n, 4], hidden [n, 16] (field order: include/raptor_quad.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libraptor_oracle.so")

PARAM_DIM, STATE_DIM, OBS_DIM, ACTION_DIM, HIDDEN_DIM = 26, 27, 26, 4, 16
NUM_WEIGHTS = 2084


class EnvConfig(C.Structure):
    """Mirror of ``rq_env_config`` (include/raptor_quad.h)."""
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("dt", C.c_float), ("gravity", C.c_float), ("episode_step_limit", C.c_uint32),
        ("domain_randomization", C.c_uint32),
        ("dr_scale_min", C.c_float), ("dr_scale_max", C.c_float),
        ("dr_thrust_to_weight_min", C.c_float), ("dr_thrust_to_weight_max", C.c_float),
        ("dr_torque_const_min", C.c_float), ("dr_torque_const_max", C.c_float),
        ("dr_motor_tau_min", C.c_float), ("dr_motor_tau_max", C.c_float),
        ("init_guidance", C.c_float), ("init_max_position", C.c_float), ("init_max_angle", C.c_float),
        ("init_max_linear_velocity", C.c_float), ("init_max_angular_velocity", C.c_float),
        ("disturbance_force_std", C.c_float), ("disturbance_torque_std", C.c_float),
        ("noise_position", C.c_float), ("noise_orientation", C.c_float),
        ("noise_linear_velocity", C.c_float), ("noise_angular_velocity", C.c_float),
        ("reward_scale", C.c_float), ("reward_constant", C.c_float),
        ("reward_termination_penalty", C.c_float),
        ("reward_position", C.c_float), ("reward_orientation", C.c_float),
        ("reward_linear_velocity", C.c_float), ("reward_angular_velocity", C.c_float),
        ("reward_action", C.c_float),
        ("termination_enabled", C.c_uint32),
        ("termination_position", C.c_float), ("termination_linear_velocity", C.c_float),
        ("termination_angular_velocity", C.c_float),
        ("action_history_raw", C.c_uint32),
    ]

    def __setattr__(self, name, value):       # a misspelt field must not silently configure nothing
        if name not in EnvConfig._field_names:
            raise AttributeError(f"oracle config has no field '{name}'")
        super().__setattr__(name, value)


EnvConfig._field_names = frozenset(n for n, _ in EnvConfig._fields_)


def build(force=False):
    """Compile the restatement with oracle/Makefile (gcc only; seconds)."""
    src = os.path.join(_HERE, "raptor_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B", "libraptor_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_max_threads.restype = C.c_int
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def default_config():
    cfg = EnvConfig()
    lib().orc_default_config(C.byref(cfg))
    return cfg


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return list(o)


def actor_sequence(weights, inp):
    """inp [T,B,22] -> out [T,B,4]; hidden starts at the checkpoint's initial_hidden_state."""
    w, wp = _f(weights)
    x, xp = _f(inp)
    T, B, _ = x.shape
    out = np.empty((T, B, 4), np.float32)
    lib().orc_actor_sequence(wp, xp, _p(out, C.c_float), C.c_uint32(T), C.c_uint32(B))
    return out


def actor_batch_step(weights, obs, hidden):
    """One recurrent step. obs [B,>=22]; hidden [B,16] updated IN PLACE; returns act [B,4]."""
    w, wp = _f(weights)
    o, op = _f(obs)
    assert hidden.dtype == np.float32 and hidden.flags.c_contiguous
    B = o.shape[0]
    act = np.empty((B, 4), np.float32)
    lib().orc_actor_batch_step(wp, op, C.c_uint32(o.shape[1]), _p(hidden, C.c_float),
                               _p(act, C.c_float), C.c_uint32(B))
    return act


def actor_batch_step_sas(weights, w_ls, b_ls, mode, seed, step, env_offset, obs, hidden):
    """One recurrent step with the SampleAndSquash output stage (mode 0 off, 1 tanh(mean), 2 sample)."""
    w, wp = _f(weights)
    o, op = _f(obs)
    B = o.shape[0]
    act = np.empty((B, 4), np.float32)
    wl = None if w_ls is None else _f(w_ls)
    bl = None if b_ls is None else _f(b_ls)
    lib().orc_actor_batch_step_sas(wp, None if wl is None else wl[1], None if bl is None else bl[1], C.c_int(mode),
                                   C.c_uint64(seed), C.c_uint32(step), C.c_uint64(env_offset), op,
                                   C.c_uint32(o.shape[1]), _p(hidden, C.c_float), _p(act, C.c_float), C.c_uint32(B))
    return act


def sample_initial_parameters(cfg, seed, epoch, env_offset, n):
    out = np.zeros((n, PARAM_DIM), np.float32)
    lib().orc_sample_initial_parameters(C.byref(cfg), C.c_uint64(seed), C.c_uint32(epoch),
                                        C.c_uint64(env_offset), C.c_uint32(n), _p(out, C.c_float))
    return out


def sample_initial_state(cfg, seed, episode, env_offset, params):
    """episode: uint32 [n], used as counter then incremented in place."""
    p, pp = _f(params)
    n = p.shape[0]
    assert episode.dtype == np.uint32
    out = np.zeros((n, STATE_DIM), np.float32)
    lib().orc_sample_initial_state(C.byref(cfg), C.c_uint64(seed), _p(episode, C.c_uint32),
                                   C.c_uint64(env_offset), C.c_uint32(n), pp, _p(out, C.c_float))
    return out


def observe(cfg, seed, epoch, env_offset, params, state):
    p, pp = _f(params)
    s, sp = _f(state)
    n = p.shape[0]
    out = np.zeros((n, OBS_DIM), np.float32)
    lib().orc_observe(C.byref(cfg), C.c_uint64(seed), C.c_uint32(epoch), C.c_uint64(env_offset),
                      C.c_uint32(n), pp, sp, _p(out, C.c_float))
    return out


def step(cfg, params, state, action):
    """-> next_state [n,27], reward [n], terminated [n] uint8"""
    p, pp = _f(params)
    s, sp = _f(state)
    a, ap = _f(action)
    n = p.shape[0]
    ns = np.zeros((n, STATE_DIM), np.float32)
    r = np.zeros(n, np.float32)
    t = np.zeros(n, np.uint8)
    lib().orc_step(C.byref(cfg), C.c_uint32(n), pp, sp, ap, _p(ns, C.c_float), _p(r, C.c_float),
                   _p(t, C.c_uint8))
    return ns, r, t


class Stats:
    """Episode statistics arrays, same meaning as the rq_env_get_* getters."""

    def __init__(self, n):
        self.returns = np.zeros(n, np.float32)
        self.steps = np.zeros(n, np.uint32)
        self.fin_returns = np.zeros(n, np.float32)
        self.fin_lengths = np.zeros(n, np.uint32)
        self.fin_counts = np.zeros(n, np.uint32)
        self.fin_terminated = np.zeros(n, np.uint32)
        self.frozen = np.zeros(n, np.uint8)
        self.episode = np.zeros(n, np.uint32)
        self.last_reward = np.zeros(n, np.float32)
        self.last_terminated = np.zeros(n, np.uint8)


def stats_update(cfg, reward, terminated, st):
    n = reward.shape[0]
    lib().orc_stats_update(C.byref(cfg), C.c_uint32(n), _p(reward, C.c_float), _p(terminated, C.c_uint8),
                           _p(st.returns, C.c_float), _p(st.steps, C.c_uint32),
                           _p(st.fin_returns, C.c_float), _p(st.fin_lengths, C.c_uint32),
                           _p(st.fin_counts, C.c_uint32), _p(st.fin_terminated, C.c_uint32))


def rollout(cfg, weights, seed, epoch0, env_offset, params, state, hidden, n_steps, flags, st,
            nthreads=1):
    """state [n,27] and hidden [n,16] are advanced IN PLACE; st (Stats) updated in place."""
    w, wp = _f(weights)
    p, pp = _f(params)
    assert state.dtype == np.float32 and state.flags.c_contiguous
    assert hidden.dtype == np.float32 and hidden.flags.c_contiguous
    n = p.shape[0]
    lib().orc_rollout(C.byref(cfg), wp, C.c_uint64(seed), C.c_uint32(epoch0), C.c_uint64(env_offset),
                      C.c_uint32(n), pp, _p(state, C.c_float), _p(hidden, C.c_float),
                      C.c_uint32(n_steps), C.c_uint32(flags),
                      _p(st.returns, C.c_float), _p(st.steps, C.c_uint32),
                      _p(st.fin_returns, C.c_float), _p(st.fin_lengths, C.c_uint32),
                      _p(st.fin_counts, C.c_uint32), _p(st.fin_terminated, C.c_uint32),
                      _p(st.frozen, C.c_uint8), _p(st.episode, C.c_uint32),
                      _p(st.last_reward, C.c_float), _p(st.last_terminated, C.c_uint8),
                      C.c_int(nthreads))


def rollout_record(cfg, weights, seed, epoch0, env_offset, params, state, hidden, n_steps, flags, st,
                   nthreads=1):
    """As ``rollout`` and returns the trajectory dict(obs [T,n,22], act [T,n,4], rew [T,n], done [T,n])."""
    w, wp = _f(weights)
    p, pp = _f(params)
    n = p.shape[0]
    tr = dict(obs=np.zeros((n_steps, n, 22), np.float32), act=np.zeros((n_steps, n, 4), np.float32),
              rew=np.zeros((n_steps, n), np.float32), done=np.zeros((n_steps, n), np.uint8))
    lib().orc_rollout_record(C.byref(cfg), wp, C.c_uint64(seed), C.c_uint32(epoch0), C.c_uint64(env_offset),
                             C.c_uint32(n), pp, _p(state, C.c_float), _p(hidden, C.c_float),
                             C.c_uint32(n_steps), C.c_uint32(flags),
                             _p(st.returns, C.c_float), _p(st.steps, C.c_uint32),
                             _p(st.fin_returns, C.c_float), _p(st.fin_lengths, C.c_uint32),
                             _p(st.fin_counts, C.c_uint32), _p(st.fin_terminated, C.c_uint32),
                             _p(st.frozen, C.c_uint8), _p(st.episode, C.c_uint32),
                             _p(st.last_reward, C.c_float), _p(st.last_terminated, C.c_uint8),
                             C.c_int(nthreads), _p(tr["obs"], C.c_float), _p(tr["act"], C.c_float),
                             _p(tr["rew"], C.c_float), _p(tr["done"], C.c_uint8))
    return tr


def teacher_relabel(weights, in_dim, h1, h2, act, out_act, obs, teacher_id, nthreads=0):
    """MLP teachers (codes: 0 identity, 1 ReLU, 2 tanh): weights [n_teachers, P], obs [T, n, 22],
    teacher_id [n] -> actions [T, n, 4]."""
    w, wp = _f(weights)
    o, op = _f(obs)
    ids = np.ascontiguousarray(teacher_id, np.uint32)
    T, n, _ = o.shape
    out = np.empty((T, n, 4), np.float32)
    lib().orc_teacher_relabel(wp, C.c_uint32(in_dim), C.c_uint32(h1), C.c_uint32(h2), C.c_int(act), C.c_int(out_act),
                              op, _p(ids, C.c_uint32), C.c_uint32(T), C.c_uint32(n), _p(out, C.c_float),
                              C.c_int(nthreads))
    return out


def mlp_relabel(weights, in_dim, widths, act, out_act, obs, teacher_id, nthreads=0):
    """Any stack of dense layers in_dim -> widths[0] -> ... -> 4 (1 to 3 hidden layers, widths <= 128; codes: 0 identity, 1 ReLU,
    2 tanh): weights [n_teachers, P] as [W1 | b1 | ... | W_out | b_out], obs [T, n, 22], teacher_id [n] -> actions [T, n, 4]."""
    w, wp = _f(weights)
    o, op = _f(obs)
    ids = np.ascontiguousarray(teacher_id, np.uint32)
    wd = np.ascontiguousarray(widths, np.uint32)
    T, n, _ = o.shape
    out = np.empty((T, n, 4), np.float32)
    lib().orc_mlp_relabel(wp, C.c_uint32(in_dim), C.c_uint32(len(wd)), _p(wd, C.c_uint32), C.c_int(act), C.c_int(out_act),
                          op, _p(ids, C.c_uint32), C.c_uint32(T), C.c_uint32(n), _p(out, C.c_float), C.c_int(nthreads))
    return out


def max_threads():
    return int(lib().orc_max_threads())
