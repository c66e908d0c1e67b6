"""Shared by the tests/test_gpu_*.py modules: the tolerances, the World fixture object (all l2f-shaped objects for one batch on the GPU plus
the oracle-side mirror) and the long-horizon closed-loop comparison."""
import os

import numpy as np

ACTOR_TOL = 1e-5        # abs, raw actions in [-2.8, 3.4]; reference KATs (checkpoint.h:197-215, h5:/example)
INIT_TOL = 2e-6         # abs, initial attitude via sinf/cosf (device vs libm)
NOISE_TOL = 2e-5        # abs per unit std, Box-Muller (hardware v_log/v_sqrt/v_sin/v_cos vs libm); asserted at 10x
CLOSED_LOOP_TOL = 2e-3  # abs on p, q, v after 500 closed-loop steps (actor ulps fed back through the dynamics)


class World:
    """All l2f-shaped objects for one batch, on the GPU, plus the oracle-side mirror."""

    def __init__(self, device, oracle, n, seed=0, offset=0, **cfg_over):
        import raptor_amd.l2f as l2f
        from raptor_amd.foundation_policy import Raptor
        self.O = oracle
        self.n, self.seed, self.offset = n, seed, offset
        self.device = device
        self.vector = v = l2f.VectorModule(n, offset)
        self.rng, self.env = v.VectorRng(), v.VectorEnvironment()
        self.params, self.state, self.next_state = v.VectorParameters(), v.VectorState(), v.VectorState()
        v.initialize_rng(device, self.rng, seed)
        v.initialize_environment(device, self.env)
        cfg = self.env.config
        for k, val in cfg_over.items():
            setattr(cfg, k, val)
        self.env.config = cfg
        self.cfg = oracle.default_config()
        for k, val in cfg_over.items():
            setattr(self.cfg, k, val)
        assert bytes(self.cfg) == bytes(self.env.config)
        self.policy = Raptor(device)
        v.sample_initial_parameters(device, self.env, self.params, self.rng)
        v.sample_initial_state(device, self.env, self.params, self.state, self.rng)
        # oracle mirror
        self.P = oracle.sample_initial_parameters(self.cfg, seed, 0, offset, n)
        self.st = oracle.Stats(n)
        self.S = oracle.sample_initial_state(self.cfg, seed, self.st.episode, offset, self.P)
        self.H = np.zeros((n, 16), np.float32)

    def sync_oracle_to_gpu_state(self):
        """Start both sides from the GPU's initial state (it differs from the oracle's by sin/cos ulps)."""
        self.S = self.state.numpy()


# ------------------------------------------------------------------------------ bf16 actor --
BF16_KAT_TOL = 5e-2     # abs on raw actions: bf16 operands (8-bit mantissa), fp32 accumulate (numpy model: 1.9e-2)


def _well_conditioned(w, weights, steps, flags, threads=8):
    """The closed loop is chaotic for a few percent of the randomised quadrotors (fast motors on
    small frames: a 1-ulp change of the initial x position grows to O(1) rad/s within 1-2 s —
    measured with the oracle against itself).  Parity over a long horizon is therefore asserted
    on the envs whose own sensitivity is small; the one-step-ahead test below covers all envs."""
    O = w.O
    Sp = w.S.copy()
    Sp[:, 0] = np.nextafter(Sp[:, 0], np.float32(10))
    Hp = w.H.copy()
    stp = O.Stats(w.n)
    stp.episode[:] = w.st.episode
    O.rollout(w.cfg, weights, w.seed, 0, w.offset, w.P, Sp, Hp, steps, flags, stp, threads)
    return Sp, stp


_FRACTIONS = []


def _report_fractions(w, steps, flags, same_history, insensitive, same_history_of_insensitive):
    """The measured fractions behind the long-horizon bars: printed (pytest -s) and, on the GPU box, collected in
    gpurun_out/closed_loop_fractions.json so that the thresholds can be checked against what was measured."""
    import json
    rec = dict(n=int(w.n), seed=int(w.seed), steps=int(steps), autoreset=int(flags),
               domain_randomization=int(w.cfg.domain_randomization), same_history=round(float(same_history), 4),
               insensitive=round(float(insensitive), 4),
               same_history_of_insensitive=round(float(same_history_of_insensitive), 4))
    _FRACTIONS.append(rec)
    print("closed-loop fractions:", rec)
    try:
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        if os.path.isdir(out):
            json.dump(_FRACTIONS, open(os.path.join(out, "closed_loop_fractions.json"), "w"), indent=1)
    except OSError:
        pass


def _closed_loop_agreement(w, weights, steps, flags, min_same_history=0.99, min_insensitive=0.80):
    S0 = w.S.copy()
    Sp, stp = _well_conditioned(w, weights, steps, flags)
    w.O.rollout(w.cfg, weights, w.seed, 0, w.offset, w.P, w.S, w.H, steps, flags, w.st, 8)
    S = w.state.numpy()
    g_cnt, g_len, g_term = w.env.finished_counts(), w.env.finished_lengths(), w.env.finished_terminated()
    same_history = (g_cnt == w.st.fin_counts) & (g_len == w.st.fin_lengths) & (g_term == w.st.fin_terminated)
    insensitive = (np.abs(Sp[:, :13] - w.S[:, :13]).max(axis=1) < 1e-5) & \
                  (stp.fin_counts == w.st.fin_counts) & (stp.fin_lengths == w.st.fin_lengths)
    _report_fractions(w, steps, flags, same_history.mean(), insensitive.mean(), same_history[insensitive].mean())
    # thresholds sit just under the fractions measured on the MI355X (profiles/r04_closed_loop_fractions.json, same figures as r02 / r03:
    # 500 steps: same history 0.996-1.0, insensitive 0.83 with domain randomisation, 0.875 without)
    assert same_history.mean() >= min_same_history, same_history.mean()
    assert insensitive.mean() >= min_insensitive, insensitive.mean()
    sel = insensitive & same_history
    # the 1-ulp-of-x probe is a proxy for sensitivity to the actor's ulps: allow 1 % escapes
    assert same_history[insensitive].mean() > 0.99
    d = np.abs(S[sel, :13] - w.S[sel, :13]).max(axis=1)
    assert np.quantile(d, 0.99) < CLOSED_LOOP_TOL, np.quantile(d, [0.5, 0.99, 1.0])
    dr = np.abs(w.env.finished_returns()[sel] - w.st.fin_returns[sel])
    assert np.quantile(dr, 0.99) < 5e-2, np.quantile(dr, [0.5, 0.99, 1.0])   # returns ~ 700 per episode
    # population level: the GPU's spread vs the oracle is no worse than the oracle's own 1-ulp spread
    all_d = np.abs(S[:, :13] - w.S[:, :13]).max(axis=1)
    ref_d = np.abs(Sp[:, :13] - w.S[:, :13]).max(axis=1)
    assert np.nanmedian(all_d) < 1e-4 and (all_d > 1e-2).mean() <= (ref_d > 1e-2).mean() + 0.05
    return sel


def _lib_set_epoch(w, epoch):
    from raptor_amd import _lib
    _lib.call("rq_rng_set_epoch", w.rng._h, epoch)
