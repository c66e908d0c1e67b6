"""The resident executor of the small-batch loop (round 6; include/raptor_quad.h rq_device_set_resident, rq_kernels.hip k_resident_small /
k_resident_loop): the reference's loop at its own batch - README.md:96-99 with `vector8`, NumPy arrays at every call - served by one
workgroup that stays on the device instead of two launches per iteration.  Results never depend on it: every test here compares it,
bit for bit, with the launches it replaces."""
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from gpu_common import World      # noqa: F401

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _loop(device, n, iters, resident, between=None, seed=0):
    """The README loop as written; -> (observations, actions, final state, hidden state, rewards, resident statistics)."""
    import raptor_amd.l2f as l2f
    from raptor_amd.foundation_policy import Raptor
    device.set_resident(resident)
    vector = l2f.vector(n)
    rng, env = vector.VectorRng(), vector.VectorEnvironment()
    params, state, next_state = vector.VectorParameters(), vector.VectorState(), vector.VectorState()
    vector.initialize_rng(device, rng, seed)
    vector.initialize_environment(device, env)
    vector.sample_initial_parameters(device, env, params, rng)
    vector.sample_initial_state(device, env, params, state, rng)
    policy = Raptor(device)
    policy.reset()
    obs = np.zeros((n, env.OBSERVATION_DIM), np.float32)
    O, A = [], []
    before = device.resident()
    for it in range(iters):
        vector.observe(device, env, params, state, obs, rng)
        action = policy.evaluate_step(obs[:, :22])
        vector.step(device, env, params, state, action, next_state, rng)
        state.assign(next_state)
        O.append(obs.copy())
        A.append(action.copy())
        if between is not None:
            between(it, locals())
    after = device.resident()
    stats = {k: after[k] - before[k] for k in ("starts", "commands", "replays")}
    out = (np.array(O), np.array(A), state.numpy().copy(), policy.hidden_state(n).copy(), env.rewards().copy(), env.terminated().copy(), stats)
    device.set_resident(True)
    return out


def _same(a, b):
    return all(np.array_equal(np.ascontiguousarray(x).view(np.uint8), np.ascontiguousarray(y).view(np.uint8)) for x, y in zip(a[:6], b[:6]))


@pytest.mark.parametrize("n", [1, 8, 12, 13, 64, 65, 200, 256])
def test_readme_loop_is_bit_identical_with_and_without_the_resident_executor(device, n):
    """Every observation, action, state, hidden state, reward and termination flag of 150 iterations: the kernel for at most 12 envs
    (rows in the poll, registers resident across commands, single-tile policy step), the general one (1 - 4 waves), and their borders."""
    off = _loop(device, n, 150, False)
    on = _loop(device, n, 150, True)
    assert _same(off, on)
    assert off[6]["commands"] == 0 and on[6]["commands"] >= 140 and on[6]["replays"] == 0, (off[6], on[6])
    assert on[6]["starts"] <= 10         # one kernel per 0.75 ms of looping (rq_objects.hpp kResidentHostLifeNs), none per iteration


def test_beyond_256_envs_the_loop_keeps_its_launches(device):
    on = _loop(device, 300, 20, True)
    assert on[6]["commands"] == 0 and on[6]["starts"] == 0


def test_anything_else_asked_of_the_device_retires_the_executor_first(device, oracle):
    """A call that is not one of the loop's four - here: reading the state back, a large fused rollout of another env, changing the
    configuration, observing with noise - finds no resident kernel on the device any more, and what the loop computes is what it
    computes without the executor, whatever is interleaved."""
    import raptor_amd.l2f as l2f
    big = World(device, oracle, 65536, seed=5)
    seen = []

    def between(it, L):
        if it == 20:
            assert L["device"].resident()["running"]
            L["state"].numpy()                                   # a copy on the device's stream
            seen.append(L["device"].resident()["running"])
        if it == 40:
            assert L["device"].resident()["running"]
            big.vector.rollout(device, big.env, big.params, big.state, big.policy, big.rng, 20, "fused", True)
            seen.append(device.resident()["running"])
        if it == 60:
            cfg = L["env"].config
            cfg.termination_position = 0.7                       # host-side only: the NEXT step must notice
            L["env"].config = cfg
        if it == 80:
            L["policy"].reset()                                  # the policy's state replaced from outside
    on = _loop(device, 8, 120, True, between)
    big2 = World(device, oracle, 65536, seed=5)
    seen_off = []

    def between_off(it, L):
        if it == 20:
            L["state"].numpy()
        if it == 40:
            big2.vector.rollout(device, big2.env, big2.params, big2.state, big2.policy, big2.rng, 20, "fused", True)
        if it == 60:
            cfg = L["env"].config
            cfg.termination_position = 0.7
            L["env"].config = cfg
        if it == 80:
            L["policy"].reset()
    off = _loop(device, 8, 120, False, between_off)
    assert seen == [False, False]
    assert _same(off, on)
    assert np.array_equal(big.state.numpy(), big2.state.numpy())
    assert on[6]["starts"] >= 4 and on[6]["replays"] == 0 and seen_off == []


def test_a_large_launch_behind_the_loop_is_not_slowed_down(device, oracle):
    """The resident wave holds registers of one SIMD; a 65 536-env fused rollout needs every SIMD of the chip.  It is retired before the
    launch is enqueued: the launch behind a README loop takes what it takes on an idle device (within 10 %)."""
    big = World(device, oracle, 65536, seed=9)

    def kernel_ms():
        device.set_rollout_timing(True)
        big.vector.rollout(device, big.env, big.params, big.state, big.policy, big.rng, 100, "fused", True)
        ms = device.last_rollout_ms()
        device.set_rollout_timing(False)
        return ms
    for _ in range(3):
        kernel_ms()
    idle = min(kernel_ms() for _ in range(5))
    behind = []
    for _ in range(5):
        _loop(device, 8, 30, True, between=lambda it, L: behind.append(kernel_ms()) if it == 29 else None)
    assert min(behind) < 1.10 * idle, (idle, behind)


def test_the_executor_is_released_with_its_env_and_with_its_device():
    """rq_env_destroy / rq_device_destroy while a resident kernel is running: it is told to leave and has left when the call returns
    (a kernel left spinning would also keep hipFree - a device-wide synchronize - waiting for its idle limit)."""
    import gc
    import raptor_amd.l2f as l2f
    from raptor_amd.foundation_policy import Raptor
    dev = l2f.Device(0)
    vector = l2f.vector(8)
    rng, env = vector.VectorRng(), vector.VectorEnvironment()
    params, state, next_state = vector.VectorParameters(), vector.VectorState(), vector.VectorState()
    vector.initialize_rng(dev, rng, 0)
    vector.initialize_environment(dev, env)
    vector.sample_initial_parameters(dev, env, params, rng)
    vector.sample_initial_state(dev, env, params, state, rng)
    policy = Raptor(dev)
    policy.reset()
    obs = np.zeros((8, env.OBSERVATION_DIM), np.float32)
    for _ in range(10):
        vector.observe(dev, env, params, state, obs, rng)
        vector.step(dev, env, params, state, policy.evaluate_step(obs[:, :22]), next_state, rng)
        state.assign(next_state)
    assert dev.resident()["running"]
    t0 = time.perf_counter()
    del env
    gc.collect()
    assert not dev.resident()["running"]
    assert time.perf_counter() - t0 < 0.5
    # and a device that goes while one is running (a fresh loop on fresh objects)
    env = vector.VectorEnvironment()
    params, state, next_state = vector.VectorParameters(), vector.VectorState(), vector.VectorState()
    vector.initialize_environment(dev, env)
    vector.sample_initial_parameters(dev, env, params, rng)
    vector.sample_initial_state(dev, env, params, state, rng)
    policy.reset()
    for _ in range(30):          # fresh buffers: the first steps allocate, and steps further than 200 us apart do not count
        vector.observe(dev, env, params, state, obs, rng)
        vector.step(dev, env, params, state, policy.evaluate_step(obs[:, :22]), next_state, rng)
        state.assign(next_state)
    assert dev.resident()["running"]
    del dev, env, params, state, next_state, policy, rng
    gc.collect()


def test_idle_and_old_kernels_leave_by_themselves_and_the_loop_goes_on(device):
    """No command for 300 us: the kernel has left (a host that went away must not leave a wave spinning), and the next steps start
    another.  1 ms old: it leaves between two commands whatever the traffic - a device-wide synchronize on another thread (a
    learner's torch.cuda.synchronize(), any hipFree) waits for a running kernel - and the host starts the next one in time: a loop
    of 600 iterations (~5 ms) needs several kernels and no replay."""
    def between(it, L):
        if it == 50:
            assert L["device"].resident()["running"]
            time.sleep(0.02)
            assert not L["device"].resident()["running"]
    on = _loop(device, 8, 600, True, between)
    off = _loop(device, 8, 600, False, lambda it, L: time.sleep(0.02) if it == 50 else None)
    assert _same(off, on)
    assert on[6]["starts"] >= 3 and on[6]["replays"] == 0, on[6]


def test_the_loop_as_the_reference_paces_it_keeps_its_launches(device):
    """README.md:94-101 sleeps dts[-1] = 10 ms after every step: a wave spinning through that sleep would serve nobody.  Steps count
    towards a resident kernel only when they follow one another within 200 us (rq_objects.hpp kResidentMaxGapNs)."""
    on = _loop(device, 8, 12, True, lambda it, L: time.sleep(0.002))
    assert on[6]["starts"] == 0 and on[6]["commands"] == 0, on[6]


def _hip_device_synchronize():
    import ctypes
    with open("/proc/self/maps") as f:           # the HIP runtime this process has mapped already (PyTorch's copy, raptor_amd/_lib.py)
        paths = {line.split()[-1] for line in f if "libamdhip64.so" in line}
    assert len(paths) == 1, paths
    hip = ctypes.CDLL(paths.pop())
    hip.hipDeviceSynchronize.restype = ctypes.c_int
    return hip.hipDeviceSynchronize


def test_a_caller_that_synchronizes_the_device_every_iteration_is_not_made_to_wait_for_idle_kernels(device):
    """A device-wide synchronize of the caller's own between two steps (a learner in the same thread: torch.cuda.synchronize(), an
    allocation that frees) waits until the resident kernel has idled out - 300 us against the ~20 us the launches take.  The library
    cannot see that call; it sees its result: a kernel that left by itself having served next to nothing.  Such kernels are started
    8, 16, ... 1 024 steps apart (kResidentMinCommands): 400 iterations pay for a handful of them, and compute what they compute
    without the executor."""
    sync = _hip_device_synchronize()

    def between(it, L):
        assert sync() == 0
    t0 = time.perf_counter()
    on = _loop(device, 8, 400, True, between)
    t_on = (time.perf_counter() - t0) / 400 * 1e6
    t0 = time.perf_counter()
    off = _loop(device, 8, 400, False, between)
    t_off = (time.perf_counter() - t0) / 400 * 1e6
    print(f"[README loop + hipDeviceSynchronize per iteration] launches {t_off:.1f} us, with the executor enabled {t_on:.1f} us; {on[6]}")
    assert _same(off, on)
    assert 1 <= on[6]["starts"] <= 8 and on[6]["replays"] == 0, on[6]
    assert t_on < t_off + 25.0, (t_on, t_off)          # (measured: + 4.5 .. 6 us; without the back-off: + 300)


@pytest.mark.timeout(300)
def test_a_command_the_kernel_never_took_is_replayed_as_launches():
    """The race the protocol has to survive: the host posts a command to a kernel that is leaving.  Forced here: kernels that leave after
    20 us of idling (RQ_RESIDENT_IDLE_TICKS) and a host that keeps posting to them regardless (RQ_RESIDENT_HOST_IDLE_NS, _HOST_LIFE_NS), with pauses
    in the loop - the commands posted into the void are noticed (`exited`), replayed on the stream, and nothing differs."""
    code = r'''
import sys, time, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import raptor_amd.l2f as l2f
from test_gpu_resident import _loop, _same
dev = l2f.Device(0)
pause = lambda it, L: time.sleep(0.001) if it %% 23 == 3 else None     # (>= 8 commands per kernel: no back-off, see rq_objects.hpp)
on = _loop(dev, 8, 400, True, pause)
off = _loop(dev, 8, 400, False, pause)
print("STATS", on[6])
assert _same(off, on), "results differ"
assert on[6]["replays"] >= 5, on[6]
''' % (ROOT, os.path.join(ROOT, "tests"))
    env = dict(os.environ, RQ_RESIDENT_IDLE_TICKS="2000", RQ_RESIDENT_HOST_IDLE_NS="100000000000", RQ_RESIDENT_HOST_LIFE_NS="100000000000")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=240, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]


def test_the_loop_at_the_references_batch_beats_its_launches(device):
    """Wall time per iteration of the README loop at 8 envs with NumPy arrays at every call: the executor at least a third faster
    than the launches it replaces (measured: 7.8 against 19.5 us; the oracle's native CPU loop: 6.9)."""
    def timed(resident):
        t = []
        for _ in range(3):
            t0 = time.perf_counter()
            _loop(device, 8, 1500, resident)
            t.append((time.perf_counter() - t0) / 1500 * 1e6)
        return min(t)
    off, on = timed(False), timed(True)
    print(f"[README loop, 8 envs, NumPy arrays] launches {off:.2f} us, resident executor {on:.2f} us per iteration")
    assert on < 0.67 * off, (on, off)


# ------------------------------------------------------------------ the policy alone (README.md:17-25) ------
def _policy_loop(device, batch, steps, resident, between=None, seed=0, wide=False, policy=None):
    """`policy.evaluate_step(observation)` again and again, as a caller with a simulator of its own does (README.md:20-24)
    -> (actions [steps, batch, 4], final hidden state, resident statistics)."""
    from raptor_amd.foundation_policy import Raptor
    device.set_resident(resident)
    if policy is None:
        policy = Raptor(device)
    policy.reset()
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((steps, batch, 26 if wide else 22)).astype(np.float32)
    before = device.resident()
    A = []
    for t in range(steps):
        A.append(policy.evaluate_step(X[t][:, :22]).copy())       # (wide: rows of 26 floats, the README's observation[:, :22])
        if between is not None:
            between(t, policy)
    after = device.resident()
    out = (np.array(A), policy.hidden_state(batch).copy(), {k: after[k] - before[k] for k in ("starts", "commands", "replays")})
    device.set_resident(True)
    return out


@pytest.mark.parametrize("batch,wide", [(1, False), (1, True), (2, True), (3, False), (8, True), (16, False)])
def test_the_policy_alone_is_bit_identical_with_and_without_the_resident_executor(device, batch, wide):
    """400 recurrent steps on random rows: every action and the hidden state at the end, with the rows riding in the poll (batch <= 2)
    and fetched behind the line (up to 16), compact rows and rows of a wider array."""
    off = _policy_loop(device, batch, 400, False, wide=wide)
    on = _policy_loop(device, batch, 400, True, wide=wide)
    assert np.array_equal(off[0].view(np.uint32), on[0].view(np.uint32)) and np.array_equal(off[1].view(np.uint32), on[1].view(np.uint32))
    assert off[2]["commands"] == 0 and on[2]["commands"] >= 390 and on[2]["replays"] == 0, (off[2], on[2])


def test_more_than_sixteen_rows_keep_their_launch(device):
    on = _policy_loop(device, 17, 30, True)
    assert on[2]["commands"] == 0 and on[2]["starts"] == 0


def test_the_references_known_answers_through_the_resident_policy_executor(device):
    """checkpoint.h:197-215 and h5:/example: 500 recurrent steps at batch 2, the C selftest's tight loop - which the executor takes over
    from its third step on - against the reference's own outputs, 1e-5."""
    from raptor_amd.foundation_policy import Raptor
    gold = os.path.join(ROOT, "tests", "golden")
    policy = Raptor(device)
    for tag in ("h", "h5"):
        x = np.fromfile(os.path.join(gold, f"kat_{tag}_input.bin"), "<f4").reshape(500, 2, 22)
        y = np.fromfile(os.path.join(gold, f"kat_{tag}_output.bin"), "<f4").reshape(500, 2, 4)
        before = device.resident()["commands"]
        err = policy.selftest(x, y, tolerance=1e-5)
        assert err < 1e-5 and device.resident()["commands"] - before >= 490, (err, device.resident())
        assert not device.resident()["running"]          # the selftest's private policy is gone, and its kernel with it


def test_whatever_else_touches_the_policy_or_the_device_retires_the_policy_executor_first(device, oracle):
    """reset(), set_hidden_state(), reading the hidden state, another policy on the same device, a README-loop iteration, a large
    rollout in between: same results as with launches, and no kernel left on the device afterwards."""
    from raptor_amd.foundation_policy import Raptor
    other = Raptor(device)
    big = World(device, oracle, 4096, seed=3)

    def between(t, policy):
        if t == 30:
            assert device.resident()["running"] == device.resident()["enabled"]
            policy.hidden_state(4)
            assert not device.resident()["running"]
        if t == 60:
            policy.reset()
        if t == 90:
            h = policy.hidden_state(4)
            policy.set_hidden_state(h * np.float32(0.5))
        if t == 120:
            other.reset()
            other.evaluate_step(np.ones((4, 22), np.float32))
        if t == 150:
            big.vector.rollout(device, big.env, big.params, big.state, big.policy, big.rng, 5, "fused", True)
        if 180 <= t < 200:
            time.sleep(0.0005)                           # a caller that has something else to do between its steps
    off = _policy_loop(device, 4, 260, False, between)
    big = World(device, oracle, 4096, seed=3)
    other = Raptor(device)
    on = _policy_loop(device, 4, 260, True, between)
    assert np.array_equal(off[0].view(np.uint32), on[0].view(np.uint32)) and np.array_equal(off[1].view(np.uint32), on[1].view(np.uint32))
    assert on[2]["starts"] >= 4 and on[2]["replays"] == 0, on[2]
    device.synchronize()
    assert not device.resident()["running"]


def test_the_policy_alone_and_the_readme_loop_take_turns(device):
    """One device, one resident kernel at a time, of either kind: a README loop, then the same policy object evaluated alone, then the
    loop again - each phase bit-identical to itself without the executor."""
    from raptor_amd.foundation_policy import Raptor

    def phases(resident):
        out = []
        out.append(_loop(device, 8, 60, resident)[:6])
        pol = Raptor(device)
        out.append(_policy_loop(device, 8, 60, resident, policy=pol)[:2])
        out.append(_loop(device, 8, 60, resident, seed=1)[:6])
        out.append(_policy_loop(device, 8, 60, resident, policy=pol, seed=1)[:2])
        return out
    off, on = phases(False), phases(True)
    for a, b in zip(off, on):
        assert all(np.array_equal(np.ascontiguousarray(x).view(np.uint8), np.ascontiguousarray(y).view(np.uint8)) for x, y in zip(a, b))


@pytest.mark.timeout(300)
def test_a_policy_command_the_kernel_never_took_is_replayed_as_a_launch():
    """Kernels that leave after 20 us of idling and a host that keeps posting to them regardless, with pauses: the rows posted into
    the void are evaluated by a launch, and nothing differs."""
    code = r'''
import sys, time, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import raptor_amd.l2f as l2f
from test_gpu_resident import _policy_loop
dev = l2f.Device(0)
pause = lambda t, p: time.sleep(0.001) if t %% 23 == 3 else None
on = _policy_loop(dev, 2, 400, True, pause)
off = _policy_loop(dev, 2, 400, False, pause)
print("STATS", on[2])
assert np.array_equal(off[0].view(np.uint32), on[0].view(np.uint32)) and np.array_equal(off[1].view(np.uint32), on[1].view(np.uint32)), "results differ"
assert on[2]["replays"] >= 5, on[2]
''' % (ROOT, os.path.join(ROOT, "tests"))
    env = dict(os.environ, RQ_RESIDENT_IDLE_TICKS="2000", RQ_RESIDENT_HOST_IDLE_NS="100000000000", RQ_RESIDENT_HOST_LIFE_NS="100000000000")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=240, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]


def test_the_policy_alone_at_batch_one_beats_its_launch(device):
    """Wall time per `policy.evaluate_step(observation)` at batch 1 through the Python binding (measured: 5 against 14 us)."""
    def timed(resident):
        t = []
        for _ in range(3):
            t0 = time.perf_counter()
            _policy_loop(device, 1, 2000, resident)
            t.append((time.perf_counter() - t0) / 2000 * 1e6)
        return min(t)
    off, on = timed(False), timed(True)
    print(f"[policy alone, batch 1, NumPy arrays] launch {off:.2f} us, resident executor {on:.2f} us per call")
    assert on < 0.67 * off, (on, off)


def test_two_policies_evaluated_in_turns_keep_their_launches(device):
    """A student and a teacher on the same rows, call after call: neither is "called again and again" - no resident kernel is started
    for one only to be retired by the other (that would cost both a kernel start per call)."""
    from raptor_amd.foundation_policy import Raptor
    a, b = Raptor(device), Raptor(device)
    a.reset(); b.reset()
    device.set_resident(True)
    X = np.random.default_rng(1).standard_normal((200, 4, 22)).astype(np.float32)
    before = device.resident()
    for t in range(200):
        ya, yb = a.evaluate_step(X[t]), b.evaluate_step(X[t])
        assert np.array_equal(ya.view(np.uint32), yb.view(np.uint32))        # same weights, same rows, same history
    after = device.resident()
    assert after["starts"] == before["starts"] and after["commands"] == before["commands"]
