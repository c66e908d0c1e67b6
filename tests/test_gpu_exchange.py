"""the path's one exchange - all-gather of episode returns - with the real librccl on one rank, a tests-only RCCL on
several ranks sharing the GPU, and bench.py's own multi-rank orchestration on the GPU.

Parity of the HIP path (through the C ABI of libraptor_quad.so) against the oracle.  Bars (DESIGN.md "Parity"):
  * integer / index / mask work, parameter sampling, observe (no noise) and env transitions for identical inputs: BIT-EXACT;
  * anything behind a transcendental (actor gates, sin/cos of the initial attitude, Box-Muller noise): float32 tolerance stated per test;
  * the actor additionally against the reference's own known-answer vectors (< 1e-5).
(Round 6 split tests/test_gpu_parity.py - 2 987 lines, one module - by SURVEY.md section 8 row group, so that a red run names its row.)
"""
import os

import numpy as np
import pytest

from gpu_common import World      # noqa: F401

pytestmark = pytest.mark.gpu

def test_overlapped_returns_exchange_on_the_gpu(device, oracle):
    """ReturnsExchange on the real streams: finished returns are copied on the engine's HIP stream without a
    host wait, the (1-rank RCCL) all-gather runs on a side stream behind an event; after finish() the gathered
    tensor equals the synchronous getter - also after the double buffers were recycled."""
    import socket
    import torch
    import torch.distributed as dist
    from raptor_amd.distributed import ReturnsExchange
    n = 4096
    w = World(device, oracle, n, seed=21, episode_step_limit=7)
    torch.cuda.set_device(0)
    for with_group in (False, True):
        if with_group:
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                port = s.getsockname()[1]
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                    device_id=torch.device("cuda", 0))
        try:
            ex = ReturnsExchange(n, n, "cuda:0", engine_stream=device.stream)
            for k in range(5):
                w.vector.rollout(device, w.env, w.params, w.state, w.policy, w.rng, 7, "fused", autoreset=True)
                ex.post(lambda buf: w.env.finished_returns(out=buf, wait=False))
            got = ex.finish().cpu().numpy()
            assert np.array_equal(got, w.env.finished_returns()) and np.any(got != 0)
        finally:
            if with_group:
                dist.destroy_process_group()


# ------------------------------------------------------------------------------ native RCCL exchange -
def test_native_rccl_exchange_one_rank(device, oracle):
    """rq_comm_* / rq_allgather_returns with a 1-rank RCCL communicator created by the C++ host itself: the
    all-gather of episode k is enqueued behind rollout k and overlaps rollout k + 1; what comes back is the
    env's finished returns of the episode it was posted after (double buffering keeps them apart)."""
    from raptor_amd.distributed import NativeReturnsExchange
    w = World(device, oracle, 4096, seed=41, episode_step_limit=20)
    ex = NativeReturnsExchange(device, 1, 0, NativeReturnsExchange.unique_id())
    assert ex.info() == (1, 0)
    d = ex.describe()               # asked of RCCL and the HIP runtime (round 5), not echoed from the arguments above
    assert d["ranks"] == 1 and d["rank"] == 0 and d["device"] == 0 and d["collectives_posted"] == 0
    assert d["version_code"] > 20000 and d["version"].count(".") == 2, d          # a real RCCL: 2.x.y
    assert "rccl" in os.path.basename(d["library_path"]).lower() and os.path.exists(d["library_path"]), d
    assert len(d["pci_bus_id"]) >= 7 and d["pci_bus_id"].count(":") == 2, d
    snaps = []
    for k in range(5):
        w.vector.rollout(device, w.env, w.params, w.state, w.policy, w.rng, 20, "fused", True)
        ex.post(w.env)
        if k % 2 == 1:          # not every episode is read back: the un-read ones must not leak into later results
            device.synchronize()
            snaps.append((w.env.finished_returns().copy(), ex.finish()))
    for fin, got in snaps:
        assert got.shape == (4096,) and np.array_equal(got, fin)
    ptr, count = ex.finish(to_host=False)
    assert count == 4096 and ptr
    with pytest.raises(Exception):
        NativeReturnsExchange(device, 2, 5, NativeReturnsExchange.unique_id())      # rank out of range


def _native_exchange_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)          # rendezvous only: ships the id
    try:
        import raptor_amd.l2f as l2f
        from raptor_amd.distributed import NativeReturnsExchange
        from raptor_amd.foundation_policy import Raptor
        n = 2048
        dev = l2f.Device(rank)
        v = l2f.VectorModule(n, rank * n)
        rng, env, params, state = v.VectorRng(), v.VectorEnvironment(), v.VectorParameters(), v.VectorState()
        v.initialize_rng(dev, rng, 9)
        v.initialize_environment(dev, env)
        cfg = env.config
        cfg.episode_step_limit = 30
        env.config = cfg
        v.sample_initial_parameters(dev, env, params, rng)
        v.sample_initial_state(dev, env, params, state, rng)
        pol = Raptor(dev)
        ident = [NativeReturnsExchange.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ident, src=0)
        ex = NativeReturnsExchange(dev, world, rank, ident[0])
        v.rollout(dev, env, params, state, pol, rng, 30, "fused", True)
        ex.post(env)
        q.put((rank, env.finished_returns().copy(), ex.finish()))
    finally:
        dist.destroy_process_group()


def test_exchange_beside_saturating_rollouts_probe(device):
    """bench.py's `native_exchange_1rank` block (round 4) at a small size: the real librccl with one rank, an exchange posted
    after every 500-step launch with the next launch enqueued behind it - the record's fields are there and sane, and what was
    gathered is the env's own finished returns."""
    import bench

    class Args:
        precision = "fp32"
    eng = bench.GpuEngine(0, Args())
    eng.device = device
    try:
        rec = bench.native_exchange_probe(eng, 4096, launches=2, repeats=2)
    except Exception as exc:      # noqa: BLE001
        if "rccl" in str(exc).lower():
            pytest.skip(f"no RCCL to bind: {exc}")
        raise
    assert rec["envs"] == 4096 and rec["gathered_returns"] == 4096 and rec["bytes_per_rank"] == 16384
    assert rec["exchange_verified"] is True and rec["rccl"]["ranks"] == 1 and rec["rccl"]["version_code"] > 20000      # RCCL's own account
    assert rec["us_per_episode_without_exchange"] > 100 and rec["us_per_episode_with_exchange"] > 100
    assert abs(rec["added_fraction"]) < 0.5 and abs(rec["rollout_slowdown_fraction"]) < 0.2
    assert 1.0 < rec["exchange_alone_us_post_to_gathered"] < 5000 and 0.5 < rec["post_call_host_us"] < 1000


@pytest.mark.timeout(300)
def test_native_rccl_exchange_across_gpus():
    """Two processes, two GPUs, RCCL over xGMI from the C++ host: every rank ends up with the concatenation of the
    ranks' finished returns in global env order.  Needs >= 2 GPUs (the 1-GPU box skips it)."""
    import socket
    import raptor_amd.l2f as l2f
    if l2f.Device.count() < 2:
        pytest.skip("needs two GPUs")
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_native_exchange_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(60)
    full = np.concatenate([res[0][1], res[1][1]])
    assert np.array_equal(res[0][2], full) and np.array_equal(res[1][2], full)


def _shared_gpu_exchange_worker(rank, world, n, episodes, fake_lib, conn):
    """One rank of test_native_exchange_two_ranks_on_one_gpu: a process of its own on GPU 0, RCCL = the tests-only
    fake (tests/fake_rccl.cpp, shared memory between the processes), selected before the library binds RCCL."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["RQ_RCCL_LIBRARY"] = fake_lib
    try:
        import raptor_amd.l2f as l2f
        from raptor_amd.distributed import NativeReturnsExchange
        from raptor_amd.foundation_policy import Raptor
        dev = l2f.Device(0)
        v = l2f.VectorModule(n, rank * n)
        rng, env, params, state = v.VectorRng(), v.VectorEnvironment(), v.VectorParameters(), v.VectorState()
        v.initialize_rng(dev, rng, 9)
        v.initialize_environment(dev, env)
        cfg = env.config
        cfg.episode_step_limit = 30
        env.config = cfg
        v.sample_initial_parameters(dev, env, params, rng)
        v.sample_initial_state(dev, env, params, state, rng)
        pol = Raptor(dev)
        if rank == 0:
            ident = NativeReturnsExchange.unique_id()
            conn.send(("id", ident))
        ident = conn.recv()                                   # the parent relays rank 0's id to every rank
        ex = NativeReturnsExchange(dev, world, rank, ident)
        assert ex.info() == (world, rank)
        d = ex.describe()
        assert (d["ranks"], d["rank"], d["version_code"]) == (world, rank, 0) and d["library_path"] == fake_lib, d      # the stand-in says so itself
        snaps = []
        for k in range(episodes):
            # no host synchronisation between posts: the copy of episode k's returns sits on the engine's stream behind
            # rollout k, the collective on the side stream; episode k + 1 is enqueued right behind
            v.rollout(dev, env, params, state, pol, rng, 30, "fused", True)
            ex.post(env)
            if k in (1, episodes - 1):                        # read back twice: after the buffers were recycled, too
                dev.synchronize()
                snaps.append((k, env.finished_returns().copy(), ex.finish()))
        conn.send(("done", snaps))
    except Exception as exc:      # noqa: BLE001
        import traceback
        conn.send(("error", f"rank {rank}: {exc}\n{traceback.format_exc()}"))


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 4])
def test_native_exchange_two_ranks_on_one_gpu(device, tmp_path, world):
    """rq_comm_create / rq_allgather_returns / rq_comm_gathered with n_ranks = 2 (round 3): two processes share the one
    GPU of this box and bind a tests-only RCCL (tests/fake_rccl.cpp: all-gather = device->host copy, a host function
    in the stream that meets the other rank in shared memory, host->device copy - enqueued on the stream the product
    hands it, completing in stream order like the real one).  Exercised with two ranks for the first time: the
    communicator creation as a collective, the double-buffered send / receive pairs across seven posts, the event
    ordering between the engine's stream and the side stream, and the GLOBAL env order of the result - which must
    equal the finished returns of the same 2 n envs rolled out unsharded (RNG keyed by global id)."""
    import shutil
    import subprocess
    import multiprocessing as mp
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    fake = str(tmp_path / "libfake_rccl.so")
    subprocess.run([hipcc, "-shared", "-fPIC", "-O2", "-std=c++17", os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                                                                 "fake_rccl.cpp"), "-o", fake, "-lrt"],
                   check=True, capture_output=True)
    n, episodes = 4096, 7
    ctx = mp.get_context("spawn")
    pipes = [ctx.Pipe() for _ in range(world)]
    procs = [ctx.Process(target=_shared_gpu_exchange_worker, args=(r, world, n, episodes, fake, pipes[r][1])) for r in range(world)]
    for pr in procs:
        pr.start()
    try:
        assert pipes[0][0].poll(240), "rank 0 produced no communicator id"
        kind, ident = pipes[0][0].recv()
        assert kind == "id", ident
        assert ident.startswith(b"/rqfake_"), "the product bound another RCCL than the one RQ_RCCL_LIBRARY names"
        for r in range(world):
            pipes[r][0].send(ident)
        res = []
        for r in range(world):
            assert pipes[r][0].poll(300), f"rank {r} hung (a rank left waiting in the collective)"
            kind, payload = pipes[r][0].recv()
            assert kind == "done", payload
            res.append(payload)
    finally:
        for pr in procs:
            pr.join(30)
            if pr.is_alive():
                pr.kill()
    # the unsharded batch on this process's own device: same seed, same config, global ids 0 .. 2n - 1
    import raptor_amd.l2f as l2f
    from raptor_amd.foundation_policy import Raptor
    v = l2f.VectorModule(world * n, 0)
    rng, env, params, state = v.VectorRng(), v.VectorEnvironment(), v.VectorParameters(), v.VectorState()
    v.initialize_rng(device, rng, 9)
    v.initialize_environment(device, env)
    cfg = env.config
    cfg.episode_step_limit = 30
    env.config = cfg
    v.sample_initial_parameters(device, env, params, rng)
    v.sample_initial_state(device, env, params, state, rng)
    pol = Raptor(device)
    whole = {}
    for k in range(episodes):
        v.rollout(device, env, params, state, pol, rng, 30, "fused", True)
        if k in (1, episodes - 1):
            whole[k] = env.finished_returns().copy()
    for i in range(2):
        k = res[0][i][0]
        local = np.concatenate([res[r][i][1] for r in range(world)])
        for r in range(world):
            got = res[r][i][2]
            assert got.shape == (world * n,)
            assert np.array_equal(got, local), f"rank {r}, episode {k}: gathered != concatenation of the ranks' returns"
        assert np.array_equal(local, whole[k]), f"episode {k}: sharded returns differ from the unsharded batch"


@pytest.mark.timeout(900)
def test_bench_py_with_two_ranks_on_one_gpu(device, tmp_path):
    """bench.py itself with WORLD_SIZE = 2 on the GPU (round 3): the product engine (GpuEngine: libraptor_quad.so), the
    two-phase consensus, the NATIVE exchange (rq_comm_* bound to tests/fake_rccl.cpp through RQ_RCCL_LIBRARY, both ranks
    on this box's one GPU through RQ_BENCH_DEVICE), 20-step regions with their share of the all-gather, the 262 144-envs
    block - what `torch.distributed.run --nproc-per-node N bench.py --gpus N` meets on a multi-GPU node, minus xGMI.
    torch.distributed only rendezvouses (gloo)."""
    import json
    import shutil
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    fake = str(tmp_path / "libfake_rccl.so")
    subprocess.run([hipcc, "-shared", "-fPIC", "-O2", "-std=c++17", os.path.join(root, "tests", "fake_rccl.cpp"), "-o", fake, "-lrt"],
                   check=True, capture_output=True)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   RQ_BENCH_DEVICE="0", RQ_RCCL_LIBRARY=fake)
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "20",
                                       "--warmup", "5", "--no-cpu-baseline", "--envs-per-gpu", "8192"], env=env, cwd=root,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            out, err = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("bench.py --gpus 2 hung")
        assert p.returncode == 0, err[-3000:]
        outs.append(out)
    records = [l for l in outs[0].strip().split("\n") if l.lstrip().startswith("{")]
    assert len(records) == 1 and outs[0].strip().split("\n")[-1] == records[0] and "{" not in outs[1]
    d = json.loads(records[0])
    assert d["n_gpus"] == 2 and d["config"]["total_envs"] == 16384 and d["config"]["engine"] == "hip"
    assert d["config"]["exchange"].startswith("native RCCL"), d["config"]["exchange"]
    assert d["config"]["gathered_returns"] == 16384
    assert d["config"]["exchange_verified"] is True and d["config4"]["exchange_verified"] is True          # round 5: layout checked before a value is printed
    assert d["config"]["rccl"]["ranks"] == 2 and d["config"]["rccl"]["library_path"] == fake and d["config"]["rccl"]["version_code"] == 0
    assert [r["rank"] for r in d["config"]["rccl"]["per_rank"]] == [0, 1]
    assert d["config"]["rccl"]["distinct_gpus"] == 1        # both ranks of THIS test share the box's one GPU, and the record shows it
    assert d["timing"]["exchange_share"]["regions_with_extra_exchange"] >= 3
    assert d["steady_state"]["exchanges"] == 10 and d["config4"]["total_envs"] == 2 * 262144 and d["config4"]["exchanges"] == 4
    assert d["value"] > 1e8 and d["roofline"]["frac"] > 0.01
    print(f"[bench.py, 2 ranks on one GPU, fake RCCL] value {d['value']:.3g} env-steps/s, region {d['timing']['region_ms']['charged']:.4f} ms, "
          f"exchange share {d['timing']['exchange_share']}")


@pytest.mark.timeout(600)
def test_the_drivers_eight_rank_command_end_to_end_on_one_gpu(device, tmp_path):
    """What the driver's first SCALE run will execute - `python bench.py --gpus 8 --steps 20 --warmup 5`, eight ranks, the product
    engine on the GPU - has only ever run through the gloo stand-in engine on CPU (tests/test_bench_orchestration.py) and with two
    ranks here.  Round 6: all eight ranks on this box's one GPU (`--allow-oversubscribe`, tests only; the native
    exchange bound to tests/fake_rccl.cpp), typed the way the driver types it when there is no launcher: consensus on the communicator,
    the verification episode, the timed regions with their share of the all-gather, the config-4 block (8 x 262 144 = 2 097 152 envs),
    ONE JSON line - inside the driver's time box: the test fails above 120 s of wall time."""
    import json
    import shutil
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    fake = str(tmp_path / "libfake_rccl.so")
    subprocess.run([hipcc, "-shared", "-fPIC", "-O2", "-std=c++17", os.path.join(root, "tests", "fake_rccl.cpp"), "-o", fake, "-lrt"],
                   check=True, capture_output=True)
    env = dict(os.environ, RQ_RCCL_LIBRARY=fake)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "RQ_BENCH_DEVICE"):
        env.pop(k, None)
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5", "--allow-oversubscribe"],
                       env=env, cwd=root, capture_output=True, text=True, timeout=500)
    wall = time.perf_counter() - t0
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().split("\n") if l.strip()]
    records = [l for l in lines if l.lstrip().startswith("{")]
    assert len(records) == 1 and lines[-1] == records[0], lines[-3:]
    d = json.loads(records[0])
    assert d["n_gpus"] == 8 and d["config"]["total_envs"] == 8 * 65536 and d["config"]["engine"] == "hip" and "oversubscribed" in d["config"]
    assert d["config"]["exchange"].startswith("native RCCL"), d["config"]["exchange"]
    assert d["config"]["gathered_returns"] == 8 * 65536
    assert d["config"]["exchange_verified"] is True and d["config"]["exchange_check"]["blocks_matching_their_rank_on_rank0"] == 8
    rccl = d["config"]["rccl"]
    assert rccl["ranks"] == 8 and [x["rank"] for x in rccl["per_rank"]] == list(range(8)) and rccl["library_path"] == fake and rccl["distinct_gpus"] == 1
    assert d["config4"]["total_envs"] == 2097152 and d["config4"]["exchange_verified"] is True and d["config4"]["exchanges"] == 4      # BASELINE config 4
    assert d["timing"]["exchange_share"]["regions_with_extra_exchange"] >= 3
    assert d["value"] > 1e8 and d["scaling"] == "weak" and d["metric"].startswith("env-steps/sec (whole node) at 65536 quadrotors")
    out = os.path.join(root, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "eight_ranks_one_gpu.json"), "w") as fh:
            json.dump({"wall_s": round(wall, 1), "command": "python bench.py --gpus 8 --steps 20 --warmup 5 --allow-oversubscribe", "record": d}, fh, indent=1)
    print(f"[bench.py --gpus 8 on one GPU, fake RCCL] wall {wall:.1f} s, value {d['value']:.3g} env-steps/s (8 ranks time-share one GPU)")
    assert wall < 120.0, f"the eight-rank command took {wall:.0f} s on one GPU: over the 120 s box"


@pytest.mark.timeout(900)
def test_plain_bench_command_launches_its_own_ranks_on_the_gpu(device, tmp_path):
    """`python bench.py --gpus 2` typed as is - no RANK / WORLD_SIZE, no torch.distributed.run around it (round 4): the command
    re-executes itself as two local ranks (free port, LOCAL_RANK = rank), rank 0's record is the one line on the launcher's
    stdout and says n_gpus 2.  Both ranks share this box's one GPU (RQ_BENCH_DEVICE) with the tests-only RCCL; without
    RQ_BENCH_DEVICE the same command is an ERROR on a one-GPU box, not a silent one-rank run."""
    import json
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    fake = str(tmp_path / "libfake_rccl.so")
    subprocess.run([hipcc, "-shared", "-fPIC", "-O2", "-std=c++17", os.path.join(root, "tests", "fake_rccl.cpp"), "-o", fake, "-lrt"],
                   check=True, capture_output=True)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "20", "--warmup", "5",
           "--no-cpu-baseline", "--no-config4", "--envs-per-gpu", "8192"]
    out = subprocess.run(cmd, env=dict(env, RQ_BENCH_DEVICE="0", RQ_RCCL_LIBRARY=fake), cwd=root, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.strip().split("\n") if l.strip()]
    records = [l for l in lines if l.lstrip().startswith("{")]
    assert len(records) == 1 and lines[-1] == records[0], lines[-5:]
    d = json.loads(records[0])
    assert d["n_gpus"] == 2 and d["config"]["total_envs"] == 16384 and d["config"]["engine"] == "hip"
    assert d["config"]["exchange"].startswith("native RCCL") and d["config"]["gathered_returns"] == 16384
    from raptor_amd import _lib
    import ctypes
    count = ctypes.c_int(0)
    _lib.call("rq_device_count", ctypes.byref(count))
    if count.value < 2:
        out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=300)
        assert out.returncode != 0 and f"this node has {count.value} GPU(s)" in out.stderr and "{" not in out.stdout


def test_native_exchange_resizes_with_the_env(device, oracle):
    """One communicator serving envs of different sizes in turn (buffers are re-sized, results never mix)."""
    from raptor_amd.distributed import NativeReturnsExchange
    ex = NativeReturnsExchange(device, 1, 0, NativeReturnsExchange.unique_id())
    for n in (100, 5000, 64):
        w = World(device, oracle, n, seed=60 + n, episode_step_limit=10)
        w.vector.rollout(device, w.env, w.params, w.state, w.policy, w.rng, 10, "fused", True)
        ex.post(w.env)
        got = ex.finish()
        assert got.shape == (n,) and np.array_equal(got, w.env.finished_returns())
