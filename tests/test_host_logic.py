import os

import numpy as np
import pytest

from raptor_amd.distributed import shard_range, shard_sizes


@pytest.mark.parametrize("n,w", [(8, 1), (8, 2), (65536, 8), (2097152, 8), (10, 3), (7, 8), (1, 4)])
def test_shard_range_partitions_exactly(n, w):
    covered = []
    for r in range(w):
        s, c = shard_range(n, w, r)
        covered.extend(range(s, s + c))
    assert covered == list(range(n))
    sizes = shard_sizes(n, w)
    assert max(sizes) - min(sizes) <= 1 and sum(sizes) == n


def test_shard_range_rejects_bad_rank():
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def test_policy_weights_file():
    from raptor_amd.foundation_policy import load_weights
    w = load_weights()
    assert w.dtype == np.float32 and w.size == 2084 and np.isfinite(w).all()


def test_checkpoint_header_round_trip(tmp_path):
    """The reference's C++ export format (byte-array blobs per namespace) written and parsed back."""
    from raptor_amd.checkpoint import load_checkpoint_header, write_checkpoint_header
    from raptor_amd.foundation_policy import load_weights
    w = load_weights()
    x = np.random.default_rng(0).standard_normal((5, 2, 22)).astype(np.float32)
    y = np.random.default_rng(1).standard_normal((5, 2, 4)).astype(np.float32)
    path = tmp_path / "checkpoint.h"
    write_checkpoint_header(path, w, (x, y))
    w2, ex = load_checkpoint_header(path)
    assert np.array_equal(w, w2)
    assert ex is not None and np.array_equal(ex[0], x) and np.array_equal(ex[1], y)
    bad = tmp_path / "bad.h"
    bad.write_text("namespace rl_tools::checkpoint::actor {\n}\n")
    with pytest.raises(ValueError):
        load_checkpoint_header(bad)


def test_checkpoint_header_of_the_reference(tmp_path):
    """When the reference tarball is present (build container only), its checkpoint.h parses to the
    shipped weights and carries the first known-answer vector."""
    import os
    import tarfile
    tar = "/root/reference/data/raptor-policy-checkpoint.tar.gz"
    if not os.path.exists(tar):
        pytest.skip("reference not present on this box")
    from raptor_amd.checkpoint import load_checkpoint_header
    from raptor_amd.foundation_policy import load_weights
    from conftest import GOLDEN
    with tarfile.open(tar) as tf:
        tf.extract("2025-04-19_16-16-17/checkpoint.h", tmp_path)
    w, ex = load_checkpoint_header(tmp_path / "2025-04-19_16-16-17" / "checkpoint.h")
    assert np.array_equal(w, load_weights())
    assert ex[0].shape == (500, 2, 22)
    assert np.array_equal(ex[0].ravel(), np.fromfile(os.path.join(GOLDEN, "kat_h_input.bin"), "<f4"))
    assert np.array_equal(ex[1].ravel(), np.fromfile(os.path.join(GOLDEN, "kat_h_output.bin"), "<f4"))


def test_hdf5_checkpoint_reader(oracle):
    """The published checkpoint.h5 (fixture) through the dependency-free HDF5 reader: same weights
    as the shipped parameter file, the second known-answer pair, the observation spec string, and
    the oracle reproduces that pair from these weights."""
    import os
    from conftest import GOLDEN
    from raptor_amd.checkpoint import load_checkpoint, load_checkpoint_h5
    from raptor_amd.foundation_policy import load_weights
    path = os.path.join(GOLDEN, "checkpoint.h5")
    w, ex, meta = load_checkpoint_h5(path)
    assert np.array_equal(w, load_weights())
    assert "OrientationRotationMatrix" in meta and "ActionHistory(1)" in meta
    assert np.array_equal(ex[0].ravel(), np.fromfile(os.path.join(GOLDEN, "kat_h5_input.bin"), "<f4"))
    assert np.array_equal(ex[1].ravel(), np.fromfile(os.path.join(GOLDEN, "kat_h5_output.bin"), "<f4"))
    assert np.abs(oracle.actor_sequence(w, ex[0]) - ex[1]).max() < 1e-5
    w2, ex2 = load_checkpoint(path)
    assert np.array_equal(w2, w) and ex2[0].shape == (500, 2, 22)


def test_hdf5_reader_rejects_garbage(tmp_path):
    from raptor_amd.hdf5_min import File, Hdf5FormatError
    p = tmp_path / "x.h5"
    p.write_bytes(b"not hdf5 at all")
    with pytest.raises(Hdf5FormatError):
        File(p)


def _hdf5_tool(name):
    import shutil
    for cand in (shutil.which(name), f"/opt/conda/bin/{name}"):
        if cand and os.path.exists(cand):
            return cand
    return None


def test_hdf5_checkpoint_writer_round_trip(tmp_path):
    """write_checkpoint_h5 -> load_checkpoint_h5 is the identity (weights, example, meta), a perturbed policy
    round-trips too, and - when libhdf5's own tools are installed - h5diff finds no difference between the
    reference's checkpoint.h5 and the file this writer produces from its contents (datasets AND attributes)."""
    import subprocess
    from conftest import GOLDEN
    from raptor_amd.checkpoint import load_checkpoint_h5, write_checkpoint_h5
    ref = os.path.join(GOLDEN, "checkpoint.h5")
    w, ex, meta = load_checkpoint_h5(ref)
    out = str(tmp_path / "written.h5")
    write_checkpoint_h5(out, w, ex, meta=meta, checkpoint_name="logs/2025-04-19_16-16-17")
    w2, ex2, meta2 = load_checkpoint_h5(out)
    assert np.array_equal(w2, w) and meta2 == meta
    assert np.array_equal(ex2[0], ex[0]) and np.array_equal(ex2[1], ex[1])
    h5diff = _hdf5_tool("h5diff")
    if h5diff:
        r = subprocess.run([h5diff, "-c", ref, out], capture_output=True, text=True)
        assert r.returncode == 0 and "not comparable" not in r.stdout, r.stdout + r.stderr
    h5dump = _hdf5_tool("h5dump")
    if h5dump:
        raw = str(tmp_path / "wh.bin")
        subprocess.run([h5dump, "-d", "/actor/layers/1/weights_hidden/parameters", "-b", "LE", "-o", raw, out],
                       check=True, stdout=subprocess.DEVNULL)
        assert np.array_equal(np.fromfile(raw, "<f4"), w[352 + 16 + 768:352 + 16 + 1536])
    # a different policy, no example, default meta
    rng = np.random.default_rng(3)
    wp = (w + rng.standard_normal(w.size).astype(np.float32) * 0.01).astype(np.float32)
    out2 = str(tmp_path / "perturbed.h5")
    write_checkpoint_h5(out2, wp)
    w3, ex3, meta3 = load_checkpoint_h5(out2)
    assert np.array_equal(w3, wp) and ex3 is None and "l2f" in meta3
    with pytest.raises(ValueError):
        write_checkpoint_h5(out2, wp[:-1])


def test_hdf5_writer_limits(tmp_path):
    from raptor_amd.hdf5_min import DatasetSpec, File, GroupSpec, Hdf5FormatError, write_file
    with pytest.raises(Hdf5FormatError):
        write_file(str(tmp_path / "a.h5"), GroupSpec({}))                       # empty group
    many = GroupSpec({f"d{i}": DatasetSpec(np.zeros(2, np.float32)) for i in range(9)})
    with pytest.raises(Hdf5FormatError):
        write_file(str(tmp_path / "b.h5"), many)                                # > 8 children in one group
    ok = GroupSpec({f"d{i}": DatasetSpec(np.full((i + 1, 3), i, np.float32), {"k": "v" * (i * 40)}) for i in range(8)},
                   {"note": "éè utf-8"})
    write_file(str(tmp_path / "c.h5"), ok)
    f = File(str(tmp_path / "c.h5"))
    assert f.root.attrs["note"] == "éè utf-8"
    for i in range(8):
        assert f.root[f"d{i}"].shape == (i + 1, 3) and f.root[f"d{i}"].attrs["k"] == "v" * (i * 40)
        assert np.all(f.root[f"d{i}"].numpy() == i)


def test_raptor_save_checkpoint_both_formats(tmp_path):
    """Raptor.from_checkpoint -> save_checkpoint -> from_checkpoint without a device (the C object is only
    created on first use): weights and the known-answer example survive both formats."""
    from conftest import GOLDEN
    from raptor_amd.foundation_policy import Raptor
    p = Raptor.from_checkpoint(os.path.join(GOLDEN, "checkpoint.h5"))
    for name in ("policy.h5", "policy.h"):
        p.save_checkpoint(str(tmp_path / name))
        q = Raptor.from_checkpoint(str(tmp_path / name))
        assert np.array_equal(q._weights, p._weights)
        assert np.array_equal(q.example[0], p.example[0]) and np.array_equal(q.example[1], p.example[1])


def test_checkpoint_observation_layout_is_checked_on_load(tmp_path):
    """`h5:/actor@meta` names the observation the policy was trained on; the shipped checkpoint's is the layout this engine's observe
    assembles (README.md:23).  Round 5 read the attribute and threw it away: a checkpoint trained on another layout loaded silently."""
    from conftest import GOLDEN
    from raptor_amd import checkpoint as ck
    from raptor_amd.foundation_policy import Raptor
    ref = os.path.join(GOLDEN, "checkpoint.h5")
    w, ex, meta = ck.load_checkpoint(ref, with_meta=True)
    assert ck.observation_of_meta(meta) == ck.ENGINE_OBSERVATION                # the reference's own file states the engine's layout
    assert Raptor.from_checkpoint(ref).observation_spec == ck.ENGINE_OBSERVATION
    other = '{"environment": {"name": "l2f","observation": "Position.OrientationQuaternion.LinearVelocity.AngularVelocity.ActionHistory(4)"}}'
    ck.write_checkpoint_h5(str(tmp_path / "other.h5"), w, meta=other)
    with pytest.raises(ValueError, match="trained on observation 'Position.OrientationQuaternion"):
        Raptor.from_checkpoint(str(tmp_path / "other.h5"))
    p = Raptor.from_checkpoint(str(tmp_path / "other.h5"), check_observation=False)         # the explicit way in
    assert p.observation_spec.startswith("Position.OrientationQuaternion") and np.array_equal(p._weights, w)
    ck.write_checkpoint_h5(str(tmp_path / "garbled.h5"), w, meta="not json {")
    with pytest.raises(ValueError, match="not JSON"):
        Raptor.from_checkpoint(str(tmp_path / "garbled.h5"))
    # no statement at all (the C++ export carries name and commit only; an HDF5 file written without meta): loads, spec unknown
    ck.write_checkpoint_h5(str(tmp_path / "silent.h5"), w, meta=None)
    assert Raptor.from_checkpoint(str(tmp_path / "silent.h5")).observation_spec is None
    Raptor.from_checkpoint(ref).save_checkpoint(str(tmp_path / "policy.h"))
    assert Raptor.from_checkpoint(str(tmp_path / "policy.h")).observation_spec is None
    assert ck.observation_of_meta('{"environment": {"name": "l2f"}}') is None and ck.observation_of_meta("[1, 2]") is None


def test_fast_tanh_teachers_are_refused_by_name(tmp_path):
    """rl-tools' FAST_TANH is an approximation whose definition is not in the reference tree; mapping it onto the exact tanh (as
    round 5 did) would relabel a teacher's states with another function, silently."""
    from raptor_amd import checkpoint as ck
    from raptor_amd.hdf5_min import DatasetSpec, GroupSpec, write_file
    rng = np.random.default_rng(0)
    lay = {}
    for i, (o, n, fn) in enumerate(((8, 22, "FAST_TANH"), (4, 8, "IDENTITY"))):
        lay[str(i)] = GroupSpec({"weights": GroupSpec({"parameters": DatasetSpec(rng.standard_normal((o, n)).astype(np.float32), {})}),
                                 "biases": GroupSpec({"parameters": DatasetSpec(np.zeros((1, o), np.float32), {})})},
                                {"type": "dense", "activation_function": fn})
    write_file(str(tmp_path / "t.h5"), GroupSpec({"actor": GroupSpec({"layers": GroupSpec(lay)}, {"type": "sequential"})}))
    with pytest.raises(ValueError, match="FAST_TANH"):
        ck.load_mlp_checkpoint_h5(str(tmp_path / "t.h5"))


def test_oracle_teacher_mlp_matches_numpy(oracle):
    """The oracle's MLP teacher (the checker of the teacher-bank kernels) against a float64 numpy evaluation."""
    rng = np.random.default_rng(5)
    in_dim, h1, h2, n_teachers, T, n = 22, 64, 32, 5, 3, 40
    per = h1 * in_dim + h1 + h2 * h1 + h2 + 4 * h2 + 4
    W = (rng.standard_normal((n_teachers, per)) * 0.3).astype(np.float32)
    obs = rng.standard_normal((T, n, 22)).astype(np.float32)
    ids = rng.integers(0, n_teachers, n).astype(np.uint32)
    for act, out_act, f, g in ((1, 0, lambda x: np.maximum(x, 0), lambda x: x), (2, 2, np.tanh, np.tanh)):
        got = oracle.teacher_relabel(W, in_dim, h1, h2, act, out_act, obs, ids)
        for i in range(n):
            w = W[ids[i]].astype(np.float64)
            o = 0
            W1 = w[o:o + h1 * in_dim].reshape(h1, in_dim); o += h1 * in_dim
            b1 = w[o:o + h1]; o += h1
            W2 = w[o:o + h2 * h1].reshape(h2, h1); o += h2 * h1
            b2 = w[o:o + h2]; o += h2
            W3 = w[o:o + 4 * h2].reshape(4, h2); o += 4 * h2
            b3 = w[o:o + 4]
            x = obs[:, i, :in_dim].astype(np.float64)
            ref = g(f(f(x @ W1.T + b1) @ W2.T + b2) @ W3.T + b3)
            assert np.abs(got[:, i] - ref).max() < 2e-5


def test_oracle_dense_stack_matches_numpy(oracle):
    """The oracle's generic dense stack (round 5: one to three hidden layers, widths up to 128 - the checker of
    k_teacher_relabel_layers) against a float64 numpy evaluation, and against the two-hidden-layer restatement bit for bit."""
    rng = np.random.default_rng(6)
    obs = rng.standard_normal((3, 24, 22)).astype(np.float32)
    for in_dim, widths, act, out_act, f, g in ((22, [128, 48, 112], 2, 2, np.tanh, np.tanh), (17, [96], 1, 0, lambda x: np.maximum(x, 0), lambda x: x),
                                               (22, [64, 32], 1, 2, lambda x: np.maximum(x, 0), np.tanh)):
        dims = [in_dim] + widths + [4]
        per = sum(dims[i + 1] * dims[i] + dims[i + 1] for i in range(len(dims) - 1))
        W = (rng.standard_normal((4, per)) * 0.25).astype(np.float32)
        ids = rng.integers(0, 4, 24).astype(np.uint32)
        got = oracle.mlp_relabel(W, in_dim, widths, act, out_act, obs, ids)
        for i in range(24):
            w, o = W[ids[i]].astype(np.float64), 0
            x = obs[:, i, :in_dim].astype(np.float64)
            for l in range(len(dims) - 1):
                M = w[o:o + dims[l + 1] * dims[l]].reshape(dims[l + 1], dims[l]); o += M.size
                b = w[o:o + dims[l + 1]]; o += dims[l + 1]
                x = (f if l < len(widths) else g)(x @ M.T + b)
            assert np.abs(got[:, i] - x).max() < 3e-5
        if len(widths) == 2:
            assert np.array_equal(got, oracle.teacher_relabel(W, in_dim, widths[0], widths[1], act, out_act, obs, ids))


def test_teacher_checkpoints_in_the_references_hdf5_layout(tmp_path):
    """A teacher = a `sequential` of `dense` layers in the layout of h5:/actor/layers/* (what the reference's
    extract_checkpoints.sh gathers, README.md:211-216): written by this package's dependency-free writer, read back bit for bit,
    accepted by libhdf5's h5dump (the same attribute conventions as the reference's own checkpoint.h5) - and the reference's
    student checkpoint, which holds a GRU, is refused with the reason."""
    import subprocess
    from raptor_amd.checkpoint import load_mlp_checkpoint_h5, write_mlp_checkpoint_h5
    rng = np.random.default_rng(9)
    dims = [22, 128, 64, 128, 4]
    layers = [((rng.standard_normal((dims[i + 1], dims[i])) * 0.2).astype(np.float32), rng.standard_normal(dims[i + 1]).astype(np.float32))
              for i in range(4)]
    path = str(tmp_path / "teacher_0.h5")
    write_mlp_checkpoint_h5(path, layers, ["relu", "relu", "relu", "identity"], meta='{"teacher": 0}')
    got, acts = load_mlp_checkpoint_h5(path)
    assert acts == ["relu", "relu", "relu", "identity"] and len(got) == 4
    for (W, b), (W0, b0) in zip(got, layers):
        assert np.array_equal(W, W0) and np.array_equal(b, b0)
    h5dump = _hdf5_tool("h5dump")
    if h5dump:
        raw = str(tmp_path / "w2.bin")
        subprocess.run([h5dump, "-d", "/actor/layers/2/weights/parameters", "-b", "LE", "-o", raw, path], check=True, stdout=subprocess.DEVNULL)
        assert np.array_equal(np.fromfile(raw, "<f4").reshape(128, 64), layers[2][0])
        head = subprocess.run([h5dump, "-A", path], capture_output=True, text=True, check=True).stdout
        assert '"dense"' in head and '"RELU"' in head and '"sequential"' in head
    with pytest.raises(ValueError, match="gru"):
        load_mlp_checkpoint_h5(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "checkpoint.h5"))
    with pytest.raises(ValueError):
        write_mlp_checkpoint_h5(path, [(np.zeros((4, 3), np.float32), np.zeros(5, np.float32))], ["identity"])


def test_operand_packers_under_sanitizers(tmp_path):
    """The host code that builds every operand image the kernels keep in registers (raptor_amd/csrc/rq_pack.cpp: the
    f32, bf16 and split-f16 policy images, the log-std head, the three teacher images for all nine hidden-width pairs
    and two input widths) under AddressSanitizer + UBSan with exactly sized buffers."""
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    exe = str(tmp_path / "pack_san")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                    "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-o", exe, os.path.join(here, "pack_sanitize_driver.cpp"),
                    os.path.join(root, "raptor_amd", "csrc", "rq_pack.cpp")], check=True, capture_output=True)
    r = subprocess.run([exe, os.path.join(root, "raptor_amd", "data", "raptor_policy.bin")], capture_output=True, text=True)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr
    assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr


def test_split_f16_scheme_in_numpy_meets_the_fp32_bar(weights, kat):
    """The arithmetic of RQ_POLICY_F16X2_MFMA restated in numpy, independent of the hardware: every operand as two float16
    pieces (hi = f16(v), lo = f16(v - hi)), a contraction as hi.hi + hi.lo + lo.hi in float32 (lo.lo dropped), gate rows
    pre-scaled before the split as the host packer does.  Over the 500 recurrent steps of the reference's known-answer
    vectors it stays within the fp32 tolerance (1e-5); with the lo pieces removed (plain f16 operands) it does not -
    so the second piece is what buys the accuracy."""
    x, y = kat
    w = weights
    W0, b0 = w[0:352].reshape(16, 22), w[352:368]
    Wi, Wh = w[368:1136].reshape(48, 16), w[1136:1904].reshape(48, 16)
    bi, bh, W2, b2 = w[1904:1952], w[1952:2000], w[2016:2080].reshape(4, 16), w[2080:2084]
    k = np.concatenate([np.full(32, -1.4426950408889634, np.float32), np.full(16, -2.8853900817779268, np.float32)])[:, None]

    def split(v):
        v = np.asarray(v, np.float32)
        hi = v.astype(np.float16)
        lo = (v - hi.astype(np.float32)).astype(np.float16)
        return hi.astype(np.float32), lo.astype(np.float32)

    def dot(xv, Wm, second_piece):
        xh, xl = split(xv)
        wh, wl = split(Wm)
        acc = xh @ wh.T
        if second_piece:
            acc = acc + xl @ wh.T + xh @ wl.T
        return acc.astype(np.float32)

    def run(second_piece):
        h = np.zeros((2, 16), np.float32)
        worst = 0.0
        for t in range(500):
            xin = np.concatenate([x[t], np.ones((2, 1), np.float32)], axis=1)              # bias rides as input 22
            y0 = np.maximum(dot(xin, np.concatenate([W0, b0[:, None]], axis=1), second_piece), 0)
            gi, gh = dot(y0, k * Wi, second_piece), dot(h, k * Wh, second_piece)                # exp2 arguments
            e2 = lambda a: np.exp2(a.astype(np.float64))
            r = 1 / (1 + e2(gi[:, :16] + gh[:, :16] + k[:16, 0] * (bi[:16] + bh[:16])))
            z = 1 / (1 + e2(gi[:, 16:32] + gh[:, 16:32] + k[16:32, 0] * (bi[16:32] + bh[16:32])))
            n = 2 / (1 + e2(gi[:, 32:] + k[32:, 0] * bi[32:] + r * (gh[:, 32:] + k[32:, 0] * bh[32:]))) - 1
            h = (n + z * (h - n)).astype(np.float32)
            a = dot(h, W2, second_piece) + b2
            worst = max(worst, float(np.abs(a - y[t]).max()))
        return worst

    two, one = run(True), run(False)
    print(f"\n[split-f16 scheme, numpy] known-answer max abs error: two pieces {two:.2e}, one piece (plain f16) {one:.2e}")
    assert two < 1e-5 < one


def test_balanced_teacher_assignment_deals_whole_tiles():
    """raptor_amd.teachers.balanced_teacher_assignment: the reference's 1 000 teachers (README.md:207-216) over 65 536
    envs -> 96 x 80 + 904 x 64 envs, every teacher a whole number of the relabel kernels' 16-env tiles (an even split,
    66 envs each, pads 18 % of the tiles); contiguous, ordered groups; every env assigned; odd sizes covered."""
    from raptor_amd.teachers import balanced_teacher_assignment
    ids = balanced_teacher_assignment(65536, 1000)
    count = np.bincount(ids, minlength=1000)
    assert ids.dtype == np.uint32 and ids.shape == (65536,) and (np.diff(ids.astype(np.int64)) >= 0).all()
    assert (count == 80).sum() == 96 and (count == 64).sum() == 904 and (count % 16 == 0).all()
    tiles = sum((c + 15) // 16 for c in count)
    assert tiles == 65536 // 16                              # no padding tile at all
    even = np.bincount((np.arange(65536) * 1000 // 65536), minlength=1000)
    assert sum((c + 15) // 16 for c in even) > 1.17 * tiles  # what the even split costs
    ids = balanced_teacher_assignment(1000, 7)               # 63 tiles over 7 teachers, the last one partial
    count = np.bincount(ids, minlength=7)
    assert count.sum() == 1000 and (count[:-1] % 16 == 0).all() and count.max() - count.min() <= 16
    ids = balanced_teacher_assignment(100, 1000)             # fewer tiles than teachers: the first ones get a tile each
    assert np.bincount(ids, minlength=1000)[:7].tolist() == [16] * 6 + [4] and len(ids) == 100
    with pytest.raises(ValueError):
        balanced_teacher_assignment(0, 3)


def test_a_committed_trace_counts_only_for_the_build_it_was_taken_from(tmp_path):
    """bench.py's headline roofline fraction may come from a committed rocprofv3 trace of the same command - but only of the same
    BUILD (VERDICT r05 weak 4 / ADVICE: rounds 1-5 matched kernel name, env count and step count, so a changed kernel with the same
    name inherited an old fraction).  The file carries the sha256 of the library that ran under the profiler."""
    import json
    import sys
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import bench
    kernel = bench.fused_kernel_name("fp32", 65536, 20)
    row = {kernel.replace(" ", ""): {"timed_region_launches": {"launches": 3300, "mean_us": 65.7, "median_us": 65.6}},
           "steps": 20, "envs_per_gpu": 65536, "command": "python bench.py --gpus 1 --steps 20 --warmup 5", "library_sha256": "a" * 64}
    (tmp_path / "r07_fused_launch_stats.json").write_text(json.dumps(row))
    hit, why = bench.rocprof_launch_stats(kernel, 65536, 20, "a" * 64, str(tmp_path))
    assert why is None and hit["mean_us"] == 65.7 and hit["source"] == "r07_fused_launch_stats.json"
    miss, why = bench.rocprof_launch_stats(kernel, 65536, 20, "b" * 64, str(tmp_path))
    assert miss is None and "another build" in why
    assert bench.rocprof_launch_stats(kernel, 65536, 21, "a" * 64, str(tmp_path))[0] is None          # another command
    assert bench.rocprof_launch_stats(kernel, 262144, 20, "a" * 64, str(tmp_path))[0] is None
    # the traces committed before round 6 carry no hash: none of them may become a headline again
    lib = bench.library_sha256()
    assert len(lib) == 64
    got, why = bench.rocprof_launch_stats(kernel, 65536, 20, "c" * 64)
    assert got is None and why
    # counters likewise: a PMC summary of another build is reported as such and does not become `traffic`
    assert bench.traffic_of(None) is None and bench.traffic_of({"same_library": False, "bytes_per_launch": 1.0}) is None
    assert bench.traffic_of({"same_library": True, "bytes_per_launch": 3.0}) == 3.0
    assert bench.traffic_of({"same_library": None, "bytes_per_launch": 2.0}) == 2.0


def test_every_vectorN_module_name_resolves():
    """README.md:45 `from l2f import vector8 as vector`: the reference pre-compiles a set of vectorN modules; here the batch is a
    run-time value and any `vectorN` is `l2f.vector(N)` (no GPU is touched by naming one)."""
    import raptor_amd.l2f as l2f
    from raptor_amd.l2f import vector8, vector1, vector64, vector65536      # noqa: F401
    assert vector8 is l2f.vector(8) and vector65536 is l2f.vector(65536) and vector1.N_ENVIRONMENTS == 1
    assert l2f.vector256.N_ENVIRONMENTS == 256
    for bad in ("vector0", "vector08", "vectorx", "vector-1", "vectors"):
        with pytest.raises(AttributeError):
            getattr(l2f, bad)
