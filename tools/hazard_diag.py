#!/usr/bin/env python3
"""Where do two identical fused bf16 rollouts differ?  (round 5: the two-waves-per-SIMD bf16 build, ActorBF16Lean, under
-amdgpu-sched-strategy=max-ilp; tools/hazard_variants.sh builds the libraries this is pointed at.)

Runs the same launch twice from identical inputs and reports the STRUCTURE of what differs: tile of the wave
(env % 64 // 16), lane position in the tile, which hidden features (Q layout: feature 4 q + r sits at lane group q,
register r of the tile), which state columns, which waves.  With --ref FILE the two runs are also compared with the result
a sound build left there (--save FILE), so that one can tell which of the two runs is the wrong one and what the wrong
values are (another tile's? the previous launch's?).

    RAPTOR_QUAD_LIB=scratch/variants/libraptor_quad_A.so python tools/hazard_diag.py --ref gpurun_out/ref.npz
"""
import argparse
import collections
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import raptor_amd.l2f as l2f                    # noqa: E402
from oracle import oracle as O                  # noqa: E402   (World mirrors its inputs on the oracle side; nothing is computed there)
from gpu_common import World               # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=131072)
ap.add_argument("--steps", type=int, default=1)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--autoreset", type=int, default=0)
ap.add_argument("--precision", default="bf16")
ap.add_argument("--sas", default="off", help="off | mean: the SampleAndSquash instantiation of the fused kernel")
ap.add_argument("--save")
ap.add_argument("--ref")
ap.add_argument("--quiet", action="store_true")
ap.add_argument("--tile", default="off", help="off | waves (every wave gets the envs of wave 0: a value leaking in from the same lane and register of "
                "ANOTHER wave is then the right one) | lanes (all 64 envs of a wave are its lane 0's: a value from another lane of the SAME wave is "
                "the right one) | all (one env everywhere: only a value from another TIME can be wrong)")
ap.add_argument("--hwid", action="store_true", help="library built with -DRQ_DEBUG_HWID: relate the waves that differ from --ref to which waves shared their SIMD")
ap.add_argument("--values", type=int, default=0, help="print right / wrong values (and where in the wave the wrong bits occur) for this many envs")
args = ap.parse_args()

device = l2f.Device(0)
tag = os.path.basename(os.environ.get("RAPTOR_QUAD_LIB", "product"))


def wave_records():
    import ctypes as C
    from raptor_amd import _lib
    n = C.c_uint32()
    _lib.call("rq_device_last_rollout_waves", device._h, None, 0, C.byref(n))
    rec = np.zeros((n.value, 4), np.uint64)
    _lib.call("rq_device_last_rollout_waves", device._h, rec.ctypes.data, n.value, C.byref(n))
    return rec


def sharing(rec, wrong_waves):
    """Per wave: did another wave of the same launch sit on the same SIMD while it ran?  Printed for all waves and for the wrong ones."""
    mask = np.uint64(0x0FFFFFFFFFFFFFFF)
    t_in, t_out = (rec[:, 0] & mask).astype(np.int64), (rec[:, 1] & mask).astype(np.int64)
    xcc = (rec[:, 1] >> np.uint64(60)).astype(np.int64) & 7
    hw = rec[:, 3].astype(np.int64)
    simd, cu, sh, se, slot = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, hw & 15
    place = ((((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd)
    cu_key = place >> 2
    n = len(rec)
    order = np.argsort(place, kind="stable")
    shared_simd = np.zeros(n, bool)
    overlap_ticks = np.zeros(n, np.int64)
    waves_on_cu = np.zeros(n, np.int64)
    for key, col in ((place, "simd"), (cu_key, "cu")):
        idx = np.argsort(key, kind="stable")
        bounds = np.nonzero(np.diff(key[idx]))[0] + 1
        for grp in np.split(idx, bounds):
            for a in grp:
                ov = np.minimum(t_out[grp], t_out[a]) - np.maximum(t_in[grp], t_in[a])
                ov[grp == a] = 0
                if col == "simd":
                    shared_simd[a] = (ov > 0).any()
                    overlap_ticks[a] = int(np.clip(ov, 0, None).max()) if len(ov) else 0
                else:
                    waves_on_cu[a] = int((ov > 0).sum()) + 1
    wrong = np.zeros(n, bool)
    wrong[list(wrong_waves)] = True
    dur = np.maximum(t_out - t_in, 1)
    def line(name, m):
        if not m.any():
            return f"    {name}: none"
        return (f"    {name}: {int(m.sum())} waves; shared their SIMD with another wave while running: {int(shared_simd[m].sum())} ({shared_simd[m].mean():.3f}); "
                f"median share of their run time overlapped {np.median(overlap_ticks[m] / dur[m]):.2f}; waves on their CU at once (median / max) "
                f"{np.median(waves_on_cu[m]):.0f} / {waves_on_cu[m].max()}; wave slots {sorted(set(slot[m].tolist()))[:10]}; run time {np.median(dur[m]) * 1e-2:.1f} us")
    print(line("all waves  ", np.ones(n, bool)))
    print(line("wrong waves", wrong))
    print(line("wrong waves that had their SIMD to themselves", wrong & ~shared_simd), flush=True)


def run(seed):
    w = World(device, O, args.n, seed=seed, episode_step_limit=4)
    w.policy.set_precision(args.precision)
    if args.tile != "off":
        S, P = w.state.numpy(), w.params.numpy()
        idx = np.arange(args.n)
        src = {"waves": idx % 64, "lanes": idx // 64 * 64, "all": idx * 0}[args.tile]
        w.state.set(S[src])
        w.params.set(P[src])
    if args.sas != "off":
        w.policy.set_sample_and_squash("mean")
    w.vector.rollout(device, w.env, w.params, w.state, w.policy, w.rng, args.steps, "fused", bool(args.autoreset))
    global last_records
    if args.hwid:
        device.synchronize()
        last_records = wave_records()
    return w.state.numpy(), w.policy.hidden_state(args.n)


def describe(name, sa, ha, sb, hb):
    ds = (sa != sb) & ~(np.isnan(sa) & np.isnan(sb))
    dh = (ha != hb) & ~(np.isnan(ha) & np.isnan(hb))
    rows = np.nonzero(ds.any(axis=1) | dh.any(axis=1))[0]
    print(f"[{tag}] {name}: {len(rows)} differing envs (state {int(ds.any(axis=1).sum())}, hidden {int(dh.any(axis=1).sum())})", flush=True)
    if len(rows) == 0 or args.quiet:
        return len(rows)
    waves = rows // 64
    tiles = collections.Counter(((rows % 64) // 16).tolist())
    per_wave_tile = collections.Counter(zip(waves.tolist(), ((rows % 64) // 16).tolist()))
    full = sum(1 for v in per_wave_tile.values() if v == 16)
    print(f"    waves touched {len(set(waves.tolist()))}, (wave, tile) pairs {len(per_wave_tile)} of which complete 16-env tiles {full}; by tile {dict(sorted(tiles.items()))}")
    print(f"    wave index mod 8 {dict(sorted(collections.Counter((waves % 8).tolist()).items()))}   first waves {sorted(set(waves.tolist()))[:12]}")
    print(f"    state columns differing (count): { {int(c): int(ds[:, c].sum()) for c in np.nonzero(ds.any(axis=0))[0]} }")
    hm = collections.Counter(tuple(np.nonzero(dh[r])[0].tolist()) for r in rows)
    print(f"    hidden-feature patterns (features -> envs): {[(k, v) for k, v in hm.most_common(6)]}")
    r0 = rows[0]
    print(f"    e.g. env {r0}: hidden a {ha[r0][:8]} b {hb[r0][:8]}  last action a {sa[r0][17:21]} b {sb[r0][17:21]}")
    if args.values and name.strip().startswith("run"):      # against the reference build: sb / hb are the right values
        for r0 in rows[:args.values]:
            w0 = (r0 // 64) * 64
            wave_s, wave_h = sb[w0:w0 + 64].view(np.uint32), hb[w0:w0 + 64].view(np.uint32)
            for c in np.nonzero(ds[r0])[0]:
                bad, good = sa[r0, c], sb[r0, c]
                bits = sa[r0:r0 + 1, c].view(np.uint32)[0]
                where = [f"state[lane {l}][col {k}]" for l, k in zip(*np.nonzero(wave_s == bits))][:4]
                where += [f"hidden[lane {l}][{k}]" for l, k in zip(*np.nonzero(wave_h == bits))][:4]
                lo = np.array([bits << 16], np.uint32).view(np.float32)[0]
                hi = np.array([bits & 0xFFFF0000], np.uint32).view(np.float32)[0]
                print(f"      env {r0} (wave {r0 // 64}, lane {r0 % 64}) col {c}: right {good!r} wrong {bad!r} = 0x{bits:08x} "
                      f"(as bf16 pair: lo {lo!r} hi {hi!r}); same bits in the RIGHT data of this wave: {where or 'nowhere'}")
    return len(rows)


total = 0
saved = {}
last_records = None
if args.hwid:
    device.set_rollout_timing(True)
ref = np.load(args.ref) if args.ref else None
for rep in range(args.reps):
    sa, ha = run(9 + rep)
    rec_a = last_records
    sb, hb = run(9 + rep)
    rec_b = last_records
    total += describe(f"seed {9 + rep} run a vs run b ({args.n} envs, {args.steps} step(s), auto-reset {args.autoreset}, sas {args.sas})", sa, ha, sb, hb)
    if ref is not None:
        describe("    run a vs the reference build", sa, ha, ref[f"s{rep}"], ref[f"h{rep}"])
        describe("    run b vs the reference build", sb, hb, ref[f"s{rep}"], ref[f"h{rep}"])
        if args.hwid:
            for name, s_, h_, rec in (("a", sa, ha, rec_a), ("b", sb, hb, rec_b)):
                rs, rh = ref[f"s{rep}"], ref[f"h{rep}"]
                bad = ((s_ != rs) & ~(np.isnan(s_) & np.isnan(rs))).any(axis=1) | ((h_ != rh) & ~(np.isnan(h_) & np.isnan(rh))).any(axis=1)
                print(f"  run {name}: where its waves ran")
                sharing(rec, set((np.nonzero(bad)[0] // 64).tolist()))
    saved[f"s{rep}"], saved[f"h{rep}"] = sa, ha
if args.save:
    np.savez(args.save, **saved)
print(f"[{tag}] TOTAL differing envs over {args.reps} seeds: {total}")
