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
from test_gpu_parity import World               # noqa: E402

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
ap.add_argument("--values", type=int, default=0, help="print right / wrong values (and where in the wave the wrong bits occur) for this many envs")
args = ap.parse_args()

device = l2f.Device(0)
tag = os.path.basename(os.environ.get("RAPTOR_QUAD_LIB", "product"))


def run(seed):
    w = World(device, O, args.n, seed=seed, episode_step_limit=4)
    w.policy.set_precision(args.precision)
    if args.sas != "off":
        w.policy.set_sample_and_squash("mean")
    w.vector.rollout(device, w.env, w.params, w.state, w.policy, w.rng, args.steps, "fused", bool(args.autoreset))
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
ref = np.load(args.ref) if args.ref else None
for rep in range(args.reps):
    sa, ha = run(9 + rep)
    sb, hb = run(9 + rep)
    total += describe(f"seed {9 + rep} run a vs run b ({args.n} envs, {args.steps} step(s), auto-reset {args.autoreset}, sas {args.sas})", sa, ha, sb, hb)
    if ref is not None:
        describe("    run a vs the reference build", sa, ha, ref[f"s{rep}"], ref[f"h{rep}"])
        describe("    run b vs the reference build", sb, hb, ref[f"s{rep}"], ref[f"h{rep}"])
    saved[f"s{rep}"], saved[f"h{rep}"] = sa, ha
if args.save:
    np.savez(args.save, **saved)
print(f"[{tag}] TOTAL differing envs over {args.reps} seeds: {total}")
