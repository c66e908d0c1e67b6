"""tools/tile_all.py [steps] - one env in all 131 072 slots of a bf16 fused rollout (RAPTOR_QUAD_LIB = an experiment build): how many distinct
outcomes, in which lanes, which columns.  Round 5, profiles/r05_wrong_value_traced.txt."""
import os, sys, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import raptor_amd.l2f as l2f
from oracle import oracle as O
from gpu_common import World
device = l2f.Device(0)
n = 131072
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
def run(seed):
    w = World(device, O, n, seed=seed, episode_step_limit=4)
    w.policy.set_precision("bf16")
    S, P = w.state.numpy(), w.params.numpy()
    src = np.zeros(n, np.int64) + 5
    w.state.set(S[src]); w.params.set(P[src])
    w.vector.rollout(device, w.env, w.params, w.state, w.policy, w.rng, steps, "fused", False)
    return np.concatenate([w.state.numpy(), w.policy.hidden_state(n)], axis=1)
np.set_printoptions(linewidth=250, precision=7)
for seed in (9, 10):
    for rep in range(3):
        X = run(seed)
        rows, counts = np.unique(X.view(np.uint32), axis=0, return_counts=True)
        right = rows[np.argmax(counts)]
        print(f"seed {seed} rep {rep}: {len(rows)} distinct outcomes; the common one {counts.max()} envs")
        bad = np.nonzero((X.view(np.uint32) != right).any(axis=1))[0]
        print("   lanes of wrong envs:", dict(sorted(collections.Counter((bad % 64 // 16).tolist()).items())))
        pat = collections.Counter()
        for r, c in zip(rows, counts):
            if (r != right).any():
                cols = tuple(np.nonzero(r != right)[0].tolist())
                pat[cols] += c
        for cols, c in pat.most_common(8):
            print(f"   {c:6d} envs differ in columns {cols}")
        k = 0
        for r, c in sorted(zip(rows.tolist(), counts.tolist()), key=lambda t: -t[1]):
            r = np.array(r, np.uint32)
            if (r != right).any() and k < 6:
                k += 1
                cols = np.nonzero(r != right)[0]
                print(f"      x{c}: cols {cols.tolist()[:12]} right {right.view(np.float32)[cols][:6]} wrong {r.view(np.float32)[cols][:6]}")
