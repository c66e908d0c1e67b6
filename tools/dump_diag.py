"""tools/dump_diag.py cal | dump NAME,NAME,...   - read the register dumps of a library built from a tools/asm_instrument.py listing
(RAPTOR_QUAD_LIB): `cal` with the calibration library first (which state column carries which dump slot), then `dump` with the names of
the dumped registers in slot order.  One env in all 131 072 slots, so every wave must dump the same values: waves that differ from the
majority are listed with the register, the lanes and the values.  Round 5, profiles/r05_wrong_value_traced.txt."""
import os, sys, json, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import raptor_amd.l2f as l2f
from oracle import oracle as O
from gpu_common import World
device = l2f.Device(0)
n = 131072
mode = sys.argv[1]
names = sys.argv[2].split(",") if len(sys.argv) > 2 else []
def run(seed):
    w = World(device, O, n, seed=seed, episode_step_limit=4)
    w.policy.set_precision("bf16")
    S, P = w.state.numpy(), w.params.numpy()
    src = np.zeros(n, np.int64) + 5
    w.state.set(S[src]); w.params.set(P[src])
    w.vector.rollout(device, w.env, w.params, w.state, w.policy, w.rng, 1, "fused", False)
    return w.state.numpy(), w.policy.hidden_state(n)
if mode == "cal":
    S, H = run(9)
    m = {}
    for c in range(S.shape[1]):
        v = S[0, c]
        if v == round(v) and 1 <= v <= 20 and (S[:, c] == v).all():
            m[int(v) - 1] = c
    print("slot -> state column:", m)
    json.dump(m, open("/tmp/cal.json", "w"))
    sys.exit(0)
cal = {int(k): v for k, v in json.load(open("/tmp/cal.json")).items()}
def bf(u):
    lo = np.array([(int(u) & 0xFFFF) << 16], np.uint32).view(np.float32)[0]
    hi = np.array([int(u) & 0xFFFF0000], np.uint32).view(np.float32)[0]
    return f"0x{int(u):08x}=({lo:.5g},{hi:.5g}|f32 {np.array([u],np.uint32).view(np.float32)[0]:.6g})"
tot_wrong = 0
for seed in (9, 10):
    for rep in range(4):
        S, H = run(seed)
        Hu = H.view(np.uint32).reshape(n // 64, 64 * 16)
        rows, inv, counts = np.unique(Hu, axis=0, return_inverse=True, return_counts=True)
        good = np.argmax(counts)
        wrong_waves = np.nonzero(inv.reshape(-1) != good)[0]
        D = np.stack([S[:, cal[k]] for k in range(len(names))], axis=1).view(np.uint32).reshape(n // 64, 64, len(names))
        drows, dinv, dcounts = np.unique(D.reshape(n // 64, -1), axis=0, return_inverse=True, return_counts=True)
        ref = drows[np.argmax(dcounts)].reshape(64, len(names))
        dump_differs = np.nonzero((D != ref[None]).any(axis=(1, 2)))[0]
        if rep == 0:
            print("   right values at lane 48:", {names[k]: bf(ref[48, k]) for k in range(len(names))})
        tot_wrong += len(wrong_waves)
        print(f"seed {seed} rep {rep}: waves with a wrong final hidden state {len(wrong_waves)}; waves whose dumped registers differ from the majority {len(dump_differs)}; both {len(set(wrong_waves) & set(dump_differs))}")
        pat = collections.Counter()
        for w in dump_differs:
            d = (D[w] != ref)
            for k in np.nonzero(d.any(axis=0))[0]:
                lanes = np.nonzero(d[:, k])[0]
                pat[(names[k], (int(lanes.min()), int(lanes.max()), len(lanes)), w in set(wrong_waves))] += 1
        for (nm, lanes, isw), c in pat.most_common(12):
            print(f"      {c:4d} waves: {nm} differs in lanes {lanes[0]}..{lanes[1]} ({lanes[2]} lanes); final result wrong: {isw}")
        for w in dump_differs[:2]:
            d = (D[w] != ref)
            for k in np.nonzero(d.any(axis=0))[0][:4]:
                l = np.nonzero(d[:, k])[0][0]
                same = [(names[k2], int(l2)) for k2 in range(len(names)) for l2 in np.nonzero(ref[:, k2] == D[w, l, k])[0][:3]][:6]
                print(f"         wave {w} {names[k]} lane {l}: right {bf(ref[l, k])} wrong {bf(D[w, l, k])}; the wrong bits occur (right data) at {same or 'nowhere in the dump'}")
print("total wrong waves", tot_wrong)
