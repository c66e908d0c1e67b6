#!/usr/bin/env python3
"""k_actor_step launch-shape sweep: 64-env groups per wave (RQ_ACTOR_GROUPS_PER_WAVE, read once per process by
launch_actor_step) against the batch size.  One process per setting:

    python tools/actor_gpw_sweep.py            # the sweep (spawns itself)
    RQ_ACTOR_GROUPS_PER_WAVE=16 python tools/actor_gpw_sweep.py --one 2097152
"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def one(n):
    import numpy as np
    import raptor_amd.l2f as l2f
    from bench import Shard, BYTES_ACTOR
    device = l2f.Device()
    sh = Shard(device, n, 0)
    sh.vector.observe(device, sh.env, sh.params, sh.state, None, sh.rng)
    reps = 50 if n <= 262144 else 10
    for _ in range(50):
        sh.policy.evaluate_step_device(sh.env)
    device.synchronize()
    per = []
    for _ in range(15):
        device.timer_start()
        for _ in range(reps):
            sh.policy.evaluate_step_device(sh.env)
        per.append(device.timer_stop() * 1e3 / reps)
    us = float(np.median(per))
    print(f"groups_per_wave {os.environ.get('RQ_ACTOR_GROUPS_PER_WAVE', 'default'):>7s}  envs {n:8d}  {us:8.2f} us  "
          f"{BYTES_ACTOR * n / us / 1e6:6.2f} TB/s algorithmic", flush=True)


if __name__ == "__main__":
    if "--one" in sys.argv:
        one(int(sys.argv[sys.argv.index("--one") + 1]))
    else:
        for n in (65536, 262144, 1048576, 2097152):
            for g in ("", "1", "2", "4", "8", "16", "32", "64"):
                if g and int(g) * 64 * 256 > n * 4 and g != "1":      # fewer than 64 waves left: pointless
                    continue
                env = dict(os.environ)
                env.pop("RQ_ACTOR_GROUPS_PER_WAVE", None)
                if g:
                    env["RQ_ACTOR_GROUPS_PER_WAVE"] = g
                subprocess.run([sys.executable, os.path.abspath(__file__), "--one", str(n)], env=env)
