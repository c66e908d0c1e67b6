#!/usr/bin/env python3
"""Data collection as post-training needs it: roll the student out with auto-reset,
record every transition on the device, relabel the recorded observations with another policy, and hand the
result to a learner as torch tensors that alias the engine's buffers (no host copies anywhere).

    python examples/collect_and_relabel.py [--envs 65536] [--steps 100]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raptor_amd.l2f as l2f                       # noqa: E402
from raptor_amd.foundation_policy import Raptor, load_weights   # noqa: E402
from raptor_amd.teachers import TeacherBank, balanced_teacher_assignment, parameter_count    # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=100)
    args = ap.parse_args()

    device = l2f.Device()
    vector = l2f.vector(args.envs)
    rng, env = vector.VectorRng(), vector.VectorEnvironment()
    params, state = vector.VectorParameters(), vector.VectorState()
    vector.initialize_rng(device, rng, 0)
    vector.initialize_environment(device, env)
    vector.sample_initial_parameters(device, env, params, rng)
    vector.sample_initial_state(device, env, params, state, rng)

    student = Raptor(device)
    student.reset()
    # a stand-in "teacher": the same topology with perturbed weights
    w = load_weights()
    teacher = Raptor(device, weights=(w + np.random.default_rng(0).standard_normal(w.size).astype(np.float32) * 0.01))
    teacher.reset()

    traj = vector.Trajectory(env, args.steps)
    t0 = time.perf_counter()
    vector.rollout(device, env, params, state, student, rng, args.steps, "fused", autoreset=True, trajectory=traj)
    device.synchronize()
    t1 = time.perf_counter()
    traj.relabel(teacher, overwrite=True, fetch=False)      # stored actions <- the teacher's, on the device
    device.synchronize()
    t2 = time.perf_counter()

    # the distillation step proper (README.md:208-216): every quadrotor has ITS teacher, an MLP; here random
    # 22-64-64-4 teachers stand in for the trained ones - one launch.  balanced_teacher_assignment deals whole 16-env
    # tiles to the teachers (contiguous groups), so no matrix work is spent on padding whatever the teacher count
    n_teachers = 64
    bank = TeacherBank(device, (np.random.default_rng(1).standard_normal((n_teachers, parameter_count(22, 64, 64))) * 0.1)
                       .astype(np.float32), 22, 64, 64, "relu", "tanh")
    traj.relabel_teachers(bank, balanced_teacher_assignment(env.N_ENVIRONMENTS, n_teachers), overwrite=True, fetch=False)
    device.synchronize()
    t3 = time.perf_counter()

    batch = traj.tensors()                                   # torch views, field-major: obs [T, 22, ld], act [T, 4, ld]
    n = env.N_ENVIRONMENTS
    obs, target = batch["obs"][:, :, :n], batch["act"][:, :, :n]
    episodes = int((batch["done"][:, :n] != 0).sum().item())
    print(f"{n} envs x {args.steps} steps: collected in {(t1 - t0) * 1e3:.1f} ms "
          f"({n * args.steps / (t1 - t0):.3g} transitions/s), relabelled in {(t2 - t1) * 1e3:.1f} ms by a policy of the "
          f"student's topology and in {(t3 - t2) * 1e3:.1f} ms by a bank of {n_teachers} MLP teachers; "
          f"{episodes} episode ends; learner tensors obs {tuple(obs.shape)} target {tuple(target.shape)} "
          f"on {obs.device}, mean |target| {target.abs().mean().item():.3f}")


if __name__ == "__main__":
    main()
