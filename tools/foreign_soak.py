#!/usr/bin/env python3
"""The gfx950 packed-fp32 op_sel fault between this library and a FOREIGN kernel on a shared GPU - the deployment
examples/collect_and_relabel.py and the zero-copy torch views invite: a PyTorch learner on the same device as the rollout engine.

The fault needs a victim (a wave executing v_pk_{add,mul,fma}_f32 ... op_sel:[0,1]) and an aggressor (ANOTHER wave of the same SIMD
executing a 16- or 8-bit MFMA).  tools/cross_stream_soak.py has this library on both sides.  Here one side is PyTorch:

  torch_aggressor   torch.matmul of bf16 matrices (hipBLASLt / rocBLAS 16-bit MFMA kernels) without pause on torch's stream; this
                    library repeats an fp32 workload of API-granular kernels (chained rollout) on its own stream and must get, bit
                    for bit, what it gets on an idle GPU.  (The product build holds no instruction of the form - tools/codeobj_check.py -
                    so the expected count is 0; a build made with RQ_NO_OPSEL_REWRITE=1 shows whether a foreign aggressor bites.)
  torch_victim      this library's bf16 fused rollouts (16-bit MFMAs, one wave per SIMD, 300-390 registers: a small foreign wave fits
                    beside it) without pause; PyTorch repeats fp32 workloads - an Adam-style elementwise update, layer_norm + gelu +
                    softmax, an fp32 matmul - and each must give, bit for bit, what it gives on an idle GPU.  Whether PyTorch's kernels
                    hold the form is PyTorch's matter; this says whether THIS library, as an aggressor, changes a learner's numbers.

    python tools/foreign_soak.py [--reps 20] [--direction both] [--json gpurun_out/r06_foreign_soak.json]
    RAPTOR_QUAD_LIB=scratch/variants/libraptor_quad_NP.so python tools/foreign_soak.py --direction torch_aggressor
Exit status 1 if any repetition differs."""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                       # noqa: E402
import raptor_amd.l2f as l2f                       # noqa: E402
from bench import Shard                            # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=65536)
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--direction", default="both", choices=["both", "torch_aggressor", "torch_victim"])
ap.add_argument("--json", default=None)
args = ap.parse_args()
tag = os.path.basename(os.environ.get("RAPTOR_QUAD_LIB", "product"))
cuda = torch.device("cuda:0")


class Background:
    """fn() over and over on a thread until stopped; start() returns once it has run three times."""

    def __init__(self, fn, name):
        self.fn, self.name, self.count, self.failed = fn, name, 0, None
        self.stop_flag = threading.Event()
        self.thread = threading.Thread(target=self._loop, daemon=True)

    def _loop(self):
        try:
            while not self.stop_flag.is_set():
                self.fn()
                self.count += 1
        except Exception as e:                     # noqa: BLE001  (reported by the main thread)
            self.failed = repr(e)

    def start(self):
        self.thread.start()
        deadline = time.monotonic() + 180.0
        while self.count < 3 and self.failed is None and time.monotonic() < deadline:
            time.sleep(0.001)
        if self.failed or self.count < 3:
            self.stop()
            sys.exit(f"{self.name} did not start: {self.failed or 'timeout'}")

    def stop(self):
        self.stop_flag.set()
        self.thread.join(120.0)
        if self.failed:
            sys.exit(f"{self.name} failed: {self.failed}")


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32) if a.dtype == np.float32 else np.ascontiguousarray(a)


results = []

# ------------------------------------------------------------------ torch is the aggressor, this library the victim
if args.direction in ("both", "torch_aggressor"):
    dev = l2f.Device(0)

    def rq_workload():
        sh = Shard(dev, args.envs, 0, seed=7, precision="fp32")
        sh.rollout(args.steps, "chained")
        return np.concatenate([sh.state.numpy(), sh.policy.hidden_state(args.envs)], axis=1)

    quiet = rq_workload()
    assert (bits(quiet) == bits(rq_workload())).all(), "the fp32 workload is not deterministic on an idle GPU"
    a = torch.randn(4096, 4096, device=cuda, dtype=torch.bfloat16)
    b = torch.randn(4096, 4096, device=cuda, dtype=torch.bfloat16)
    sa, sb = torch.randn(4096, 64, 64, device=cuda, dtype=torch.bfloat16), torch.randn(4096, 64, 64, device=cuda, dtype=torch.bfloat16)
    q, k, v = (torch.randn(32, 8, 512, 64, device=cuda, dtype=torch.bfloat16) for _ in range(3))
    ha, hb = torch.randn(1024, 1024, device=cuda, dtype=torch.float16), torch.randn(1024, 1024, device=cuda, dtype=torch.float16)

    def many(fn, times=20):
        def loop():
            for _ in range(times):
                fn()
            torch.cuda.synchronize()
        return loop

    def churn():
        x = torch.empty(1 << 22, device=cuda)
        torch.cuda.synchronize()
        del x
        torch.cuda.empty_cache()

    # several shapes of foreign 16-bit MFMA kernel: big GEMM tiles fill a CU's registers and LDS (little room for another wave on their
    # SIMDs), small batched GEMMs and attention kernels leave more
    aggressors = {"torch.matmul bf16 4096^3": many(lambda: torch.matmul(a, b)),
                  "torch.bmm bf16 4096 x 64^3": many(lambda: torch.bmm(sa, sb)),
                  "scaled_dot_product_attention bf16 [32, 8, 512, 64]": many(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v)),
                  "torch.matmul f16 1024^3": many(lambda: torch.matmul(ha, hb), 50),
                  # no MFMA at all: a thread that allocates, frees and synchronises the device the whole time - what invalidates another
                  # thread's open hipGraph capture (the chained rollout captures one per new env): the rollout must fall back to plain
                  # launches, not fail (rq_capi_rollout.cpp rollout_impl)
                  "hipMalloc / hipFree / hipDeviceSynchronize churn": many(churn, 5)}
    for what, fn in aggressors.items():
        bg = Background(fn, what)
        bg.start()
        bad_reps = bad_envs = 0
        for _ in range(args.reps):
            d = (bits(rq_workload()) != bits(quiet)).any(axis=1)
            bad_reps += int(d.any())
            bad_envs += int(d.sum())
        bg.stop()
        results.append({"direction": "torch_aggressor", "library": tag, "victim": f"{args.steps} fp32 chained steps on {args.envs} envs (this library)",
                        "aggressor": f"{what}, {bg.count} batches on torch's stream", "repetitions": args.reps,
                        "repetitions_that_differ": bad_reps, "envs_that_differ": bad_envs})
        print(json.dumps(results[-1]), flush=True)

# ------------------------------------------------------------------ this library is the aggressor, torch the victim
if args.direction in ("both", "torch_victim"):
    dev_a = l2f.Device(0)
    g = torch.Generator(device="cpu").manual_seed(1)
    n_el = 1 << 22
    p0 = torch.randn(n_el, generator=g).to(cuda)
    grad = torch.randn(n_el, generator=g).to(cuda)
    x0 = torch.randn(4096, 1024, generator=g).to(cuda)
    wmat = torch.randn(1024, 1024, generator=g).to(cuda)
    lnw, lnb = torch.randn(1024, generator=g).to(cuda), torch.randn(1024, generator=g).to(cuda)

    def adam_like():
        p, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
        for t in range(1, 41):
            gk = grad * (1.0 + 0.01 * t) + 1e-4 * p
            m.mul_(0.9).add_(gk, alpha=0.1)
            v.mul_(0.999).addcmul_(gk, gk, value=0.001)
            p.addcdiv_(m / (1 - 0.9 ** t), (v / (1 - 0.999 ** t)).sqrt_().add_(1e-8), value=-1e-3)
        return torch.stack([p, m, v])

    def norm_gelu_softmax():
        y = x0
        for _ in range(20):
            y = torch.nn.functional.layer_norm(y, (1024,), lnw, lnb)
            y = torch.nn.functional.gelu(y)
            y = torch.softmax(y, dim=-1) * 1024.0 - 1.0
        return y

    def fp32_matmul():
        y = x0
        for _ in range(10):
            y = torch.tanh(torch.matmul(y, wmat) * 0.03)
        return y

    workloads = {"adam_like_elementwise": adam_like, "layer_norm_gelu_softmax": norm_gelu_softmax, "fp32_matmul_tanh": fp32_matmul}
    quiet = {}
    for name, fn in list(workloads.items()):
        r0 = fn().cpu().numpy()
        r1 = fn().cpu().numpy()
        if not (bits(r0) == bits(r1)).all():
            results.append({"direction": "torch_victim", "victim": name, "skipped": "not deterministic on an idle GPU"})
            print(json.dumps(results[-1]), flush=True)
            del workloads[name]
            continue
        quiet[name] = r0
    sh = Shard(dev_a, args.envs, 0, seed=3, precision="bf16")

    def bf16_rollouts():
        sh.rollout(2000, "fused")                  # a few milliseconds per launch, auto-reset: the stream is busy almost all the time
        dev_a.synchronize()

    bg = Background(bf16_rollouts, "this library's bf16 rollouts")
    bg.start()
    counts = {name: [0, 0] for name in workloads}
    for _ in range(args.reps):
        for name, fn in workloads.items():
            d = bits(fn().cpu().numpy()) != bits(quiet[name])
            counts[name][0] += int(d.any())
            counts[name][1] += int(d.sum())
    bg.stop()
    for name in workloads:
        results.append({"direction": "torch_victim", "library": tag, "victim": f"torch fp32 {name}",
                        "aggressor": f"{bg.count} bf16 fused rollouts of 2000 steps on {args.envs} envs (this library, its own stream)",
                        "repetitions": args.reps, "repetitions_that_differ": counts[name][0], "elements_that_differ": counts[name][1]})
        print(json.dumps(results[-1]), flush=True)

if args.json:
    os.makedirs(os.path.dirname(os.path.abspath(args.json)), exist_ok=True)
    with open(args.json, "w") as fh:
        json.dump({"torch": torch.__version__, "hip": torch.version.hip, "device": torch.cuda.get_device_name(0), "results": results}, fh, indent=1)
        fh.write("\n")
sys.exit(1 if any(r.get("repetitions_that_differ") for r in results) else 0)
