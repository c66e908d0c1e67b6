#!/usr/bin/env python3
"""Median launch time of the three API-granular kernels at a given batch (bench.kernel_probe), for same-box A/B of two
library builds (RAPTOR_QUAD_LIB):   python tools/api_kernel_time.py [--envs 2097152]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raptor_amd.l2f as l2f                       # noqa: E402
import bench                                       # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=2097152)
args = ap.parse_args()
out = bench.kernel_probe(l2f.Device(), args.envs, 5 if args.envs > 500000 else 50)
print(json.dumps({k: (v["us_per_launch"], v["us_per_launch_min_max"], v["frac"]) for k, v in out.items()}))
