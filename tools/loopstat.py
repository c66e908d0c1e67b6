#!/usr/bin/env python3
"""Instruction mix of the hot loop of one kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only).

    python tools/loopstat.py scratch/rq_kernels.s 'k_rollout_fusedILb0ELb1ELb0ENS_9ActorF32TILb0E'

The hot loop is taken to be the longest span between a label and a backward branch to it.
"""
import collections
import re
import sys

path, needle = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if needle in l and l.rstrip().endswith(":") is False and re.match(r"^_Z\w+:", l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = lines[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\w+):", l)
    if m:
        labels[m.group(1)] = i
best = (0, 0, 0)
for i, l in enumerate(body):
    m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\w+)", l) or re.match(r"\s+s_branch\s+(\.LBB\w+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i and i - labels[m.group(1)] > best[0]:
        best = (i - labels[m.group(1)], labels[m.group(1)], i)
_, lo, hi = best
cls = collections.Counter()
ops = collections.Counter()
for l in body[lo:hi + 1]:
    m = re.match(r"\s+([a-z_0-9]+)\s", l + " ")
    if not m or l.strip().startswith((";", ".")):
        continue
    op = m.group(1)
    ops[op] += 1
    if op.startswith("v_mfma"):
        cls["mfma"] += 1
    elif op.startswith(("v_exp", "v_rcp", "v_rsq", "v_sqrt", "v_log", "v_sin", "v_cos")):
        cls["trans"] += 1
    elif op.startswith("v_accvgpr"):
        cls["accvgpr"] += 1
    elif op.startswith(("v_permlane", "v_readlane", "v_writelane", "v_readfirstlane", "ds_bpermute", "ds_swizzle")):
        cls["lane"] += 1
    elif op.startswith("v_mov") or op.startswith("v_pk_mov"):
        cls["v_mov"] += 1
    elif op.startswith("v_"):
        cls["valu"] += 1
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        cls["vmem"] += 1
    elif op.startswith("ds_"):
        cls["lds"] += 1
    elif op.startswith("s_nop") or op.startswith("s_waitcnt"):
        cls["wait/nop"] += 1
    elif op.startswith("s_"):
        cls["salu"] += 1
    else:
        cls["other"] += 1
print(f"loop lines {lo}..{hi} of kernel ({hi - lo} lines)")
print(dict(cls), "total", sum(cls.values()))
print("top ops:", ops.most_common(40))
