#!/usr/bin/env python3
"""Extracts every number the reference's own training run left in its tree into tests/golden/reference_log.json:

    python tests/golden/make_reference_log.py [/root/reference/data/raptor-policy-checkpoint.tar.gz]

Source: `2025-04-19_16-16-17/logs.tfevents` (a TensorBoard event file: 11 scalar tags) and the
`/actor@meta` attribute of `2025-04-19_16-16-17/checkpoint.h5` inside the tarball.  These are DATA the real l2f +
rl-tools produced (evaluation statistics of the policy this repository ships, per training epoch) - the only
reference-produced numbers about the environment half of the path that exist here; the fixture holds the values, no
reference source text.  Runs in the build container only (/root/reference does not travel); the JSON is committed.

What is kept: for the ten evaluation tags every epoch's value (float32 as logged, 7 significant digits) with the
step axis; for `loss` (146 103 points) a summary plus every 100th point; per tag n / first / last / min / max /
mean of the last 20; pooled late-epoch statistics with the standard error over epochs (tools/env_constraint_study.py
and tests/test_closed_loop.py read those)."""
import io
import json
import os
import struct
import sys
import tarfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
RUN = "2025-04-19_16-16-17"


def _varint(b, i):
    x = s = 0
    while True:
        c = b[i]
        i += 1
        x |= (c & 0x7F) << s
        s += 7
        if not c & 0x80:
            return x, i


def _fields(b):
    """protobuf wire format: (field number, wire type, value) of one message"""
    i = 0
    while i < len(b):
        key, i = _varint(b, i)
        f, w = key >> 3, key & 7
        if w == 0:
            v, i = _varint(b, i)
        elif w == 1:
            v, i = b[i:i + 8], i + 8
        elif w == 2:
            n, i = _varint(b, i)
            v, i = b[i:i + n], i + n
        elif w == 5:
            v, i = b[i:i + 4], i + 4
        else:
            raise ValueError(f"wire type {w}")
        yield f, w, v


def read_scalars(data):
    """TFRecord framing (u64 length, u32 crc, payload, u32 crc) of tensorflow.Event messages ->
    {tag: [(step, wall_time, simple_value)]}"""
    out, i = {}, 0
    while i + 12 <= len(data):
        (n,) = struct.unpack("<Q", data[i:i + 8])
        i += 12
        rec, i = data[i:i + n], i + n + 4
        wall = step = summary = None
        for f, w, v in _fields(rec):
            if f == 1 and w == 1:
                wall = struct.unpack("<d", v)[0]
            elif f == 2 and w == 0:
                step = v
            elif f == 5 and w == 2:
                summary = v
        if summary is None:
            continue
        for f, w, v in _fields(summary):
            if f != 1:
                continue
            tag = val = None
            for g, w2, u in _fields(v):
                if g == 1:
                    tag = u.decode()
                elif g == 2 and w2 == 5:
                    val = struct.unpack("<f", u)[0]
            if tag is not None and val is not None:
                out.setdefault(tag, []).append((step, wall, val))
    return out


def pooled(share, length, lo):
    """late-epoch pool [lo:]: share terminated, episode length, and the mean length of the TERMINATED episodes they
    imply ((L - (1 - s) 500) / s), each with the standard error over epochs (bootstrap for the ratio)"""
    s, l = share[lo:], length[lo:]
    rng = np.random.default_rng(0)
    boots = []
    for _ in range(1000):
        k = rng.integers(0, len(s), len(s))
        a, b = s[k].mean(), l[k].mean()
        boots.append((b - (1 - a) * 500.0) / a)
    return {"epochs": [int(lo), int(len(share))], "share_terminated": round(float(s.mean()), 5),
            "share_terminated_se": round(float(s.std() / np.sqrt(len(s))), 5),
            "episode_length": round(float(l.mean()), 3), "episode_length_se": round(float(l.std() / np.sqrt(len(l))), 3),
            "terminated_episode_length": round(float((l.mean() - (1 - s.mean()) * 500.0) / s.mean()), 2),
            "terminated_episode_length_se": round(float(np.std(boots)), 2)}


def main(tarball):
    from raptor_amd import hdf5_min
    with tarfile.open(tarball) as tf:
        events = tf.extractfile(f"{RUN}/logs.tfevents").read()
        h5 = tf.extractfile(f"{RUN}/checkpoint.h5").read()
    tags = read_scalars(events)
    doc = {"source": f"{os.path.basename(tarball)}:{RUN}/logs.tfevents + checkpoint.h5:/actor@meta",
           "made_by": "tests/golden/make_reference_log.py", "episode_step_limit": 500, "tags": {}, "series": {}}
    tmp = os.path.join(HERE, "_meta_tmp.h5")
    with open(tmp, "wb") as fh:
        fh.write(h5)
    try:
        meta = hdf5_min.File(tmp).root["actor"].attrs["meta"]
    finally:
        os.remove(tmp)
    doc["actor_meta"] = json.loads(meta)
    for tag, rows in sorted(tags.items()):
        v = np.array([r[2] for r in rows], np.float64)
        doc["tags"][tag] = {"n": len(v), "first": float(np.float32(v[0])), "last": float(np.float32(v[-1])),
                            "min": float(v.min()), "max": float(v.max()), "mean_last_20": round(float(v[-20:].mean()), 6)}
        steps = [r[0] if r[0] is not None else 0 for r in rows]
        if tag == "loss":
            doc["series"][tag] = {"every": 100, "step": steps[::100], "value": [float(f"{x:.7g}") for x in v[::100]]}
        else:
            doc["series"][tag] = {"step": steps, "value": [float(f"{x:.7g}") for x in v]}
    walls = [r[1] for r in tags["crazyflie/share_terminated"]]
    doc["seconds_per_epoch"] = round((walls[-1] - walls[0]) / (len(walls) - 1), 3)
    doc["gradient_steps_per_epoch"] = round(len(tags["loss"]) / len(tags["evaluation/share_terminated"]), 3)
    # how many episodes one epoch's evaluation holds: the logged shares are multiples of 1 / episodes
    for pre in ("crazyflie", "evaluation"):
        s = np.unique(np.round(np.array(doc["series"][pre + "/share_terminated"]["value"]), 6))
        doc.setdefault("episodes_per_evaluation", {})[pre] = int(round(1.0 / np.diff(s).min()))
    # pooled late-epoch statistics (the policy shipped is the last epoch's; the evaluation of ONE epoch of the
    # nominal-Crazyflie tag is 100 episodes: its last value, 0.05, carries a sampling error of 0.02)
    doc["pooled"] = {}
    for pre in ("crazyflie", "evaluation"):
        share = np.array(doc["series"][pre + "/share_terminated"]["value"])
        length = np.array(doc["series"][pre + "/episode_length/mean"]["value"])
        doc["pooled"][pre] = {f"last_{len(share) - lo}": pooled(share, length, lo) for lo in (len(share) - 20, len(share) - 100, len(share) - 200)}
    # return = a x length + b x share over the late epochs: a = reward per step of the surviving episodes' regime, b bounds the
    # termination penalty (b = (r_terminated - r) L_terminated - penalty)
    doc["return_regression_last_500"] = {}
    for pre in ("crazyflie", "evaluation"):
        share = np.array(doc["series"][pre + "/share_terminated"]["value"])[-500:]
        length = np.array(doc["series"][pre + "/episode_length/mean"]["value"])[-500:]
        ret = np.array(doc["series"][pre + "/return/mean"]["value"])[-500:]
        A = np.stack([length, share], 1)
        coef = np.linalg.lstsq(A, ret, rcond=None)[0]
        doc["return_regression_last_500"][pre] = {"reward_per_step": round(float(coef[0]), 4), "per_unit_share": round(float(coef[1]), 2),
                                                  "rms_residual": round(float(np.sqrt(np.mean((A @ coef - ret) ** 2))), 3)}
    path = os.path.join(HERE, "reference_log.json")
    with open(path, "w") as fh:
        json.dump(doc, fh, separators=(",", ":"))
        fh.write("\n")
    print("wrote", path, os.path.getsize(path), "bytes;", len(tags), "tags")
    for k in ("episodes_per_evaluation", "pooled", "return_regression_last_500", "actor_meta"):
        print(k, json.dumps(doc[k], indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/data/raptor-policy-checkpoint.tar.gz")
