#!/usr/bin/env python3
"""Fixture extractor: RAPTOR policy checkpoint -> flat little-endian f32 data files.

Runs ONLY in the build container (it reads /root/reference, which does not exist on
the GPU box).  Its outputs are committed:

  raptor_amd/data/raptor_policy.bin     2 084 f32 actor parameters, order below
  tests/golden/kat_h_input.bin          known-answer input  #1  [500,2,22] f32
  tests/golden/kat_h_output.bin         known-answer output #1  [500,2,4]  f32
  tests/golden/kat_h5_input.bin         known-answer input  #2  [500,2,22] f32
  tests/golden/kat_h5_output.bin        known-answer output #2  [500,2,4]  f32
  tests/golden/checkpoint.h5            the published HDF5 checkpoint itself (a 142 KB data file:
                                        weights + known-answer pair #2), fixture of the HDF5 reader
  tests/golden/MANIFEST.json            sha256 + shape of every file above

Sources (members of /root/reference/data/raptor-policy-checkpoint.tar.gz, directory
2025-04-19_16-16-17/):
  checkpoint.h   byte arrays `alignas(float) const unsigned char memory[] = {...}`
                 at lines 39,50 (layer_0 W[16,22], b[16]); 75,87,99,111,123 (layer_1
                 W_input[48,16], W_hidden[48,16], b_input[48], b_hidden[48],
                 initial_hidden_state[16]); 149,160 (layer_2 W[4,16], b[4]);
                 199 (example input), 210 (example output).
  checkpoint.h5  /example/input, /example/output (second, different known-answer pair);
                 /actor/layers/* used only to cross-check the .h weights.

Only numeric payloads are extracted (weights and example I/O are data, MIT licensed,
(c) 2025 Jonas Eschmann); no reference source text is copied.

Parameter order in raptor_policy.bin (all row-major, (out,in)):
  W0[16,22] b0[16] Wi[48,16] Wh[48,16] bi[48] bh[48] h0[16] W2[4,16] b2[4]
"""
import hashlib
import json
import os
import re
import subprocess
import sys
import tarfile
import tempfile

import numpy as np

REF_TAR = "/root/reference/data/raptor-policy-checkpoint.tar.gz"
CK_DIR = "2025-04-19_16-16-17"
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
H5DUMP = "/opt/conda/bin/h5dump"

ORDER = [
    ("actor/layer_0/weights", (16, 22)),
    ("actor/layer_0/biases", (16,)),
    ("actor/layer_1/weights_input", (48, 16)),
    ("actor/layer_1/weights_hidden", (48, 16)),
    ("actor/layer_1/biases_input", (48,)),
    ("actor/layer_1/biases_hidden", (48,)),
    ("actor/layer_1/initial_hidden_state", (16,)),
    ("actor/layer_2/weights", (4, 16)),
    ("actor/layer_2/biases", (4,)),
]
H5_PATHS = [
    "/actor/layers/0/weights/parameters", "/actor/layers/0/biases/parameters",
    "/actor/layers/1/weights_input/parameters", "/actor/layers/1/weights_hidden/parameters",
    "/actor/layers/1/biases_input/parameters", "/actor/layers/1/biases_hidden/parameters",
    "/actor/layers/1/initial_hidden_state/parameters",
    "/actor/layers/2/weights/parameters", "/actor/layers/2/biases/parameters",
]


def parse_checkpoint_h(text):
    """Return {namespace path: float32 array} for every `memory[] = {...}` blob."""
    blobs = {}
    stack = []
    ns_re = re.compile(r"^\s*namespace\s+([\w:]+)\s*\{\s*$")
    for line in text.split("\n"):
        m = ns_re.match(line)
        if m:
            name = m.group(1)
            name = name.replace("rl_tools::checkpoint::", "").replace("::", "/")
            stack.append(name)
            continue
        if line.strip() == "}":
            if stack:
                stack.pop()
            continue
        if "const unsigned char memory[]" in line:
            body = line[line.index("{") + 1: line.rindex("}")]
            raw = np.array([int(t) for t in body.split(",")], dtype=np.uint8)
            path = "/".join(s for s in stack if s != "parameters_memory")
            blobs[path] = raw.view("<f4").copy()
    return blobs


def h5_dataset(h5file, path, tmpdir):
    out = os.path.join(tmpdir, "d.bin")
    subprocess.run([H5DUMP, "-d", path, "-b", "LE", "-o", out, h5file],
                   check=True, stdout=subprocess.DEVNULL)
    return np.fromfile(out, dtype="<f4")


def main():
    if not os.path.exists(REF_TAR):
        sys.exit("reference tarball not present (this script runs in the build container only)")
    gold = os.path.join(REPO, "tests", "golden")
    data = os.path.join(REPO, "raptor_amd", "data")
    os.makedirs(gold, exist_ok=True)
    os.makedirs(data, exist_ok=True)
    with tempfile.TemporaryDirectory() as td:
        with tarfile.open(REF_TAR) as tf:
            tf.extractall(td)
        ck = os.path.join(td, CK_DIR)
        blobs = parse_checkpoint_h(open(os.path.join(ck, "checkpoint.h")).read())
        parts = []
        for (path, shape), h5p in zip(ORDER, H5_PATHS):
            a = blobs[path]
            assert a.size == int(np.prod(shape)), (path, a.size, shape)
            b = h5_dataset(os.path.join(ck, "checkpoint.h5"), h5p, td)
            assert b.size == a.size and np.array_equal(a, b), f".h and .h5 weights differ at {path}"
            parts.append(a)
        weights = np.concatenate(parts).astype("<f4")
        assert weights.size == 2084
        files = {
            os.path.join(data, "raptor_policy.bin"): (weights, [2084]),
            os.path.join(gold, "kat_h_input.bin"): (blobs["example/input"], [500, 2, 22]),
            os.path.join(gold, "kat_h_output.bin"): (blobs["example/output"], [500, 2, 4]),
            os.path.join(gold, "kat_h5_input.bin"):
                (h5_dataset(os.path.join(ck, "checkpoint.h5"), "/example/input", td), [500, 2, 22]),
            os.path.join(gold, "kat_h5_output.bin"):
                (h5_dataset(os.path.join(ck, "checkpoint.h5"), "/example/output", td), [500, 2, 4]),
        }
        manifest = {}
        for fn, (arr, shape) in files.items():
            assert arr.size == int(np.prod(shape)), (fn, arr.size)
            arr.astype("<f4").tofile(fn)
            manifest[os.path.relpath(fn, REPO)] = {
                "shape": shape, "dtype": "<f4",
                "sha256": hashlib.sha256(arr.astype("<f4").tobytes()).hexdigest()}
        import shutil
        shutil.copy(os.path.join(ck, "checkpoint.h5"), os.path.join(gold, "checkpoint.h5"))
        manifest["tests/golden/checkpoint.h5"] = {
            "shape": [os.path.getsize(os.path.join(gold, "checkpoint.h5"))], "dtype": "bytes",
            "sha256": hashlib.sha256(open(os.path.join(gold, "checkpoint.h5"), "rb").read()).hexdigest()}
        manifest["_source"] = {
            "tarball": "data/raptor-policy-checkpoint.tar.gz", "dir": CK_DIR,
            "checkpoint_name": "logs/2025-04-19_16-16-17",
            "commit_hash": "c9bcfde8acd3f0d616edbfc3ba5a53d4497c7fa7",
            "observation": "Position.OrientationRotationMatrix.LinearVelocity."
                           "AngularVelocityDelayed(0).ActionHistory(1)",
            "license": "MIT (c) 2025 Jonas Eschmann"}
        with open(os.path.join(gold, "MANIFEST.json"), "w") as f:
            json.dump(manifest, f, indent=1, sort_keys=True)
        same = np.array_equal(blobs["example/input"],
                              np.fromfile(os.path.join(gold, "kat_h5_input.bin"), "<f4"))
        print("weights:", weights.size, "h==h5 weights: True; KAT#1 == KAT#2 inputs:", same)


if __name__ == "__main__":
    main()
