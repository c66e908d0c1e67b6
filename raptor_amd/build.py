"""In-tree build of libraptor_quad.so (hipcc, gfx950 only).

    python -m raptor_amd.build            # incremental
    python -m raptor_amd.build --force

hipcc cross-compiles without a GPU; the resulting .so sits next to this file so it travels
with the source tree (it is git-ignored, not gpurun-ignored).
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
INCLUDE = os.path.join(os.path.dirname(PKG), "include")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(PKG, "libraptor_quad.so")

ARCH = "gfx950"
# -ffp-contract=off: only explicit fmaf() calls fuse (DESIGN.md "Arithmetic contract")
DEVICE_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-mllvm", "-amdgpu-mfma-vgpr-form=1", f"--offload-arch={ARCH}",
                "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]
DEVICE_FLAGS += os.environ.get("RQ_EXTRA_HIPCC_FLAGS", "").split()      # compiler-flag experiments only
if os.environ.get("RQ_NO_MFMA_VGPR_FORM"):                                 # (an -mllvm option cannot be given twice to override it)
    _i = DEVICE_FLAGS.index("-amdgpu-mfma-vgpr-form=1")
    del DEVICE_FLAGS[_i - 1:_i + 1]
SOURCES = ["rq_kernels.hip", "rq_kernels_16bit.hip", "rq_teacher.hip", "rq_capi.cpp", "rq_comm.cpp", "rq_pack.cpp"]
HEADERS = ["rq_kernels.hpp", "rq_device_math.hpp", "rq_rollout.hpp", "rq_host.hpp", os.path.join(INCLUDE, "raptor_quad.h")]
# per-source flags.  Round 4 built rq_kernels_16bit.hip with -mllvm -amdgpu-sched-strategy=max-ilp (a lone wave stalls ~3 cycles
# when an instruction reads the result of the one right before it, tools/lonewave.hip; 24 -> 5 such pairs in the bf16 loop).  With
# it the TWO-waves-per-SIMD bf16 build (ActorBF16Lean, > 65 536 envs) gave run-to-run different results - whole 16-env tiles, with
# and without auto-reset, default scheduler: never (tests/test_gpu_parity.py::test_fused_rollout_is_deterministic) - and the gain
# on the one-wave build was inside the box-to-box spread (1.449 -> 1.406 us/step on one box, nothing on the next): dropped.
SOURCE_FLAGS = {}

def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libraptor_quad.so cannot be built")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, force, obj_dir=OBJ, extra=()):
    obj = os.path.join(obj_dir, os.path.splitext(src)[0] + ".o")
    deps = [os.path.join(CSRC, src)] + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    if not force and not _stale(obj, deps):
        return obj, False
    cmd = [_hipcc()] + DEVICE_FLAGS + SOURCE_FLAGS.get(src, []) + list(extra) + ["-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
    return obj, True


def build(force=False, verbose=False, variant=None, extra_flags=()):
    """variant: an experiment build beside the product's - same sources compiled with `extra_flags` (-D switches of an
    A/B) into scratch/variants/libraptor_quad_<variant>.so; RAPTOR_QUAD_LIB=<that path> makes raptor_amd load it."""
    obj_dir, lib = OBJ, LIB
    if variant:
        vdir = os.path.join(os.path.dirname(PKG), "scratch", "variants")
        obj_dir, lib = os.path.join(vdir, "_obj_" + variant), os.path.join(vdir, f"libraptor_quad_{variant}.so")
    os.makedirs(obj_dir, exist_ok=True)
    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        results = list(ex.map(lambda s: _compile(s, force, obj_dir, extra_flags), SOURCES))
    objs = [o for o, _ in results]
    LIB_ = lib
    if force or any(c for _, c in results) or _stale(LIB_, objs):
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB_] + objs + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose:
            print("built", LIB_)
    elif verbose:
        print("up to date:", LIB_)
    return LIB_


if __name__ == "__main__":
    # python -m raptor_amd.build [--force] [--variant NAME -DFLAG ...]
    _variant = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else None
    _extra = [a for a in sys.argv[1:] if a not in ("--force", "--variant", _variant)]
    build(force="--force" in sys.argv, verbose=True, variant=_variant, extra_flags=_extra)


def listings(out_dir=None):
    """gfx950 assembly listings of the kernel sources (hipcc -S, same flags as the build) -> [paths]; cached by the
    sources' modification times.  What tools/mfma_hazard_lint.py and the instruction-mix tools read."""
    out_dir = out_dir or os.path.join(CSRC, "_obj")
    os.makedirs(out_dir, exist_ok=True)
    newest = max(os.path.getmtime(os.path.join(CSRC, h) if not os.path.isabs(h) else h) for h in HEADERS + SOURCES)
    out = []
    for src in SOURCES:
        if not src.endswith(".hip"):
            continue
        lst = os.path.join(out_dir, os.path.splitext(src)[0] + ".s")
        if not os.path.exists(lst) or os.path.getmtime(lst) < newest:
            cmd = [_hipcc()] + DEVICE_FLAGS + SOURCE_FLAGS.get(src, []) + ["-x", "hip", "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", lst]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        out.append(lst)
    return out
