"""In-tree build of libraptor_quad.so (hipcc, gfx950 only).

    python -m raptor_amd.build            # incremental
    python -m raptor_amd.build --force
    python -m raptor_amd.build --variant NAME [--patch tools/variants/x.patch ...] [-DFLAG ...]     # experiment builds, below

hipcc cross-compiles without a GPU; the resulting .so sits next to this file so it travels
with the source tree (it is git-ignored, not gpurun-ignored).

The PRODUCT build takes no switches from the environment: one set of flags, the op_sel pass always on, and a gate behind the link
that decodes the code objects of the finished library (tools/codeobj_check.py).  EXPERIMENT builds (`--variant`) are the only ones
that honour RQ_NO_OPSEL_REWRITE / RQ_EXTRA_HIPCC_FLAGS / RQ_NO_MFMA_VGPR_FORM and `--patch`: they compile a patched COPY of csrc/
into scratch/variants/ and never touch the product's objects.
"""
import hashlib
import importlib.util
import os
import re
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
INCLUDE = os.path.join(ROOT, "include")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(PKG, "libraptor_quad.so")

ARCH = "gfx950"
# -ffp-contract=off: only explicit fmaf() calls fuse (DESIGN.md "Arithmetic contract")
DEVICE_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-mllvm", "-amdgpu-mfma-vgpr-form=1", f"--offload-arch={ARCH}",
                "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]
SOURCES = ["rq_kernels.hip", "rq_kernels_16bit.hip", "rq_teacher.hip", "rq_capi.cpp", "rq_capi_vector.cpp", "rq_capi_policy.cpp",
           "rq_capi_rollout.cpp", "rq_capi_teacher.cpp", "rq_comm.cpp", "rq_pack.cpp"]
HEADERS = ["rq_kernels.hpp", "rq_device_math.hpp", "rq_rollout.hpp", "rq_host.hpp", "rq_objects.hpp"]
# per-source flags (none today).  Round 4 built rq_kernels_16bit.hip with -mllvm -amdgpu-sched-strategy=max-ilp; the gain on the bf16 build
# that ships was inside the box-to-box spread, and the two-waves-per-SIMD bf16 build gave run-to-run different results with it - the
# gfx950 fault round 5 found (gfx950_errata.py), which _compile() now rewrites out of every listing whatever the scheduler.
SOURCE_FLAGS = {}
VARIANT_TOGGLES = ("RQ_NO_OPSEL_REWRITE", "RQ_EXTRA_HIPCC_FLAGS", "RQ_NO_MFMA_VGPR_FORM")


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


def _llvm(tool):
    """clang / ld.lld / clang-offload-bundler of the ROCm the hipcc in use belongs to."""
    for base in (os.path.join(os.path.dirname(os.path.realpath(_hipcc())), "..", "lib", "llvm", "bin"), "/opt/rocm/lib/llvm/bin"):
        path = os.path.join(base, tool)
        if os.path.exists(path):
            return path
    raise RuntimeError(f"{tool} not found beside hipcc: libraptor_quad.so cannot be built")


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("build step failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
    return r


_targets = None


def _bundle_targets():
    """(-targets=... of the offload bundle, device triple) as THIS hipcc would write them (`hipcc -###` of an empty unit), so that the
    hand-assembled bundle below is the one its runtime looks for."""
    global _targets
    if _targets is None:
        probe = os.path.join(OBJ, "_probe.hip")
        os.makedirs(OBJ, exist_ok=True)
        open(probe, "w").close()
        r = subprocess.run([_hipcc(), "-###", f"--offload-arch={ARCH}", "-x", "hip", "-c", probe, "-o", os.devnull], capture_output=True, text=True)
        m = re.search(r'"-targets=(host-[^",]+,hip[^",]*' + ARCH + r')"', r.stderr + r.stdout)
        t = re.search(r'"-triple"\s+"(amdgcn[^"]*)"', r.stderr + r.stdout)
        if not m or not t:
            raise RuntimeError("cannot read the offload bundle targets from `hipcc -###`")
        _targets = (m.group(1), t.group(1))
    return _targets


def _compile(src, force, src_dir, obj_dir, flags, rewrite):
    """One translation unit -> object.  .cpp: hipcc -c.  .hip: the device side goes through its LISTING - hipcc -S, the gfx950 op_sel
    pass over it (raptor_amd/gfx950_errata.py: round 5 measured that a packed-fp32 instruction of one op_sel form misreads an operand
    beside another wave's 16-bit MFMA; the pass exchanges its operands, same arithmetic, and raises on anything it cannot prove
    sound), assembler, lld, offload bundle - and the host side is compiled around that bundle.  The listing stays beside the object:
    it is what the lints read (listings())."""
    stem = os.path.splitext(src)[0]
    obj = os.path.join(obj_dir, stem + ".o")
    lst = os.path.join(obj_dir, stem + ".s")
    path = os.path.join(src_dir, src)
    deps = [path] + [os.path.join(src_dir, h) for h in HEADERS] + [os.path.join(INCLUDE, "raptor_quad.h"),
                                                                    os.path.join(PKG, "gfx950_errata.py"), os.path.abspath(__file__)]
    is_hip = src.endswith(".hip")
    if not force and not _stale(obj, deps) and (not is_hip or os.path.exists(lst)):
        return obj, False
    # -cuid: clang derives the id that keeps a unit's internal device symbols apart from the unit's PATH; given by name instead, the
    # library is the same file wherever the tree is built - and the profiles that cite its sha256 (bench.py rocprof_launch_stats) stay its
    flags = flags + SOURCE_FLAGS.get(src, []) + [f"-cuid=rq_{stem}"]
    if not is_hip:
        _run([_hipcc()] + flags + ["-x", "hip", "-c", path, "-o", obj])
        return obj, True
    from . import gfx950_errata
    raw = os.path.join(obj_dir, stem + ".raw.s")
    dev, hsaco, fatbin = (os.path.join(obj_dir, stem + e) for e in (".dev.o", ".hsaco", ".hipfb"))
    targets, triple = _bundle_targets()
    _run([_hipcc()] + flags + ["-x", "hip", "-S", "--cuda-device-only", path, "-o", raw])
    if rewrite:
        gfx950_errata.rewrite_listing(raw, lst)          # raises ErrataError rather than ship what it cannot rewrite
    else:
        shutil.copyfile(raw, lst)
    os.remove(raw)
    _run([_llvm("clang"), "-x", "assembler", "-target", triple, f"-mcpu={ARCH}", "-c", lst, "-o", dev])
    _run([_llvm("ld.lld"), "-shared", "--no-undefined", dev, "-o", hsaco])
    _run([_llvm("clang-offload-bundler"), "-type=o", "-bundle-align=4096", f"-targets={targets}",
          "-input=/dev/null", f"-input={hsaco}", f"-output={fatbin}"])
    _run([_hipcc()] + flags + ["-x", "hip", "--cuda-host-only", "-c", path, "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fatbin, "-o", obj])
    for f in (dev, hsaco, fatbin):
        os.remove(f)
    return obj, True


def _codeobj_check():
    """tools/codeobj_check.py as a module: the gate that shares no code with the op_sel pass (it decodes machine words)."""
    spec = importlib.util.spec_from_file_location("rq_codeobj_check", os.path.join(ROOT, "tools", "codeobj_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _variant_sources(vdir, variant, patches):
    """A copy of csrc/ with the experiment patches applied (tools/variants/*.patch carry what used to sit behind #ifdefs in the
    product sources: the two-waves-per-SIMD bf16 type, register zeroing, HW_ID records, the plain-C++ packed math)."""
    top = os.path.join(vdir, "_src_" + variant)                     # keeps the tree's shape: the sources include ../../include/raptor_quad.h
    shutil.rmtree(top, ignore_errors=True)
    src_dir = os.path.join(top, "raptor_amd", "csrc")
    shutil.copytree(CSRC, src_dir, ignore=shutil.ignore_patterns("_obj"))
    shutil.copytree(INCLUDE, os.path.join(top, "include"))
    for p in patches:
        _run(["patch", "-p1", "-s", "-d", os.path.dirname(src_dir), "-i", os.path.abspath(p)])
    return src_dir


def build(force=False, verbose=False, variant=None, extra_flags=(), patches=()):
    """variant: an experiment build beside the product's - a patched copy of the sources compiled with `extra_flags` (-D switches of an
    A/B) into scratch/variants/libraptor_quad_<variant>.so; RAPTOR_QUAD_LIB=<that path> makes raptor_amd load it."""
    obj_dir, lib, src_dir, flags, rewrite = OBJ, LIB, CSRC, list(DEVICE_FLAGS), True
    if variant:
        vdir = os.path.join(ROOT, "scratch", "variants")
        os.makedirs(vdir, exist_ok=True)
        src_dir = _variant_sources(vdir, variant, patches)
        flags += os.environ.get("RQ_EXTRA_HIPCC_FLAGS", "").split() + list(extra_flags)
        if os.environ.get("RQ_NO_MFMA_VGPR_FORM"):                # (an -mllvm option cannot be given twice to override it)
            i = flags.index("-amdgpu-mfma-vgpr-form=1")
            del flags[i - 1:i + 1]
        rewrite = not os.environ.get("RQ_NO_OPSEL_REWRITE")
        tag = hashlib.sha256(" ".join(flags + [str(rewrite)] + [open(p).read() for p in patches]).encode()).hexdigest()[:8]
        obj_dir, lib = os.path.join(vdir, f"_obj_{variant}_{tag}"), os.path.join(vdir, f"libraptor_quad_{variant}.so")
        force = True                                              # the copy's mtimes are new anyway
    else:
        if patches or extra_flags:
            raise ValueError("patches and extra flags belong to --variant builds: the product is built one way")
        ignored = [k for k in VARIANT_TOGGLES if os.environ.get(k)]
        if ignored and verbose:
            print("ignored for the product build (experiment builds only, --variant):", ", ".join(ignored))
    os.makedirs(obj_dir, exist_ok=True)
    if not variant:
        build_pyfast(force, verbose)
    _bundle_targets()
    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        results = list(ex.map(lambda s: _compile(s, force, src_dir, obj_dir, flags, rewrite), SOURCES))
    objs = [o for o, _ in results]
    if force or any(c for _, c in results) or _stale(lib, objs):
        tmp = lib + ".tmp"
        _run([_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", tmp] + objs + ["-ldl"])
        if rewrite:       # fail closed: a library that holds the faulty form is not installed
            ok, lines, _ = _codeobj_check().check_library(tmp)
            if not ok:
                os.remove(tmp)
                raise RuntimeError("libraptor_quad.so holds the gfx950 packed-fp32 op_sel form (tools/codeobj_check.py):\n" + "\n".join(lines))
        os.replace(tmp, lib)
        if verbose:
            print("built", lib)
    elif verbose:
        print("up to date:", lib)
    return lib


def build_pyfast(force=False, verbose=False):
    """raptor_amd/_rq_fast<EXT_SUFFIX>: the veneer's optional CPython helper (csrc/rq_pyfast.c: array address in 60 ns).  gcc only; a
    missing compiler or Python.h is not an error - raptor_amd/_lib.py then takes addresses the slower way.  -> path or None"""
    import sysconfig
    src = os.path.join(CSRC, "rq_pyfast.c")
    out = os.path.join(PKG, "_rq_fast" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))
    if not force and not _stale(out, [src]):
        return out
    cc = shutil.which("gcc") or shutil.which("cc")
    inc = sysconfig.get_paths().get("include")
    if not cc or not inc or not os.path.exists(os.path.join(inc, "Python.h")):
        return None
    r = subprocess.run([cc, "-O2", "-shared", "-fPIC", "-Wall", "-I", inc, src, "-o", out], capture_output=True, text=True)
    if r.returncode != 0:
        if verbose:
            print("_rq_fast not built:", r.stderr[-500:])
        return None
    if verbose:
        print("built", out)
    return out


def library_sha256(path=LIB):
    """What bench.py and the profile tools record beside a measurement: which build it was made with."""
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def listings():
    """The gfx950 assembly listings the library was assembled from (one per kernel source, written by the build: the compiler's device
    listing after the op_sel pass) -> [paths].  What tools/mfma_hazard_lint.py, tools/opsel_lint.py and the instruction-mix tools read."""
    build()
    paths = [os.path.join(OBJ, os.path.splitext(src)[0] + ".s") for src in SOURCES if src.endswith(".hip")]
    missing = [p for p in paths if not os.path.exists(p)]
    if missing:
        raise RuntimeError("listing(s) missing after a build: " + ", ".join(missing))
    return paths


if __name__ == "__main__":
    # python -m raptor_amd.build [--force] [--variant NAME [--patch FILE ...] -DFLAG ...]
    argv = sys.argv[1:]
    _variant, _patches, _extra, i = None, [], [], 0
    while i < len(argv):
        if argv[i] == "--variant":
            _variant = argv[i + 1]; i += 2
        elif argv[i] == "--patch":
            _patches.append(argv[i + 1]); i += 2
        elif argv[i] == "--force":
            i += 1
        else:
            _extra.append(argv[i]); i += 1
    build(force="--force" in argv, verbose=True, variant=_variant, extra_flags=_extra, patches=_patches)
