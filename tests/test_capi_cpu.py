"""CPU-side checks of the C-ABI library: it loads, exports every declared symbol, agrees with
the oracle on the configuration defaults, and fails loudly without a GPU (no fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import HAS_GPU, ROOT


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "raptor_quad.h")).read()
    return sorted(set(re.findall(r"^RQ_API\s+[\w\s\*]+?\b(rq_\w+)\(", hdr, flags=re.M)))


def test_library_exports_every_declared_symbol():
    from raptor_amd import _lib
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 55
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/raptor_quad.h but not exported"
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared, set(declared) ^ set(_lib.EXPORTED_SYMBOLS)


def test_abi_version_and_status_strings():
    from raptor_amd import _lib
    lib = _lib.load()
    assert lib.rq_abi_version() == 5
    assert lib.rq_status_string(0) == b"ok"
    assert b"device" in lib.rq_status_string(-2)


def test_default_config_matches_oracle(oracle):
    from raptor_amd import _lib
    cfg = _lib.EnvConfig()
    _lib.call("rq_env_default_config", ctypes.byref(cfg))
    ref = oracle.default_config()
    assert ctypes.sizeof(cfg) == ctypes.sizeof(ref) == cfg.struct_size == 148
    assert bytes(cfg) == bytes(ref)


def test_field_enums_match_header():
    hdr = open(os.path.join(ROOT, "include", "raptor_quad.h")).read()
    from raptor_amd import _lib
    assert int(re.search(r"RQ_PARAM_DIM = (\d+)", hdr).group(1)) == _lib.PARAM_DIM
    assert int(re.search(r"RQ_STATE_DIM = (\d+)", hdr).group(1)) == _lib.STATE_DIM
    assert int(re.search(r"#define RQ_OBSERVATION_DIM (\d+)", hdr).group(1)) == _lib.OBSERVATION_DIM
    assert int(re.search(r"#define RQ_POLICY_NUM_WEIGHTS (\d+)", hdr).group(1)) == _lib.POLICY_NUM_WEIGHTS


def test_null_arguments_are_rejected_not_crashing():
    from raptor_amd import _lib
    lib = _lib.load()
    assert lib.rq_device_count(None) == -1
    assert lib.rq_env_default_config(None) == -1
    assert lib.rq_device_create(0, None) == -1
    assert b"null" in lib.rq_last_error()
    assert lib.rq_device_destroy(None) == 0
    assert lib.rq_env_destroy(None) == 0


@pytest.mark.skipif(HAS_GPU, reason="only meaningful on a box without a GPU")
def test_no_gpu_means_loud_failure():
    import raptor_amd.l2f as l2f
    from raptor_amd.foundation_policy import Raptor
    with pytest.raises(l2f.RaptorQuadError) as e:
        l2f.Device()
    assert e.value.status == -2
    with pytest.raises(l2f.RaptorQuadError):
        Raptor().evaluate_step(np.zeros((2, 22), np.float32))


def test_product_never_touches_the_oracle():
    """The product path must not import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "raptor_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "raptor_oracle" not in txt and "libraptor_oracle" not in txt, fn
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), fn
    import subprocess
    out = subprocess.run(["ldd", os.path.join(pkg, "libraptor_quad.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out and "amdhip64" in out


def test_vector_module_surface():
    """Names of the l2f vector API (README.md:44-61,96-99)."""
    import raptor_amd.l2f as l2f
    from raptor_amd.l2f import vector8 as vector
    for name in ("VectorRng", "VectorEnvironment", "VectorParameters", "VectorState", "initialize_rng",
                 "initialize_environment", "sample_initial_parameters", "sample_initial_state", "observe", "step"):
        assert hasattr(vector, name), name
    env = vector.VectorEnvironment()
    assert env.N_ENVIRONMENTS == 8 and env.OBSERVATION_DIM >= 22
    assert l2f.vector(65536).N_ENVIRONMENTS == 65536
    from raptor_amd.foundation_policy import Raptor
    assert hasattr(Raptor, "reset") and hasattr(Raptor, "evaluate_step")


def test_header_is_plain_c_and_the_c_example_links(tmp_path):
    """include/raptor_quad.h compiles as C11 (no C++ in the boundary) and a plain-C program written
    against it links with libraptor_quad.so."""
    import subprocess
    pkg = os.path.join(ROOT, "raptor_amd")
    exe = str(tmp_path / "readme_loop")
    r = subprocess.run(["gcc", "-std=c11", "-O1", "-Wall", "-Wextra", "-pedantic", "-Werror",
                        "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "readme_loop.c"),
                        "-L" + pkg, "-lraptor_quad", "-Wl,-rpath," + pkg, "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    if not HAS_GPU:
        run = subprocess.run([exe, os.path.join(pkg, "data", "raptor_policy.bin")], capture_output=True, text=True)
        assert run.returncode == 1 and "no HIP device" in run.stderr      # loud, not a crash


def test_cpp_wrapper_example_links(tmp_path):
    """include/raptor_quad.hpp (header-only C++17 layer with the reference's free-function shape)."""
    import subprocess
    pkg = os.path.join(ROOT, "raptor_amd")
    exe = str(tmp_path / "readme_loop_cpp")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-pedantic", "-Werror",
                        "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "readme_loop.cpp"),
                        "-L" + pkg, "-lraptor_quad", "-Wl,-rpath," + pkg, "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    if not HAS_GPU:
        run = subprocess.run([exe, os.path.join(pkg, "data", "raptor_policy.bin")], capture_output=True, text=True)
        assert run.returncode == 1 and "no HIP device" in run.stderr


def test_comm_entry_points_validate_and_fail_loudly_without_a_gpu():
    """The RCCL exchange of the C++ host (rq_comm_*): argument checks work anywhere; without a GPU the id cannot
    be produced and the call says so (no fallback transport exists)."""
    from raptor_amd import _lib
    lib = _lib.load()
    buf = ctypes.create_string_buffer(_lib.COMM_ID_BYTES)
    assert lib.rq_comm_unique_id(None, 128) == -1
    assert lib.rq_comm_unique_id(buf, 64) == -1 and b"128" in lib.rq_last_error()
    h = ctypes.c_void_p()
    assert lib.rq_comm_create(None, 2, 0, buf, 128, ctypes.byref(h)) == -1
    assert lib.rq_allgather_returns(None, None) == -1
    assert lib.rq_comm_gathered(None, None, None, None) == -1
    assert lib.rq_comm_destroy(None) == 0
    if not HAS_GPU:
        # RCCL absent (-2), no device behind it (-3, ROCm's build) - or an id, which some RCCL builds (the one
        # PyTorch bundles, shared when torch is already imported) hand out before any device is touched
        rc = lib.rq_comm_unique_id(buf, 128)
        assert rc in (0, -2, -3)
        assert (rc == 0 and any(buf.raw)) or (rc != 0 and len(lib.rq_last_error()) > 0)


def test_product_library_holds_no_two_wave_bf16_kernel():
    """The two-waves-per-SIMD bf16 actor build (ActorBF16Lean) gave run-to-run different results under another instruction schedule;
    round 5 found the cause (a gfx950 fault only two waves of a SIMD with 16-bit MFMAs can meet, profiles/r05_bf16_two_wave_hunt.md)
    and the build was the slower one anyway: it exists in experiment builds only.  The product library must not contain a kernel
    instantiated with it, and rq_comm_describe / rq_comm_info refuse null handles."""
    from raptor_amd import _lib
    lib = _lib.load()
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"ActorBF16E" in blob                           # the one-wave build is there (mangled kernel names are in the code object)
    assert b"ActorBF16Lean" not in blob
    assert lib.rq_comm_describe(None, None) == -1 and lib.rq_comm_info(None, None, None) == -1


def test_split_f16_operand_image_reconstructs_the_weights():
    """RQ_POLICY_F16X2_MFMA keeps every weight as two f16 numbers, hi = f16(w') and lo = f16(w' - hi) with w'
    the weight after the gate pre-scaling.  rq_policy_pack_image is host code: decode the image with numpy's float16
    and check hi + lo against the operands themselves, element by element (2^-21 relative or 2^-25 absolute where the
    residual is an f16 subnormal), and that the f32 / bf16 images come out with their documented sizes."""
    import ctypes as C
    from raptor_amd import _lib
    w = np.fromfile(os.path.join(ROOT, "raptor_amd", "data", "raptor_policy.bin"), dtype="<f4")
    assert w.size == 2084

    def image(precision):
        need = C.c_size_t()
        _lib.call("rq_policy_pack_image", w.ctypes.data, w.size, precision, None, 0, C.byref(need))
        img = np.zeros(need.value, np.float32)
        _lib.call("rq_policy_pack_image", w.ctypes.data, w.size, precision, img.ctypes.data, img.size, C.byref(need))
        return img.reshape(-1, 64)

    assert image(_lib.POLICY_FP32).shape[0] == 72 and image(_lib.POLICY_BF16_MFMA).shape[0] == 60     # f32: 18 quads of 70 + 2 padding images
    img = image(_lib.POLICY_F16X2_MFMA)
    assert img.shape == (96, 64)
    halves = img[:72].view(np.uint32)
    def pieces(base):                         # 4 dwords x 64 lanes -> [lane, 8] float16 values
        d = halves[base:base + 4]             # [4, 64]
        lo16 = (d & 0xFFFF).astype(np.uint16).view(np.float16)
        hi16 = (d >> 16).astype(np.uint16).view(np.float16)
        return np.stack([lo16, hi16], axis=-1).transpose(1, 0, 2).reshape(64, 8)       # element e = 2 * dword + half
    W0, B0, WI, WH = w[:352].reshape(16, 22), w[352:368], w[368:1136].reshape(48, 16), w[1136:1904].reshape(48, 16)
    W2 = w[2016:2080].reshape(4, 16)
    kS, kT = np.float32(-1.4426950408889634), np.float32(-2.8853900817779268)
    hi, lo = pieces(0), pieces(4)             # layer_0
    assert np.isfinite(hi.astype(np.float64)).all()
    for lane in range(64):
        q, i = lane >> 4, lane & 15
        for e in range(8):
            f = 4 * e + q
            want = 0.0 if e >= 6 or f == 23 else (B0[i] if f == 22 else W0[i, f])
            got = float(hi[lane, e]) + float(lo[lane, e])
            assert abs(got - float(want)) <= max(2.0 ** -21 * abs(float(want)), 2.0 ** -25), (lane, e)
    for base, rows, scale in ((8, 0, kS), (16, 16, kS)):                   # r and z gates: [W_i | W_h] rows, pre-scaled
        hi, lo = pieces(base), pieces(base + 4)
        for lane in range(64):
            q, i = lane >> 4, lane & 15
            for e in range(8):
                want = float(np.float32(scale * (WI[rows + i, 4 * q + e] if e < 4 else WH[rows + i, 4 * q + e - 4])))
                got = float(hi[lane, e]) + float(lo[lane, e])
                assert abs(got - want) <= max(2.0 ** -21 * abs(want), 2.0 ** -25), (base, lane, e)
    hi, lo = pieces(40 + 4 * 2), pieces(56 + 4 * 2)                      # layer_2, tile 2: rows 8..11, k-slots 4..7
    for lane in range(64):
        q, i = lane >> 4, lane & 15
        for e in range(8):
            want = float(W2[i & 3, 4 * q + e - 4]) if (e >= 4 and (i >> 2) == 2) else 0.0
            got = float(hi[lane, e]) + float(lo[lane, e])
            assert abs(got - want) <= max(2.0 ** -21 * abs(want), 2.0 ** -25), (lane, e)


def test_no_mfma_result_is_touched_before_it_is_final():
    """tools/mfma_hazard_lint.py over the listings of both kernel sources: no instruction - inline asm or the compiler's
    own, on any path out of the MFMA, a taken branch included - reads or overwrites an MFMA's destination earlier than
    the wait states hipcc itself leaves on straight-line code.  Round 3 shipped (for two commits, caught by the GPU
    tests) a build in which a register move of the fused kernel's carried accumulators sat behind the branch that follows
    the prologue, 4 wait states after the MFMA: hipcc pads along the fall-through path only."""
    import subprocess
    import sys
    from raptor_amd import build
    paths = build.listings()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "mfma_hazard_lint.py")] + paths, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "0 hazard(s)" in r.stdout


def test_no_kernel_is_exposed_to_the_packed_op_sel_fault():
    """tools/opsel_lint.py over the listings of every kernel source.  Measured in round 5 (profiles/r05_bf16_two_wave_hunt.md,
    tools/hazard_probe7.hip): on gfx950 a v_pk_add/mul/fma_f32 whose low result takes src0's low and src1's HIGH dword
    (op_sel:[0,1,..]) reads that dword as 0 in lanes 48..63 while ANOTHER wave of the same SIMD executes a 16- or 8-bit MFMA - the
    cause of the run-to-run differences of the two-waves-per-SIMD bf16 rollout build of rounds 3 - 4.  The build rewrites the form
    into its sound twin in every listing it assembles (raptor_amd/gfx950_errata.py), inline asm included: no kernel may hold it -
    not even the fp32 ones, which could meet another stream's 16-bit MFMAs on a shared SIMD."""
    import subprocess
    import sys
    from raptor_amd import build
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "opsel_lint.py")] + build.listings(), capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "0 kernel(s) hold the form; 0 exposed kernel(s)" in r.stdout, r.stdout[-2000:]


def test_the_op_sel_lint_and_its_rewrite():
    """The lint flags exactly the measured form, and tools/opsel_rewrite.py turns it into the sound twin (operands and their
    modifiers exchanged: the same sum / product)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import opsel_lint
    import opsel_rewrite
    risky = ["v_pk_add_f32 v[18:19], v[20:21], v[34:35] op_sel:[0,1]", "v_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,1] op_sel_hi:[1,0]",
             "v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,1,1] neg_lo:[0,0,1]", "v_pk_add_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,1] op_sel_hi:[0,1]"]
    sound = ["v_pk_add_f32 v[18:19], v[20:21], v[34:35] op_sel:[1,0]", "v_pk_add_f32 v[0:1], v[2:3], v[4:5] op_sel:[1,1]",
             "v_pk_add_f32 v[0:1], v[2:3], v[4:5] op_sel_hi:[1,0]", "v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,0,1]",
             "v_pk_mov_b32 v[0:1], v[2:3], v[4:5] op_sel:[0,1]", "v_pk_add_f16 v0, v1, v2 op_sel:[0,1]", "v_pk_add_f32 v[0:1], v[2:3], v[4:5]"]
    for x in risky:
        assert opsel_lint.RISKY.search(x), x
        y, changed = opsel_rewrite.rewrite("\t" + x)
        assert changed and not opsel_lint.RISKY.search(y.strip()), (x, y)
    for x in sound:
        assert not opsel_lint.RISKY.search(x), x
        assert opsel_rewrite.rewrite("\t" + x) == ("\t" + x, False)
    assert opsel_rewrite.rewrite("\tv_pk_mul_f32 v[0:1], v[4:5], v[2:3] op_sel:[0,1] op_sel_hi:[0,0] neg_hi:[1,0]")[0] == \
        "\tv_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[1,0] op_sel_hi:[0,0] neg_hi:[0,1]"
    # a constant or scalar source has no high half to exchange into: no sound twin exists, and the pass REFUSES (round 6: it used to
    # hand the line back untouched and the build shipped it)
    from raptor_amd.gfx950_errata import ErrataError
    for bad in ("\tv_pk_add_f32 v[0:1], v[2:3], s[4:5] op_sel:[0,1]", "\tv_pk_mul_f32 v[0:1], 1.0, v[4:5] op_sel:[0,1]",
                "\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,1,0]",                          # operand count does not parse
                "\tv_pk_add_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,1,0]",                          # three selectors, two sources
                "label: v_pk_add_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,1]",                       # a line the pass cannot take apart
                "\tv_pk_add_f32 v[0:1], v[2:3], v[4:5] op_sel:0,1"):
        with pytest.raises(ErrataError):
            opsel_rewrite.rewrite(bad)


def test_the_op_sel_pass_checks_its_own_output(tmp_path):
    """rewrite_listing(): every line it did not rewrite is byte-identical, line and instruction counts are unchanged, a second
    matcher that shares no regular expression with the pass finds nothing of the form afterwards - and each of those checks fires."""
    from raptor_amd import gfx950_errata as E
    src, dst = tmp_path / "a.s", tmp_path / "b.s"
    body = ["\t.text", "k:", "\tv_pk_add_f32 v[18:19], v[20:21], v[34:35] op_sel:[0,1]", "\ts_nop 0 ; a comment naming v_pk_add_f32 op_sel:[0,1]",
            "\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,1,1] neg_lo:[0,0,1]", "\tv_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[1,0]", "\ts_endpgm"]
    src.write_text("\n".join(body) + "\n")
    assert E.rewrite_listing(str(src), str(dst)) == 2
    out = dst.read_text().splitlines()
    assert len(out) == len(body) and [i for i, (a, b) in enumerate(zip(body, out)) if a != b] == [2, 4]
    assert out[2] == "\tv_pk_add_f32 v[18:19], v[34:35], v[20:21] op_sel:[1,0]"
    assert not any(E.mentions_form(x) for x in out) and E.mentions_form(body[2]) and not E.mentions_form(body[3])
    # a rewrite() that goes blind (here: replaced by one that never changes anything) is caught by the second matcher
    real = E.rewrite
    try:
        E.rewrite = lambda line: (line, False)
        with pytest.raises(E.ErrataError, match="still holds"):
            E.rewrite_listing(str(src), str(dst))
        E.rewrite = lambda line: ((line + " ", False) if "s_endpgm" in line else real(line))
        with pytest.raises(E.ErrataError, match="changed although"):
            E.rewrite_listing(str(src), str(dst))
    finally:
        E.rewrite = real


def _codeobj_check():
    import importlib.util
    spec = importlib.util.spec_from_file_location("codeobj_check", os.path.join(ROOT, "tools", "codeobj_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_shipped_code_objects_hold_no_faulty_packed_op_sel_encoding():
    """The gate that is independent of the rewrite pass: the gfx950 code objects are taken out of the library
    that ships, llvm-objdump finds the instruction boundaries, and the VOP3P machine words are decoded here (opcode 0x30 / 0x31 / 0x32,
    op_sel bits [13:11]) - not the listing, not the pass's regular expression.  The decoder's opcode table is cross-checked against
    the disassembler's mnemonics so that it cannot pass by having gone blind."""
    from raptor_amd import _lib
    C = _codeobj_check()
    ok, lines, total = C.check_library(_lib.LIB_PATH)
    assert ok, "\n".join(lines)
    assert total["code_objects"] == 3 and total["faulty"] == 0
    assert total["packed_f32"] > 20000 and total["instructions"] > 300000          # the hand-packed env step is in there
    # the decoder against encodings read off a disassembly by hand
    assert C.decode(0xD3B24202) == ("v_pk_add_f32", 0, 0, 0)           # neg_hi:[0,1], op_sel_hi[2]
    assert C.decode(0xD3B14806) == ("v_pk_mul_f32", 1, 0, 0)           # op_sel:[1,0]
    assert C.decode(0xD3B05012) == ("v_pk_fma_f32", 0, 1, 0)           # op_sel:[0,1,0]: the faulty form
    assert C.decode(0xD3B34800) is None and C.decode(0xD3D400DA) is None and C.decode(0x68000002) is None      # v_pk_mov_b32, an MFMA, a VOP2


@pytest.mark.timeout(300)
def test_the_code_object_gate_fires_on_the_measured_repro(tmp_path):
    """Negative control: tools/opsel_repro.hip (the smallest program that shows the fault on the MI355X) holds the form in inline asm;
    built the way any hipcc user would build it, the gate must find it in the finished binary."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "opsel_repro")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", os.path.join(ROOT, "tools", "opsel_repro.hip"), "-o", exe], check=True)
    C = _codeobj_check()
    blobs = C.code_objects(exe)
    assert len(blobs) == 1
    res = C.check_code_object(blobs[0])
    assert res["packed_f32"] == res["mnemonics"] and len(res["faulty"]) >= 1
    assert all("op_sel:[0,1]" in text for _, _, text in res["faulty"])
    ok, lines, _ = C.check_library(exe)
    assert not ok


def test_the_experiment_patch_still_applies_to_the_product_sources(tmp_path):
    """What used to sit behind #ifdef RQ_BF16_FUSED_LEAN / RQ_DEBUG_* / RQ_PK_PLAIN_C in the kernel sources is
    tools/variants/hunt_experiments.patch, applied to a COPY of csrc/ by experiment builds (raptor_amd.build --variant --patch,
    tools/hazard_variants.sh).  The product sources hold no experiment switch, and the patch must keep applying."""
    import glob
    import shutil
    import subprocess
    csrc = os.path.join(ROOT, "raptor_amd", "csrc")
    for path in glob.glob(os.path.join(csrc, "*.h*")) + glob.glob(os.path.join(csrc, "*.cpp")):
        text = open(path).read()
        assert "#ifdef" not in text and "#if defined" not in text and "RQ_DEBUG" not in text, path
    dst = tmp_path / "raptor_amd" / "csrc"
    shutil.copytree(csrc, dst, ignore=shutil.ignore_patterns("_obj"))
    r = subprocess.run(["patch", "-p1", "-s", "-d", str(tmp_path / "raptor_amd"), "-i", os.path.join(ROOT, "tools", "variants", "hunt_experiments.patch")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ActorBF16Lean" in (dst / "rq_kernels_16bit.hip").read_text()


def test_the_hazard_lint_sees_a_move_behind_a_taken_branch():
    """The lint on a reduced rendition of the build it was written for: the move on the taken path (4 and 5 wait states behind
    the MFMA) is reported, the padded variant is not."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "mfma_hazard_lint.py"),
                        os.path.join(ROOT, "tests", "golden", "mfma_hazard_bad.s")], capture_output=True, text=True)
    assert r.returncode == 1
    assert "2 hazard(s)" in r.stdout and "bad_prime" in r.stdout and "good_prime" not in r.stdout


def test_the_python_helpers_direct_calls_match_the_header_and_decline_what_they_do_not_take(tmp_path):
    """raptor_amd/csrc/rq_pyfast.c calls four entry points of the library by ADDRESS, with the signatures written out in that file
    (it does not link against the library).  Held to include/raptor_quad.h here: a C file that assigns the header's functions to the
    helper's function-pointer types must compile without a diagnostic.  And the helper declines - status 1, nothing called - whatever
    the fast path does not take: missing handles, another dtype, a strided observation buffer, a wrong shape."""
    import re
    import subprocess
    src = open(os.path.join(ROOT, "raptor_amd", "csrc", "rq_pyfast.c")).read()
    typedefs = re.findall(r"^typedef int \(\*\w+_fn\)\([^;]*\);", src, re.M)
    assert len(typedefs) == 5
    c = tmp_path / "sig.c"
    c.write_text('#include <stdint.h>\n#include "raptor_quad.h"\n' + "\n".join(typedefs).replace("void*", "void *") + """
/* the helper's types hold the handles as void pointers: the header's functions converted to them must differ in pointee types only */
int main(void) {
    observe_fn a = (observe_fn)rq_observe; evaluate_step_fn b = (evaluate_step_fn)rq_policy_evaluate_step;
    step_fn c = (step_fn)rq_step; assign_fn d = (assign_fn)rq_state_assign; rollout_fn e = (rollout_fn)rq_rollout;
    _Static_assert(sizeof(int (*)(rq_device*, rq_env*, const rq_params*, const rq_state*, float*, rq_rng*)) == sizeof(observe_fn), "");
    int (*pa)(rq_device*, rq_env*, const rq_params*, const rq_state*, float*, rq_rng*) = rq_observe;
    int (*pb)(rq_policy*, rq_env*, const float*, uint32_t, uint32_t, float*) = rq_policy_evaluate_step;
    int (*pc)(rq_device*, rq_env*, const rq_params*, const rq_state*, const float*, rq_state*, rq_rng*, float*) = rq_step;
    int (*pd)(rq_state*, const rq_state*) = rq_state_assign;
    int (*pe)(rq_device*, rq_env*, const rq_params*, rq_state*, rq_policy*, rq_rng*, uint32_t, int, uint32_t) = rq_rollout;
    return (a && b && c && d && e && pa && pb && pc && pd && pe) ? 0 : 1;
}
""")
    r = subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-c", str(c), "-I", os.path.join(ROOT, "include"), "-o", str(tmp_path / "sig.o")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # the argument lists the helper's typedefs declare, against the header's, position by position (pointer-ness and integer width)
    hdr = open(os.path.join(ROOT, "include", "raptor_quad.h")).read()
    for name, td in zip(("rq_observe", "rq_policy_evaluate_step", "rq_step", "rq_state_assign", "rq_rollout"), typedefs):
        decl = re.search(r"RQ_API int %s\(([^;]*)\);" % name, hdr, re.S).group(1)
        want = ["*" in a for a in decl.split(",")]
        got = ["*" in a for a in td[td.index(")(") + 2:-2].split(",")]
        assert len(want) == len(got) and all(w == g for w, g in zip(want, got)), (name, decl, td)
    from raptor_amd import _lib
    if _lib.fast is None:
        pytest.skip("the helper is not built here")
    import ctypes as C
    f, obs, act = _lib.fast, np.zeros((8, 26), np.float32), np.zeros((8, 4), np.float32)
    h, none = C.c_void_p(4096), C.c_void_p(0)
    assert f.NOT_HANDLED == 1
    assert f.observe(1, none, h, h, h, obs, h, 8, 26) == 1 and f.observe(1, h, h, None, h, obs, h, 8, 26) == 1
    assert f.observe(1, h, h, h, h, obs[:, :22], h, 8, 26) == 1 and f.observe(1, h, h, h, h, obs.astype(np.float64), h, 8, 26) == 1
    assert f.observe(1, h, h, h, h, obs, h, 9, 26) == 1 and f.observe(1, h, h, h, h, [[0.0] * 26] * 8, h, 8, 26) == 1
    ro = obs.copy(); ro.setflags(write=False)
    assert f.observe(1, h, h, h, h, ro, h, 8, 26) == 1
    assert f.step(1, h, h, h, h, act[:, :3], h, h, 8) == 1 and f.step(1, h, h, h, h, act, None, h, 8) == 1
    assert f.step(1, h, h, h, h, np.zeros((8, 8), np.float32)[:, :4], h, h, 8) == 1
    assert f.evaluate_step(1, None, obs[:, :22], act, 22) == 1 and f.evaluate_step(1, h, obs[:, :21], act, 22) == 1
    assert f.evaluate_step(1, h, obs[:, :22], act[:4], 22) == 1 and f.evaluate_step(1, h, obs[:, ::2], act, 13) == 1
    assert f.assign(1, h, None) == 1 and f.assign(1, none, h) == 1
    assert f.rollout(1, h, h, h, h, None, h, 20, 0, 1) == 1 and f.rollout(1, h, h, h, h, h, h, 1 << 40, 0, 1) == 1
    with pytest.raises(TypeError):
        f.observe(1, h)
