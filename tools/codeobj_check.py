#!/usr/bin/env python3
"""The gate for the gfx950 packed-fp32 op_sel fault that does not depend on the pass that removes it.

raptor_amd/gfx950_errata.py rewrites a compiler LISTING; tools/opsel_lint.py reads the same listings with the same regular
expression.  If a future hipcc prints the modifier differently, both go blind together.  This tool looks at what actually ships: it
takes the gfx950 code objects out of the linked library (ELF section .hip_fatbin -> clang offload bundles -> hipv4-amdgcn-amd-amdhsa--gfx950
entries, parsed here), lets llvm-objdump find the instruction boundaries, and decodes the machine words itself:

    VOP3P, word 0:  [31:23] = 0x1A7   [22:16] opcode   [15] clamp   [14] op_sel_hi[2]   [13:11] op_sel[2:0]   [10:8] neg_hi   [7:0] vdst
           word 1:  [8:0] src0   [17:9] src1   [26:18] src2   [28:27] op_sel_hi[1:0]   [31:29] neg_lo
    opcodes (gfx90a / gfx940 / gfx950):  0x30 v_pk_fma_f32   0x31 v_pk_mul_f32   0x32 v_pk_add_f32   (0x33 v_pk_mov_b32: not affected)

The faulty form (tools/hazard_probe7.hip, profiles/r05_bf16_two_wave_hunt.md): one of those three with op_sel[0] = 0 and op_sel[1] = 1.
Nothing is imported from raptor_amd/; the disassembler's TEXT is used for one thing only - a cross-check that the opcode table above
is this toolchain's (the number of words decoded as packed fp32 must equal the number of v_pk_{fma,mul,add}_f32 mnemonics, else the
check itself fails: it never passes by having gone blind).

    python tools/codeobj_check.py raptor_amd/libraptor_quad.so        exit status 1 if any instruction of the form (or a failed cross-check)
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

PK_F32 = {0x30: "v_pk_fma_f32", 0x31: "v_pk_mul_f32", 0x32: "v_pk_add_f32"}
_BUNDLE_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
_RAW = re.compile(r"//\s*([0-9A-Fa-f]+):((?:\s+[0-9A-Fa-f]{8})+)\s*$")
_SYM = re.compile(r"^[0-9a-f]+ <(.+)>:\s*$")


def _objdump():
    for base in (os.environ.get("ROCM_PATH", "/opt/rocm"), "/opt/rocm"):
        p = os.path.join(base, "lib", "llvm", "bin", "llvm-objdump")
        if os.path.exists(p):
            return p
    raise RuntimeError("llvm-objdump of ROCm not found")


def elf_section(data, name):
    """The bytes of section `name` of an ELF64 little-endian image (no external tool)."""
    if data[:4] != b"\x7fELF" or data[4] != 2 or data[5] != 1:
        raise ValueError("not a little-endian ELF64 file")
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)
    def sh(i):
        return struct.unpack_from("<IIQQQQIIQQ", data, shoff + i * shentsize)
    str_off = sh(shstrndx)[4]
    for i in range(shnum):
        h = sh(i)
        end = data.index(b"\0", str_off + h[0])
        if data[str_off + h[0]:end].decode() == name:
            return data[h[4]:h[4] + h[5]]
    raise ValueError(f"no section {name}")


def code_objects(lib_path, arch="gfx950"):
    """Every device code object for `arch` bundled in the library -> [bytes]."""
    fat = elf_section(open(lib_path, "rb").read(), ".hip_fatbin")
    out, at = [], 0
    while True:
        at = fat.find(_BUNDLE_MAGIC, at)
        if at < 0:
            break
        n, = struct.unpack_from("<Q", fat, at + 24)
        p = at + 32
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", fat, p)
            p += 24
            triple = fat[p:p + tlen].decode()
            p += tlen
            if triple.startswith("hip") and triple.endswith(arch) and size:
                out.append(fat[at + off:at + off + size])
        at += len(_BUNDLE_MAGIC)
    return out


def decode(word0):
    """-> (name, op_sel[0], op_sel[1], op_sel[2]) for a packed fp32 fma / mul / add, else None."""
    if (word0 >> 23) != 0x1A7:
        return None
    op = (word0 >> 16) & 0x7F
    if op not in PK_F32:
        return None
    sel = (word0 >> 11) & 7
    return PK_F32[op], sel & 1, (sel >> 1) & 1, (sel >> 2) & 1


def check_code_object(blob, arch="gfx950"):
    """-> dict(instructions, packed_f32, mnemonics, faulty=[(kernel, address, text)])"""
    with tempfile.NamedTemporaryFile(suffix=".elf") as tmp:
        tmp.write(blob)
        tmp.flush()
        r = subprocess.run([_objdump(), "-d", f"--mcpu={arch}", tmp.name], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("llvm-objdump failed: " + r.stderr[-500:])
    res = {"instructions": 0, "packed_f32": 0, "mnemonics": 0, "faulty": [], "kernels": 0}
    kernel = "?"
    for line in r.stdout.splitlines():
        m = _SYM.match(line)
        if m:
            kernel = m.group(1)
            res["kernels"] += 1
            continue
        m = _RAW.search(line)
        if not m:
            continue
        res["instructions"] += 1
        text = line.split("//")[0].strip()
        if text.split()[0].replace("_e64", "") in PK_F32.values():
            res["mnemonics"] += 1
        word0 = int(m.group(2).split()[0], 16)
        d = decode(word0)
        if d is None:
            continue
        res["packed_f32"] += 1
        if d[1] == 0 and d[2] == 1:
            res["faulty"].append((kernel, m.group(1), text))
    return res


def check_library(lib_path, arch="gfx950"):
    """-> (ok, report lines, totals).  ok is False if any instruction of the form ships, if the library holds no code object, or if
    the decoder and the disassembler disagree on what a packed fp32 instruction is."""
    objs = code_objects(lib_path, arch)
    lines, ok = [], True
    total = {"code_objects": len(objs), "instructions": 0, "packed_f32": 0, "faulty": 0}
    if not objs:
        return False, [f"{lib_path}: no {arch} code object found"], total
    for i, blob in enumerate(objs):
        res = check_code_object(blob, arch)
        total["instructions"] += res["instructions"]
        total["packed_f32"] += res["packed_f32"]
        total["faulty"] += len(res["faulty"])
        lines.append(f"code object {i}: {res['kernels']} symbols, {res['instructions']} instructions, {res['packed_f32']} packed fp32 by encoding "
                     f"({res['mnemonics']} by mnemonic), {len(res['faulty'])} of the faulty form")
        if res["packed_f32"] != res["mnemonics"]:
            ok = False
            lines.append("  CROSS-CHECK FAILED: the VOP3P opcode table of this tool is not this toolchain's")
        if res["instructions"] < 1000:
            ok = False
            lines.append("  CROSS-CHECK FAILED: hardly any instruction decoded - the disassembly format changed")
        for kernel, addr, text in res["faulty"]:
            ok = False
            lines.append(f"  FAULTY {addr} {text}    in {kernel[:90]}")
    if total["packed_f32"] == 0:
        ok = False
        lines.append("CROSS-CHECK FAILED: no packed fp32 instruction at all in a library whose env step is written in them")
    return ok, lines, total


def main():
    status = 0
    for path in sys.argv[1:]:
        ok, lines, total = check_library(path)
        print("\n".join(lines))
        print(f"{path}: {total['code_objects']} code objects, {total['instructions']} instructions, {total['packed_f32']} packed fp32, "
              f"{total['faulty']} of the faulty op_sel form -> {'ok' if ok else 'FAILED'}")
        status |= 0 if ok else 1
    return status


if __name__ == "__main__":
    sys.exit(main())
