#!/usr/bin/env python3
"""Static check of a gfx950 listing (hipcc -S --cuda-device-only) for the MFMA hazards hipcc does not cover.

hipcc pads the wait states an MFMA needs only for instructions it can see and - as round 3 found - only along the
fall-through path of a branch that follows the MFMA.  This walks every path out of every v_mfma for as many wait states as
the rule asks and reports

  RAW/WAW  a non-MFMA instruction (inline asm or not) that reads or writes the MFMA's destination earlier than
           PASSES + 2 wait states after it (8-pass 16x16 MFMAs: 10, what hipcc itself pads to);
  WAR      (with --notes; informational) a VALU instruction that writes a register the MFMA reads as its C operand
           (C != D) earlier than 7 wait states after it.  hipcc's own code does this at 0 .. 6 wait states in these very
           listings (it pads this case for other MFMA shapes only), so it is not counted as a hazard.

Wait states: one per instruction, N + 1 for `s_nop N`; `s_waitcnt` counts one (what it may stall for is not relied on).
    python tools/mfma_hazard_lint.py file.s [more.s]        exit status 1 when something is reported
"""
import re
import sys

REG = re.compile(r'\b([va])(?:\[(\d+):(\d+)\]|(\d+)\b)')
NO_DST = ("global_store", "ds_write", "buffer_store", "scratch_store", "flat_store", "v_cmp", "v_cmpx", "s_", "ds_gws",
          "global_atomic", "buffer_atomic", "v_readlane", "v_readfirstlane", "v_nop", "buffer_wbl2", "buffer_inv")
WAR_STATES = 7


def raw_states(op):
    """wait states hipcc itself leaves between this MFMA and a VALU reader of its result (measured on its own code)"""
    if "16x16x4_f32" in op or "32x32" in op:
        return 10
    if "4x4x" in op:
        return 4
    return 8            # 16x16x32 f16 / bf16 and the other 16-bit 16x16 forms


def regs_of(tok):
    out = set()
    for m in REG.finditer(tok):
        bank = m.group(1)
        if m.group(4) is not None:
            out.add((bank, int(m.group(4))))
        else:
            out.update((bank, r) for r in range(int(m.group(2)), int(m.group(3)) + 1))
    return out


class Ins:
    __slots__ = ("line", "op", "dst", "src", "asm", "text", "states", "target", "ends")

    def __init__(self, line, text, in_asm):
        self.line, self.text, self.asm = line, text, in_asm
        parts = text.split(None, 1)
        self.op = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        self.dst, self.src = set(), set()
        first_is_dst = ops and not self.op.startswith(NO_DST)
        for k, o in enumerate(ops):
            (self.dst if (k == 0 and first_is_dst) else self.src).update(regs_of(o))
        if "mac" in self.op:                       # v_fmac / v_pk_fmac: the destination is read as well
            self.src |= self.dst
        m = re.match(r's_nop (\d+)', text)
        self.states = int(m.group(1)) + 1 if m else 1
        self.target = None
        if self.op.startswith("s_cbranch") or self.op == "s_branch":
            self.target = ops[0] if ops else None
        self.ends = self.op in ("s_endpgm", "s_setpc_b64", "s_swappc_b64")


def functions(path):
    name, body, labels, in_asm = None, [], {}, False
    for n, raw in enumerate(open(path), 1):
        line = raw.rstrip("\n")
        t = line.strip()
        m = re.match(r'^(_Z\w+):', line)
        if m:
            if name and body:
                yield name, body, labels
            name, body, labels, in_asm = m.group(1), [], {}, False
            continue
        if name is None:
            continue
        if t.startswith(";;#ASMSTART"):
            in_asm = True; continue
        if t.startswith(";;#ASMEND"):
            in_asm = False; continue
        m = re.match(r'^(\.LBB\w+):', line)
        if m:
            labels[m.group(1)] = len(body); continue
        if not t or t.startswith(";") or t.startswith(".") or t.startswith("//"):
            if t.startswith(".Lfunc_end"):
                yield name, body, labels
                name, body = None, []
            continue
        body.append(Ins(n, t.split(";")[0].strip(), in_asm))
    if name and body:
        yield name, body, labels


def check(path):
    found = []
    for name, body, labels in functions(path):
        for i, mf in enumerate(body):
            if not mf.op.startswith("v_mfma"):
                continue
            d = mf.dst
            c = set()
            ops = mf.text.split(None, 1)[1].split(",")
            if len(ops) >= 4:
                c = regs_of(ops[3]) - d
            need = raw_states(mf.op)
            stack, seen = [(i + 1, 0, frozenset(d))], set()
            while stack:
                j, st, d = stack.pop()
                d = set(d)
                while j < len(body) and st < need and (d or c):
                    if (j, st) in seen:
                        break
                    seen.add((j, st))
                    x = body[j]
                    if not x.op.startswith("v_mfma"):
                        if (x.src | x.dst) & d:
                            found.append((path, name, mf.line, x.line, st, "RAW/WAW", mf.text, x.text, x.asm))
                        elif st < WAR_STATES and x.dst & c and x.op.startswith("v_"):
                            found.append((path, name, mf.line, x.line, st, "WAR", mf.text, x.text, x.asm))
                    else:
                        # a later MFMA that reads or overwrites D is ordered behind this one by the matrix pipe (and by the
                        # compiler's own padding): the window ends for those registers
                        d -= (x.dst | x.src)
                    if x.ends:
                        break
                    st += x.states
                    if x.target is not None and x.target in labels:
                        stack.append((labels[x.target], st, frozenset(d)))
                        if x.op == "s_branch":
                            break
                    j += 1
    return found


if __name__ == "__main__":
    bad = []
    for p in sys.argv[1:]:
        if not p.startswith("--"):
            bad += check(p)
    notes = "--notes" in sys.argv
    uniq = {}
    for b in bad:
        uniq.setdefault((b[0], b[2], b[3], b[5]), b)
    errors = 0
    for (path, name, l0, l1, st, kind, a, b_, in_asm) in uniq.values():
        if kind == "WAR" and not notes:        # informational: the hardware reads C in the MFMA's first passes
            continue
        errors += kind != "WAR"
        print(f"{path}:{l1}: {kind} {st} wait states after the MFMA at line {l0}{' (inline asm)' if in_asm else ''}\n    {a}\n    {b_}\n    in {name[:90]}")
    print(f"{errors} hazard(s), {sum(1 for k in uniq if k[3] == 'WAR')} C-operand note(s) in {len([a for a in sys.argv[1:] if not a.startswith('--')])} file(s)")
    sys.exit(1 if errors else 0)
