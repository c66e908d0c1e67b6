#!/usr/bin/env python3
"""Static check of a gfx950 listing (hipcc -S --cuda-device-only): is every register a vector-memory LOAD writes waited for
(`s_waitcnt vmcnt(N)`) on EVERY path before an instruction reads or overwrites it?

Round 5: the two-waves-per-SIMD bf16 rollout build differed from run to run under another instruction scheduler - always
lanes 48..63 (the last quarter of a load's data return), never with two waves per CU, more often the more waves a CU held:
the picture of a load whose data is used before it has arrived.  hipcc's wait-count pass is trusted for everything it emits;
this walks the control-flow graph of every kernel and re-derives the invariant from the listing.

Model (gfx9 family): loads and stores of the vector-memory path (global_* / scratch_* / buffer_* / flat_*) share ONE counter and
retire in issue order; `s_waitcnt vmcnt(N)` leaves at most the last N in flight; a call (s_swappc) is taken to return with none
in flight.  State = the ordered tuple of operations in flight; every (instruction, state) pair is visited once.

    python tools/waitcnt_lint.py file.s [more.s] [--kernel REGEX]         exit status 1 when something is reported
"""
import re
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from mfma_hazard_lint import functions          # noqa: E402

VMEM = ("global_load", "global_store", "global_atomic", "scratch_load", "scratch_store", "buffer_load", "buffer_store",
        "buffer_atomic", "flat_load", "flat_store", "flat_atomic")


def vmcnt_of(text):
    m = re.search(r"vmcnt\((\d+)\)", text)
    if m:
        return int(m.group(1))
    m = re.match(r"s_waitcnt\s+(0x[0-9a-f]+|\d+)\s*$", text)          # raw immediate: vmcnt = bits [3:0] | [15:14] << 4
    if m:
        v = int(m.group(1), 0)
        return (v & 0xF) | ((v >> 14) & 0x3) << 4
    return None


def check(path, kernel_re=None, max_states=400000):
    found = []
    for name, body, labels in functions(path):
        if kernel_re and not re.search(kernel_re, name):
            continue
        loads = {}                                                    # instruction index -> registers the load writes
        for i, x in enumerate(body):
            if x.op.startswith(VMEM) and ("_load" in x.op or ("atomic" in x.op and x.dst)):
                loads[i] = frozenset(x.dst)
        seen, work, reported, states = set(), [(0, ())], set(), 0
        while work:
            i, fly = work.pop()
            while i < len(body):
                if (i, fly) in seen:
                    break
                seen.add((i, fly))
                states += 1
                if states > max_states:
                    found.append((path, name, 0, 0, "state budget exhausted: kernel not fully checked", "", ""))
                    work = []
                    break
                x = body[i]
                if x.op == "s_waitcnt":
                    n = vmcnt_of(x.text)
                    if n is not None and len(fly) > n:
                        fly = fly[len(fly) - n:] if n else ()
                elif x.op == "s_swappc_b64":
                    fly = ()
                else:
                    touched = x.src | x.dst
                    for j in fly:
                        regs = loads.get(j)
                        if regs and regs & touched and (j, i) not in reported:
                            reported.add((j, i))
                            # a later LOAD into the same register is ordered behind the earlier one by the memory pipeline
                            # (loads retire in issue order): a note, not a finding
                            kind = ("note: reloads a register whose earlier load may still be in flight" if i in loads and not (regs & x.src)
                                    else "uses a register whose load may still be in flight")
                            found.append((path, name, body[j].line, x.line, kind, body[j].text, x.text))
                    if x.op.startswith(VMEM):
                        fly = fly + (i,)
                if x.ends:
                    break
                if x.target is not None and x.target in labels:
                    work.append((labels[x.target], fly))
                    if x.op == "s_branch":
                        break
                i += 1
    return found


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    kre = sys.argv[sys.argv.index("--kernel") + 1] if "--kernel" in sys.argv else None
    if kre in args:
        args.remove(kre)
    bad = []
    for p in args:
        bad += check(p, kre)
    notes = "--notes" in sys.argv
    errors = [b for b in bad if not b[4].startswith("note")]
    for path, name, l0, l1, what, a, b in (bad if notes else errors):
        print(f"{path}:{l1}: {what} (issued at line {l0})\n    {a}\n    {b}\n    in {name[:100]}")
    print(f"{len(errors)} finding(s), {len(bad) - len(errors)} load-after-load note(s) in {len(args)} file(s)")
    sys.exit(1 if errors else 0)
