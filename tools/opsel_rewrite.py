#!/usr/bin/env python3
"""Rewrite every packed-fp32 instruction of a gfx950 listing that can meet the op_sel fault (tools/opsel_lint.py,
profiles/r05_bf16_two_wave_hunt.md) into its sound twin: v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 whose low result takes
src0.low and src1.HIGH (op_sel:[0,1,..]) get src0 and src1 exchanged, with every per-source modifier (op_sel, op_sel_hi, neg_lo,
neg_hi) exchanged along - the same sum / product of the same values (fp add and multiply commute), now selecting the high half of
src0, which is not affected.

    python tools/opsel_rewrite.py IN.s OUT.s      prints how many instructions were rewritten; tools/asm_build.sh builds OUT.s
"""
import re
import sys

INS = re.compile(r"^(\s*)(v_pk_(?:add|mul|fma)_f32)(?:_e64)?\s+(.*?)\s*(;.*)?$")
MOD = re.compile(r"\b(op_sel|op_sel_hi|neg_lo|neg_hi):\[([01,]+)\]")
DEFAULT = {"op_sel": 0, "op_sel_hi": 1, "neg_lo": 0, "neg_hi": 0}


def rewrite(line):
    m = INS.match(line)
    if not m:
        return line, False
    indent, op, rest, comment = m.group(1), m.group(2), m.group(3), m.group(4) or ""
    mods = {k: [int(x) for x in v.split(",")] for k, v in MOD.findall(rest)}
    head = MOD.sub("", rest).strip()
    extra = ""
    if head.endswith("clamp"):
        head, extra = head[:-5].strip(), " clamp"
    ops = [o.strip() for o in head.rstrip(",").split(",")]
    nsrc = 3 if op.endswith("fma_f32") else 2
    if len(ops) != 1 + nsrc:
        return line, False
    sel = mods.get("op_sel", [0] * nsrc)
    if not (sel[0] == 0 and sel[1] == 1):
        return line, False
    if not (ops[1].startswith("v[") and ops[2].startswith("v[")):        # a constant or scalar pair has no high half to select: left alone, reported
        print("left alone:", line.strip(), file=sys.stderr)
        return line, False
    ops[1], ops[2] = ops[2], ops[1]
    out = []
    for k in ("op_sel", "op_sel_hi", "neg_lo", "neg_hi"):
        v = mods.get(k, [DEFAULT[k]] * nsrc)
        v[0], v[1] = v[1], v[0]
        if any(x != DEFAULT[k] for x in v):
            out.append(f"{k}:[{','.join(str(x) for x in v)}]")
    text = f"{indent}{op} {', '.join(ops)}{(' ' + ' '.join(out)) if out else ''}{extra}"
    return text + ((" " + comment) if comment else ""), True


def main():
    src, dst = sys.argv[1], sys.argv[2]
    n = 0
    with open(dst, "w") as f:
        for line in open(src):
            new, changed = rewrite(line.rstrip("\n"))
            n += changed
            f.write(new + "\n")
    print(f"{dst}: {n} instructions rewritten", file=sys.stderr)


if __name__ == "__main__":
    main()
