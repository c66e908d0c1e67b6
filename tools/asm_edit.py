#!/usr/bin/env python3
"""Edit ONE kernel of a gfx950 assembly listing in place of the compiler, for experiments that must not change anything else
(round 5's hunt for the run-to-run differences of the two-waves-per-SIMD bf16 rollout build: a source-level `s_nop` is also a
scheduling barrier and moves everything around it; an inserted line in the listing is not).

    python tools/asm_edit.py IN.s OUT.s --kernel SUBSTRING [--range A:B] -e EDIT [-e EDIT ...]

EDIT is one of
    after:REGEX:TEXT      insert the line(s) TEXT (`;` separates instructions) after every instruction whose text matches REGEX
    before:REGEX:TEXT     the same, before it
    replace:REGEX:TEXT    replace the whole instruction by TEXT
--sites A:B restricts the edits to the A-th .. (B-1)-th instruction any edit matches (for bisecting);
--range A:B[,C:D ...] restricts the edits to instructions A <= index < B (or C <= index < D ...) of the kernel (indices count instructions, not lines; the
listing written with --list shows them).  tools/asm_build.sh turns OUT.s into a library.
"""
import argparse
import re
import sys

ap = argparse.ArgumentParser()
ap.add_argument("src")
ap.add_argument("dst")
ap.add_argument("--kernel", required=True)
ap.add_argument("--range", default=None)
ap.add_argument("--sites", default=None, help="A:B - only the edit sites (instructions an edit matches, in program order) A <= k < B")
ap.add_argument("--list", action="store_true", help="print the kernel's instructions with their indices and stop")
ap.add_argument("-e", "--edit", action="append", default=[], dest="edits")
args = ap.parse_args()

lines = open(args.src).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and args.kernel in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))


def is_instruction(l):
    s = l.strip()
    return l.startswith("\t") and s and not s.startswith((".", ";", "//"))


index = {}
n = 0
for i in range(start, end + 1):
    if is_instruction(lines[i]):
        index[i] = n
        n += 1
if args.list:
    for i in range(start, end + 1):
        print(f"{index[i]:5d} {lines[i]}" if i in index else f"      {lines[i]}")
    sys.exit(0)
ranges = [(0, n)]
if args.range:
    ranges = [(int(a or 0), int(b or n)) for a, b in (r.split(":") for r in args.range.split(","))]

edits = []
for e in args.edits:
    kind, rx, text = e.split(":", 2)
    edits.append((kind, re.compile(rx), ["\t" + t.strip() for t in text.split(";") if t.strip()]))
slo, shi = 0, 1 << 30
if args.sites:
    a, b = args.sites.split(":")
    slo, shi = int(a or 0), int(b or (1 << 30))
out = lines[:start]
count = 0
site = -1
for i in range(start, end + 1):
    l = lines[i]
    if i not in index or not any(lo <= index[i] < hi for lo, hi in ranges):
        out.append(l)
        continue
    body = l.strip().split(";")[0].strip()
    pre, post, repl = [], [], None
    if any(rx.search(body) for _, rx, _ in edits):
        site += 1
        if not (slo <= site < shi):
            out.append(l)
            continue
    for kind, rx, text in edits:
        if rx.search(body):
            count += 1
            if kind == "before":
                pre += text
            elif kind == "after":
                post += text
            elif kind == "replace":
                repl = text
            else:
                raise SystemExit("unknown edit " + kind)
    out += pre + (repl if repl is not None else [l]) + post
out += lines[end + 1:]
open(args.dst, "w").write("\n".join(out))
print(f"{args.dst}: kernel at line {start + 1}, {n} instructions, range(s) {ranges}, {count} edit sites", file=sys.stderr)
