#!/bin/bash
# Build libraptor_quad.so of another git revision next to the working tree's, for same-box A/B timing:
#   tools/ab_build.sh HEAD~1            -> scratch/ab/HEAD~1.so
#   RAPTOR_QUAD_LIB=scratch/ab/HEAD~1.so python bench.py      (raptor_amd/_lib.py honours the variable)
set -e
rev=${1:?revision}
cd "$(dirname "$0")/.."
out=scratch/ab/$(echo "$rev" | tr '/~^' '___')
rm -rf "$out"; mkdir -p "$out/raptor_amd" "$out/include"
git archive "$rev" raptor_amd/csrc raptor_amd/build.py raptor_amd/gfx950_errata.py include | tar -x -C "$out"
mkdir -p "$out/tools"; git show "$rev:tools/codeobj_check.py" > "$out/tools/codeobj_check.py" 2>/dev/null || rm -f "$out/tools/codeobj_check.py"      # the post-link gate (round 6 on)
touch "$out/raptor_amd/__init__.py"
(cd "$out" && python -c "
import sys; sys.path.insert(0, '.')
import importlib.util
spec = importlib.util.spec_from_file_location('b', 'raptor_amd/build.py'); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
print(b.build(force=True))")
cp "$out/raptor_amd/libraptor_quad.so" "$out.so"
echo "$out.so"
