#!/bin/bash
# tools/asm_build.sh NAME EDITED.s BASE_VARIANT SOURCE.hip [hipcc flags of the base variant ...]
# An experiment library whose DEVICE code for one translation unit is the hand-edited listing EDITED.s (tools/asm_edit.py) and
# everything else is the base variant's (scratch/variants/_obj_<BASE_VARIANT>/*.o): assemble -> link the code object -> bundle ->
# compile the host side of SOURCE.hip around that bundle -> link scratch/variants/libraptor_quad_<NAME>.so.
set -e
cd "$(dirname "$0")/.."
name=$1; edited=$2; base=$3; src=$4; shift 4
LLVM=/opt/rocm/lib/llvm/bin
w=scratch/asmvar/$name; mkdir -p "$w"
$LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c "$edited" -o "$w/dev.o"
$LLVM/ld.lld -shared "$w/dev.o" -o "$w/dev.hsaco"
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 \
    -input=/dev/null -input="$w/dev.hsaco" -output="$w/dev.hipfb"
stem=$(basename "$src" .hip)
hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 --offload-arch=gfx950 -fvisibility=hidden -Wno-unused-function \
    -Iinclude "$@" -x hip --cuda-host-only -c "$src" -Xclang -fcuda-include-gpubinary -Xclang "$w/dev.hipfb" -o "$w/$stem.o"
objs=""
for o in scratch/variants/_obj_$base/*.o; do
    if [ "$(basename "$o")" = "$stem.o" ]; then objs="$objs $w/$stem.o"; else objs="$objs $o"; fi
done
hipcc -shared -fPIC --offload-arch=gfx950 -o "scratch/variants/libraptor_quad_$name.so" $objs -ldl
rm -f "$w/dev.o" "$w/dev.hsaco"
echo "built scratch/variants/libraptor_quad_$name.so"
