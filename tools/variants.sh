#!/bin/bash
# Experiment builds of libraptor_quad.so beside the product's (scratch/variants/libraptor_quad_<name>.so), for same-box A/B:
#   tools/variants.sh name "-DFLAG ..." [name2 "flags2" ...]      then on the GPU box:
#   RAPTOR_QUAD_LIB=scratch/variants/libraptor_quad_<name>.so python tools/launch_fit.py --precision bf16
set -e
cd "$(dirname "$0")/.."
while [ $# -ge 2 ]; do
  python -m raptor_amd.build --variant "$1" $2
  shift 2
done
