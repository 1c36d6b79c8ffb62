#!/bin/bash
# Lab: variants of the M-step's gather / scatter (tools/build_variants.py name=-DFLAG ...), same box, two passes.
# Usage: tools/lab/gs_variants.sh gfx950 ep ...
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "$@"; do
  echo "== $v"
  RLVI_LIB_PATH=$GRAFT_REPO_ROOT/rlvi_amd/librlvi_$v.so python3 tools/lab/gather_scatter_cost.py 2>&1 | grep "us/launch" | sed -n '1p;2p;4p;5p'
done; done
