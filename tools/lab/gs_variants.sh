#!/bin/bash
# Lab: the M-step's gather / scatter variants (tools/build_variants.py sg=... s1=... ), same box, two passes
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in gfx950 sg s1 s2 s3 sgs1 sgs2; do
  echo "== $v"
  RLVI_LIB_PATH=$GRAFT_REPO_ROOT/rlvi_amd/librlvi_$v.so python3 tools/lab/gather_scatter_cost.py 2>&1 | grep "us/launch" | sed -n '1p;2p;5p'
done; done
