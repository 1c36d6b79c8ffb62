#!/bin/bash
# Lab: is the M-step on short rows bound by rows or by bytes?  bf16 rows of 104 against fp32 rows of 52 (the same
# bytes), permuted against in-order sample indices.
cd $GRAFT_REPO_ROOT
for rows in 65536 131072 262144; do
  for spec in "104 bf16" "52 f32" "100 f32"; do
    set -- $spec
    python3 tools/time_parts.py --what mstep --rows $rows --classes $1 --dtype $2 2>&1 | grep "us/launch"
    python3 tools/time_parts.py --what mstep --rows $rows --classes $1 --dtype $2 --seq-idx --tag seq 2>&1 | grep "us/launch"
  done
done
