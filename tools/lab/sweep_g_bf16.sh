#!/bin/bash
# Lab: lanes per row (RLVI_MSTEP_G; 0 = the launcher's choice) for bf16 rows of 200 ... 2000 elements
cd $GRAFT_REPO_ROOT
for rows in 4096 16384; do for C in 200 256 512 768 1000 2000; do for g in 0 8 16 32 64; do
  python3 tools/time_parts.py --what mstep --rows $rows --classes $C --dtype bf16 --tune RLVI_MSTEP_G=$g --tag "G=$g" 2>&1 | grep "us/launch" | cut -c1-95
done; done; done
