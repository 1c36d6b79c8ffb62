#!/bin/bash
for rep in 1 2; do
for what in mstep mstep_warm; do
  for cfg in "RLVI_MSTEP_CUWIDE=0 --tune RLVI_MSTEP_HOLD=0" "RLVI_MSTEP_CUWIDE=0 --tune RLVI_MSTEP_HOLD=-1" "RLVI_MSTEP_CUWIDE=1 --tune RLVI_MSTEP_HOLD=0"; do
    echo -n "$what [$cfg]: "; python tools/time_parts.py --what $what --tune $cfg 2>&1 | tail -1 | cut -c30-120
  done
done; done
