#!/bin/bash
for rep in 1 2; do
for h in 0 -1; do
  echo -n "hold=$h mstep: "; python tools/time_parts.py --what mstep --tune RLVI_MSTEP_HOLD=$h 2>&1 | tail -1
  echo -n "hold=$h step : "; python tools/time_parts.py --what step --tune RLVI_MSTEP_HOLD=$h 2>&1 | tail -1
done; done
echo -n "4x: "; python tools/time_parts.py --what mstep --rows 262144 --steps 60 2>&1 | tail -1
