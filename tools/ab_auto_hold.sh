#!/bin/bash
# Same box, same run: the M-step's default 16-wave barrier form (RLVI_MSTEP_AUTO=0) against the self-timed hold
# (RLVI_MSTEP_AUTO=1, at 80 / 100 / 120 % of the measured issue time), HBM-cold rotation and single buffer pair
# (Infinity-Cache-warm), and the whole step.
for what in mstep mstep_warm step; do
  python tools/time_parts.py --what $what --tag "$what AUTO=0" --tune RLVI_MSTEP_AUTO=0 | tail -1
  for pct in 80 100 120; do
    python tools/time_parts.py --what $what --tag "$what AUTO=1 pct=$pct" --tune RLVI_MSTEP_AUTO=1 --tune RLVI_MSTEP_AUTO_PCT=$pct | tail -1
  done
done
python tools/time_parts.py --what mstep --rows 49152 --tag "49152 AUTO=0" --tune RLVI_MSTEP_AUTO=0 | tail -1
python tools/time_parts.py --what mstep --rows 49152 --tag "49152 AUTO=1" --tune RLVI_MSTEP_AUTO=1 | tail -1
python tools/time_parts.py --what mstep --classes 128 --tag "x128 AUTO=0" --tune RLVI_MSTEP_AUTO=0 | tail -1
python tools/time_parts.py --what mstep --classes 128 --tag "x128 AUTO=1" --tune RLVI_MSTEP_AUTO=1 | tail -1
python tools/time_parts.py --what mstep --classes 104 --dtype bf16 --tag "bf16 x104 AUTO=0" --tune RLVI_MSTEP_AUTO=0 | tail -1
python tools/time_parts.py --what mstep --classes 104 --dtype bf16 --tag "bf16 x104 AUTO=1" --tune RLVI_MSTEP_AUTO=1 | tail -1
