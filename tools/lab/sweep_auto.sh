for pct in 100 130 150 170 200 -60 -80 -100 -120; do
  python tools/time_parts.py --what mstep --tag "mstep AUTO pct=$pct" --tune RLVI_MSTEP_AUTO=1 --tune RLVI_MSTEP_AUTO_PCT=$pct | grep -v amdgpu | tail -2
done
for pct in 150 -80 -100; do
  python tools/time_parts.py --what mstep_warm --tag "warm AUTO pct=$pct" --tune RLVI_MSTEP_AUTO=1 --tune RLVI_MSTEP_AUTO_PCT=$pct | grep -v amdgpu | tail -2
done
