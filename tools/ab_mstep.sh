for rep in 1 2; do
for v in gfx950 rmwtail hold300 hold400 hold500 hold600; do
  echo -n "$v: "; RLVI_LIB_PATH=rlvi_amd/librlvi_$v.so python tools/time_parts.py --what mstep 2>&1 | tail -1
done; done
