#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/lab4; mkdir -p $out/thr
rocprofv3 --kernel-trace --stats --output-format csv -d $out/thr -- python3 tools/time_parts.py --what thr --n 65536 --steps 100 > $out/thr/out.log 2> $out/thr/err.log
python3 tools/prof_summary.py $out/thr $out/thr/summary.md; grep -E "rlvi::" $out/thr/summary.md | cut -c1-150
for v in gfx950 plainst; do
  RLVI_LIB_PATH=$PWD/rlvi_amd/librlvi_$v.so timeout -k 10 200 python tools/time_parts.py --what step --tag step_$v 2>&1 | grep -v "amdgpu.ids\|reps"
  RLVI_LIB_PATH=$PWD/rlvi_amd/librlvi_$v.so timeout -k 10 200 python tools/time_parts.py --what mstep --tag mstep_$v 2>&1 | grep -v "amdgpu.ids\|reps"
done
