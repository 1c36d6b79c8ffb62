#!/bin/bash
# M-step at the small shapes of BASELINE.json's configs: the launcher's choice (RLVI_MSTEP_G=0, RLVI_MSTEP_FORM=-1)
# against forced lanes per row / forms.  usage: tools/sweep_small.sh [out-file]
set -e
o=${1:-gpurun_out/sweep_small.txt}
: > $o
for shape in "32 10 f32" "128 10 f32" "1024 10 f32" "4096 10 f32" "16384 10 f32" "65536 10 f32" \
             "1024 100 f32" "4096 100 f32" "8192 100 f32" "16384 100 f32" "32768 100 f32" \
             "1024 101 bf16" "8192 101 bf16" "65536 101 bf16" "1024 101 f32" "16384 101 f32" "32768 101 f32" \
             "1024 104 bf16" "4096 104 bf16" "16384 104 bf16"; do
  set -- $shape
  python tools/time_parts.py --what mstep --rows $1 --classes $2 --dtype $3 >> $o 2>&1
done
grep -v "reps\|amdgpu.ids" $o
