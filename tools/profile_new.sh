#!/bin/bash
# rocprofv3 --kernel-trace --stats of the round's new kernels (tools/lab/run_new_kernels.py), the in-batch E+M leg and
# the threshold legs.  Usage: tools/profile_new.sh <tag>
tag=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_new_$tag
mkdir -p $out/new
rocprofv3 --kernel-trace --stats --output-format csv -d $out/new -- python3 tools/lab/run_new_kernels.py > $out/new/log.txt 2>&1 || exit 1
python3 tools/prof_summary.py $out $out/summary.md
grep -E "rlvi::" $out/summary.md | cut -c1-170
