#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_membench; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- tools/membench > $out/stdout.log 2>$out/err.log
python3 tools/prof_summary.py $out $out/summary.md; grep -E "k_flat|k_rows|copyBuffer" $out/summary.md | cut -c1-140
