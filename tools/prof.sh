#!/bin/bash
# usage: tools/prof.sh <outdir-name> <python args...>   -- rocprofv3 kernel-trace + stats of one command
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; shift
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 "$@" > $out/stdout.log 2> $out/stderr.log
python3 tools/prof_summary.py $out $out/summary.md
head -8 $out/summary.md | cut -c1-150
