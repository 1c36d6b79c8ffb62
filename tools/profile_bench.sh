#!/bin/bash
# Runs on the GPU box: (1) rocprofv3 --kernel-trace --stats of the bench command, (2) two separate
# PMC passes (FETCH_SIZE, WRITE_SIZE) as MI355X_MICROARCH.md prescribes, (3) the plain bench line.
# Usage: tools/profile_bench.sh <tag>
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_$tag
mkdir -p $out/trace $out/pmc_fetch $out/pmc_write
# kernel trace: the bench command itself (hipGraph of 240 steps, the kernels inside the graph are
# traced too); PMC passes: eager launches of the same step
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --profile-only > $out/trace/bench.json 2> $out/trace/err.log || exit 1
ARGS="bench.py --profile-only --no-graph --steps 120 --warmup 12"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $ARGS > $out/pmc_fetch/bench.json 2> $out/pmc_fetch/err.log || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $ARGS > $out/pmc_write/bench.json 2> $out/pmc_write/err.log || exit 1
python3 tools/prof_summary.py $out $out/summary.md
grep -E "rlvi::" $out/summary.md | cut -c1-170
python3 bench.py > $out/bench.json 2> $out/bench.err || exit 1
cat $out/bench.json
