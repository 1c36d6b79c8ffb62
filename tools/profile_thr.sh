#!/bin/bash
# Runs on the GPU box: rocprofv3 --kernel-trace --stats of the threshold legs (tools/time_parts.py: hipGraph of
# 200 calls): the truncating call re-run on its own output, the criterion alone on an untruncated vector -- warm
# (the previous call's key as the guess) and without any guess.  Usage: tools/profile_thr.sh <tag>
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_thr_$tag
mkdir -p $out/trunc $out/crit $out/crit_cold
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trunc -- python3 tools/time_parts.py --what thr > $out/trunc/log.txt 2> $out/trunc/err.log || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/crit -- python3 tools/time_parts.py --what thr_fn > $out/crit/log.txt 2> $out/crit/err.log || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/crit_cold -- python3 tools/time_parts.py --what thr_fn --tune RLVI_THR_WARM=0 > $out/crit_cold/log.txt 2> $out/crit_cold/err.log || exit 1
for d in trunc crit crit_cold; do
  echo "== $d"; grep "us/launch" $out/$d/log.txt
  f=$(find $out/$d -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print("| kernel | calls | avg us | min us | max us | % |")
print("|---|---|---|---|---|---|")
for r in rows[:4]:
    print(f'| `{r["Name"][:110]}` | {r["Calls"]} | {float(r["AverageNs"])/1e3:.2f} | {float(r["MinNs"])/1e3:.2f} | {float(r["MaxNs"])/1e3:.2f} | {float(r["Percentage"]):.1f} |')
PY
done > $out/summary.md
cat $out/summary.md
