#!/bin/bash
# Runs on the GPU box: rocprofv3 --kernel-trace --stats of the in-batch E+M leg (one launch per call,
# fused_em.hip) and of the same leg forced to the three-launch composition.  Usage: tools/profile_fused.sh <tag>
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_fused_$tag
mkdir -p $out/one $out/three
rocprofv3 --kernel-trace --stats --output-format csv -d $out/one -- python3 tools/time_parts.py --what fused > $out/one/log.txt 2> $out/one/err.log || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/three -- python3 tools/time_parts.py --what fused --tune RLVI_FUSED_EM=0 > $out/three/log.txt 2> $out/three/err.log || exit 1
for d in one three; do
  echo "== $d"; cat $out/$d/log.txt | grep "us/launch"
  f=$(find $out/$d -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print("| kernel | calls | avg us | min us | max us | % |")
print("|---|---|---|---|---|---|")
for r in rows[:6]:
    print(f'| `{r["Name"][:110]}` | {r["Calls"]} | {float(r["AverageNs"])/1e3:.2f} | {float(r["MinNs"])/1e3:.2f} | {float(r["MaxNs"])/1e3:.2f} | {float(r["Percentage"]):.1f} |')
PY
done > $out/summary.md
cat $out/summary.md
