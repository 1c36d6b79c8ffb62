#!/bin/bash
# Runs on the GPU box: HBM traffic of the in-batch E+M leg, one launch against the three-launch composition --
# FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (MI355X_MICROARCH.md), eager launches.
# Usage: tools/profile_fused_pmc.sh <tag>
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_fused_pmc_$tag
for mode in one three; do
  extra=""; [ $mode = three ] && extra="--tune RLVI_FUSED_EM=0"
  for c in FETCH_SIZE WRITE_SIZE; do
    d=$out/${mode}_$c; mkdir -p $d
    rocprofv3 --pmc $c --output-format csv -d $d -- python3 tools/time_parts.py --what fused --steps 20 $extra > $d/log.txt 2> $d/err.log || exit 1
  done
done
python3 - "$out" <<'PY' > $out/summary.md
import csv, glob, sys, collections
out = sys.argv[1]
print("| leg | kernel | counter | dispatches | mean per dispatch (KB) |")
print("|---|---|---|---|---|")
for mode in ("one", "three"):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob(f"{out}/{mode}_{c}/**/*counter_collection.csv", recursive=True)[0]
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "rlvi::" in r["Kernel_Name"] and r["Counter_Name"] == c:
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print(f"| {mode} | `{k[:70]}` | {c} | {len(v)} | {sum(v)/len(v):.1f} |")
PY
cat $out/summary.md
