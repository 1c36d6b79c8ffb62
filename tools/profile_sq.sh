#!/bin/bash
# Runs on the GPU box: one PMC pass of SQ counters over the bench step (eager launches): where the waves of the two
# dominant kernels spend their cycles, and the LDS bank conflicts.  Usage: tools/profile_sq.sh <tag>
tag=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_sq_$tag
mkdir -p $out
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU --output-format csv -d $out -- python3 bench.py --profile-only --no-graph --steps 120 --warmup 12 > $out/bench.json 2> $out/err.log || { tail -5 $out/err.log; exit 1; }
python3 - "$out" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES":
        cnt[k] += 1
print("| kernel | launches | waves | wave cycles (quad) | parked (s_waitcnt / barrier) | issue stall | issuing | VALU insts per wave | LDS conflict / LDS active |")
print("|---|---|---|---|---|---|---|---|---|")
for k, c in acc.items():
    n = max(cnt[k], 1)
    if "rlvi::" not in k:
        continue
    wc = c["SQ_WAVE_CYCLES"] or 1.0
    print(f'| `{k}` | {n} | {c["SQ_WAVES"]/n:.0f} | {wc/n:.3g} | {100*c["SQ_WAIT_ANY"]/wc:.1f} % | {100*c["SQ_WAIT_INST_ANY"]/wc:.1f} % | {100*c["SQ_ACTIVE_INST_ANY"]/wc:.1f} % | {c["SQ_INSTS_VALU"]/max(c["SQ_WAVES"],1):.0f} | {c["SQ_LDS_BANK_CONFLICT"]/n:.3g} / {c["SQ_LDS_IDX_ACTIVE"]/n:.3g} |')
PY
