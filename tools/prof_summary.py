#!/usr/bin/env python3
"""Condense a rocprofv3 *_kernel_stats.csv (and optional PMC counter csv) into a short table
with kernel names cut to their first 70 characters.  Usage: prof_summary.py <dir> [out.md]"""
import csv
import glob
import os
import sys


def short(name, n=70):
    name = name.replace("void ", "")
    return name if len(name) <= n else name[:n] + "..."


def main():
    d = sys.argv[1]
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)):
        rows = list(csv.DictReader(open(f)))
        out.write(f"# {os.path.relpath(f, d)}\n\n| kernel | calls | avg us | min us | max us | % |\n|---|---|---|---|---|---|\n")
        for r in rows[:12]:
            out.write(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} | "
                      f"{float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} | {float(r['Percentage']):.1f} |\n")
        out.write("\n")
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        agg = {}
        for r in csv.DictReader(open(f)):
            k = (short(r["Kernel_Name"], 50), r["Counter_Name"])
            a = agg.setdefault(k, [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
        out.write(f"# {os.path.relpath(f, d)}\n\n| kernel | counter | dispatches | mean value |\n|---|---|---|---|\n")
        for (k, c), (n, s) in sorted(agg.items()):
            out.write(f"| `{k}` | {c} | {n} | {s/n:.1f} |\n")
        out.write("\n")


if __name__ == "__main__":
    main()
