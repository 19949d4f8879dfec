#!/usr/bin/env python3
"""Per-kernel totals from a rocprofv3 --kernel-trace output directory: python3 tools/kstats.py <dir> [top]"""
import collections, csv, glob, sys
per = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        per[r["Kernel_Name"].split("(")[0][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
tot = sum(sum(v) for v in per.values())
top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
print("%-60s %6s %11s %9s %9s %7s" % ("kernel", "calls", "total_ms", "avg_ms", "max_ms", "share"))
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:top]:
    print("%-60s %6d %11.3f %9.4f %9.4f %6.1f%%" % (k, len(v), sum(v), sum(v) / len(v), max(v), 100 * sum(v) / tot))
