"""Per-kernel totals / medians of a rocprofv3 kernel trace csv: trace_stats.py <kernel_trace.csv> [top N]."""
import collections, csv, statistics, sys
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    d[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
top = int(sys.argv[2]) if len(sys.argv) > 2 else 16
tot = sum(sum(v) for v in d.values())
for name, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:top]:
    print(f"{sum(v) / 1e3:9.2f} ms {100 * sum(v) / tot:5.1f}%  n={len(v):6d}  median {statistics.median(v):8.2f}  min {min(v):8.2f}  max {max(v):9.2f} us  {name[:100]}")
