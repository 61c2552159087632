#!/usr/bin/env python
"""Summarise a per-launch record file of tools/host_resnet_bench.c (HOST_BENCH_RECORDS): launches grouped by (command, kernel, dims), sorted by time."""
import re, sys, collections
rows = []
for line in open(sys.argv[1]):
    m = re.match(r"\s*(\d+) (\S+?)\|(.*?)\s+dims (\d+) (\d+) (\d+) (\d+) (\d+)\s+([\d.]+) ms\s+([\d.]+) TFLOP/s\s+([\d.]+) GB/s", line)
    if m:
        rows.append((m.group(2), m.group(3)[-60:], tuple(int(m.group(i)) for i in range(4, 9)), float(m.group(9)), float(m.group(10)), float(m.group(11))))
agg = collections.OrderedDict()
for cmd, k, d, ms, tf, gb in rows:
    a = agg.setdefault((cmd, k, d), [0, 0.0, tf, gb]); a[0] += 1; a[1] += ms
print("total recorded ms %.3f in %d launches" % (sum(r[3] for r in rows), len(rows)))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 50
for (cmd, k, d), (n, ms, tf, gb) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%6.3f ms x%2d %6.1f TF %6.0f GB/s  %-22s %-28s %s" % (ms, n, tf, gb, cmd, str(d), k[-48:]))
