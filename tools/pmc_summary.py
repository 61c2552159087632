#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc csv output (one directory per counter group) into per-kernel sums / per-dispatch means."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "")
    depth = 0
    for i, ch in enumerate(n):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return n[:i]
    return n


def main(root):
    for group in sorted(os.listdir(root)):
        d = os.path.join(root, group)
        if not os.path.isdir(d):
            continue
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            print("## %s: no counter_collection csv (see %s.log)\n" % (group, group))
            continue
        acc = defaultdict(lambda: defaultdict(float))
        disp = defaultdict(set)
        for f in files:
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    k = short(row["Kernel_Name"])
                    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
                    disp[k].add(row["Dispatch_Id"])
        counters = sorted({c for v in acc.values() for c in v})
        print("## %s (sum over dispatches)\n" % group)
        print("| kernel | dispatches | " + " | ".join(counters) + " |")
        print("|---|---|" + "---|" * len(counters))
        for k in sorted(acc, key=lambda k: -max(acc[k].values())):
            print("| `%s` | %d | " % (k[:150], len(disp[k])) + " | ".join("%.4g" % acc[k].get(c, 0) for c in counters) + " |")
        print()


if __name__ == "__main__":
    main(sys.argv[1])
