#!/usr/bin/env python
"""Summarise a rocprofv3 (--kernel-trace --stats) sqlite result db: per-kernel totals, and per-dispatch rows."""
import sqlite3
import sys


def _short(name):
    """Kernel symbol without its argument list (durations in the db: top_kernels in us, kernels.duration in ns)."""
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


def main(path, per_dispatch=False):
    db = sqlite3.connect(path)
    cur = db.cursor()
    print("| kernel | calls | total ms | avg us | % |")
    print("|---|---|---|---|---|")
    for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        short = _short(name)
        print("| `%s` | %d | %.3f | %.1f | %.2f |" % (short, calls, total / 1e3, avg, pct))
    if per_dispatch:
        print()
        for name, dur, gx, gy, gz, wx, vg, ag, lds in cur.execute("select name,duration,grid_x,grid_y,grid_z,workgroup_x,vgpr_count,accum_vgpr_count,lds_size from kernels order by start"):
            print("%-110s %10.1f us grid=(%d,%d,%d)/%d vgpr=%d agpr=%d lds=%d" % (_short(name), dur / 1e3, gx // max(wx, 1), gy, gz, wx, vg, ag, lds))


if __name__ == "__main__":
    main(sys.argv[1], len(sys.argv) > 2)
