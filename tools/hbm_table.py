#!/usr/bin/env python
"""Achieved HBM bandwidth per kernel: bytes per launch from the PMC passes (tools/pmc_traffic.py json) over the average
duration from the rocprofv3 kernel trace of the same command (tools/prof_summary.py on the results db).
usage: hbm_table.py <pmc_traffic.json> <rocprof results.db> > profiles/..._hbm_kernels.md"""
import json, sqlite3, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from pmc_summary import short


def durations(db):
    """{short kernel name: [calls, total ns]} from rocprofv3's `top_kernels` view (total_duration in us, like prof_summary.py)"""
    con = sqlite3.connect(db)
    out = {}
    for name, calls, total in con.execute("select name,total_calls,total_duration from top_kernels"):
        a = out.setdefault(short(name), [0, 0.0])
        a[0] += calls
        a[1] += total
    return out


def main():
    tr = json.load(open(sys.argv[1]))["kernels"]
    du = durations(sys.argv[2])
    print("| kernel | launches (trace) | avg us | read MB/launch | written MB/launch | TB/s |")
    print("|---|---|---|---|---|---|")
    rows = []
    for k, v in tr.items():
        if k not in du or du[k][0] == 0:
            continue
        avg_us = du[k][1] / du[k][0]   # top_kernels.total_duration is in microseconds
        b = v["read_bytes_per_launch"] + v["write_bytes_per_launch"]
        rows.append((du[k][1], k, du[k][0], avg_us, v["read_bytes_per_launch"] / 1e6, v["write_bytes_per_launch"] / 1e6, b / (avg_us * 1e-6) / 1e12))
    for _, k, n, us, r, w, tbs in sorted(rows, reverse=True)[:28]:
        print("| `%s` | %d | %.1f | %.1f | %.1f | %.2f |" % (k[:120], n, us, r, w, tbs))


if __name__ == "__main__":
    main()
