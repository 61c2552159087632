#!/usr/bin/env python
"""SURVEY section 8(f).3, measured first: is a step launch-bound?  From a rocprofv3 --kernel-trace result db: the fraction of wall time, over the last
`window_ms` of the trace (the timed steps; the compile / autotune phase lies before), during which at least one kernel was running on the GPU, and
the longest gaps.   usage: python tools/gpu_busy.py <results.db> [window_ms]"""
import sqlite3
import sys


def main(path, window_ms=200.0):
    cur = sqlite3.connect(path).cursor()
    rows = sorted((s, e) for s, e in cur.execute("select start, end from kernels"))
    if not rows:
        print("no kernels")
        return
    t1 = max(e for _, e in rows)
    t0 = t1 - int(window_ms * 1e6)
    busy, gaps, cur_s, cur_e, n = 0, [], None, None, 0
    for s, e in rows:
        if e <= t0:
            continue
        s = max(s, t0)
        n += 1
        if cur_e is None:
            cur_s, cur_e = s, e
        elif s <= cur_e:
            cur_e = max(cur_e, e)
        else:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
            cur_s, cur_e = s, e
    busy += cur_e - cur_s
    span = t1 - max(t0, min(s for s, _ in rows))
    gaps.sort(reverse=True)
    print("window %.1f ms (the end of the trace): %d kernels, GPU busy %.2f ms = %.1f %% of the window; idle %.2f ms in %d gaps (largest: %s us; median %.1f us)" % (
        span / 1e6, n, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6, len(gaps), ", ".join("%.0f" % (g / 1e3) for g in gaps[:5]), (gaps[len(gaps) // 2] / 1e3) if gaps else 0.0))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 200.0)
