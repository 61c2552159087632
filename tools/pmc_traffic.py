#!/usr/bin/env python
"""HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_pass.sh (rocprofv3 --pmc, one counter per pass).
Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: both counters are in KiB;
FETCH_SIZE tallies the 128-byte requests of 16-byte-per-lane reads at 64 bytes, so reads are DOUBLED; WRITE_SIZE is
uncalibrated and taken as is.  usage: pmc_traffic.py gpurun_out/pmc out.json [batch]"""
import csv, glob, json, os, sys
from collections import defaultdict
from pmc_summary import short


def per_kernel(root, group, counter):
    acc, n = defaultdict(float), defaultdict(set)
    for f in glob.glob(os.path.join(root, group, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter: continue
                k = short(row["Kernel_Name"])
                acc[k] += float(row["Counter_Value"])
                n[k].add(row["Dispatch_Id"])
    return {k: (acc[k], len(n[k])) for k in acc}


def main():
    root, out = sys.argv[1], sys.argv[2]
    fetch, write = per_kernel(root, "fetch", "FETCH_SIZE"), per_kernel(root, "write", "WRITE_SIZE")
    res = {"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over `bench.py --steps 1 --warmup 0 --batch %s`; "
                     "KiB -> bytes, reads x2 (gfx950 128-byte requests tallied at 64 bytes), writes uncorrected" % (sys.argv[3] if len(sys.argv) > 3 else "?"),
           "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        fb, fn = fetch.get(k, (0.0, 0))
        wb, wn = write.get(k, (0.0, 0))
        launches = max(fn, wn, 1)
        res["kernels"][k] = {"launches": launches, "read_bytes_per_launch": 2.0 * 1024.0 * fb / launches, "write_bytes_per_launch": 1024.0 * wb / launches}
    with open(out, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    for k, v in sorted(res["kernels"].items(), key=lambda kv: -(kv[1]["read_bytes_per_launch"] + kv[1]["write_bytes_per_launch"]) * kv[1]["launches"])[:24]:
        print("%8.1f MB read %8.1f MB written per launch x %3d  %s" % (v["read_bytes_per_launch"] / 1e6, v["write_bytes_per_launch"] / 1e6, v["launches"], k[:130]))


if __name__ == "__main__":
    main()
