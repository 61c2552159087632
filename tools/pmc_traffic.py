#!/usr/bin/env python
"""HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_pass.sh (rocprofv3 --pmc, one counter per pass).
Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: both counters are in KiB;
FETCH_SIZE tallies the 128-byte requests of 16-byte-per-lane reads at 64 bytes, so reads are DOUBLED; WRITE_SIZE is
uncalibrated and taken as is.  Where the size-resolved request passes exist (rdsize / wrsize: TCC_EA0_RDREQ{,_32B,_64B,_128B}_sum, TCC_EA0_WRREQ{,_64B}_sum),
the exact byte counts they give are reported next to those (read_bytes_by_request_size_per_launch, write_...): on known volumes (tools/fetch_calib.cpp,
profiles/r05_v7_counter_calibration.txt) they are exact, and FETCH_SIZE x 2 agrees with them because every read request of these kernels is a 128-byte one.
All of these count requests of the XCDs' L2s to the fabric: a line another XCD fetched a microsecond earlier, or one that fell out of the 4 MB L2 between two uses,
is counted again although the 256 MB Infinity Cache serves it -- "traffic" is an upper bound of the HBM bytes.  usage: pmc_traffic.py gpurun_out/pmc out.json [batch]"""
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
    sized = {c: per_kernel(root, g, c) for g, c in (("rdsize", "TCC_EA0_RDREQ_sum"), ("rdsize", "TCC_EA0_RDREQ_32B_sum"), ("rdsize", "TCC_EA0_RDREQ_64B_sum"), ("rdsize", "TCC_EA0_RDREQ_128B_sum"),
                                                      ("wrsize", "TCC_EA0_WRREQ_sum"), ("wrsize", "TCC_EA0_WRREQ_64B_sum"))}
    for k in sorted(set(fetch) | set(write)):
        fb, fn = fetch.get(k, (0.0, 0))
        wb, wn = write.get(k, (0.0, 0))
        launches = max(fn, wn, 1)
        res["kernels"][k] = {"launches": launches, "read_bytes_per_launch": 2.0 * 1024.0 * fb / launches, "write_bytes_per_launch": 1024.0 * wb / launches}
        if k in sized["TCC_EA0_RDREQ_sum"]:
            (r, n), r32, r64, r128 = sized["TCC_EA0_RDREQ_sum"][k], sized["TCC_EA0_RDREQ_32B_sum"].get(k, (0.0, 0))[0], sized["TCC_EA0_RDREQ_64B_sum"].get(k, (0.0, 0))[0], sized["TCC_EA0_RDREQ_128B_sum"].get(k, (0.0, 0))[0]
            res["kernels"][k]["read_bytes_by_request_size_per_launch"] = (32.0 * r32 + 64.0 * r64 + 128.0 * r128 + 64.0 * max(0.0, r - r32 - r64 - r128)) / max(n, 1)
        if k in sized["TCC_EA0_WRREQ_sum"]:
            (w, n), w64 = sized["TCC_EA0_WRREQ_sum"][k], sized["TCC_EA0_WRREQ_64B_sum"].get(k, (0.0, 0))[0]
            res["kernels"][k]["write_bytes_by_request_size_per_launch"] = (64.0 * w64 + 32.0 * max(0.0, w - w64)) / max(n, 1)
    with open(out, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    for k, v in sorted(res["kernels"].items(), key=lambda kv: -(kv[1]["read_bytes_per_launch"] + kv[1]["write_bytes_per_launch"]) * kv[1]["launches"])[:24]:
        print("%8.1f MB read %8.1f MB written per launch (by request size: %s read, %s written) x %3d  %s" % (v["read_bytes_per_launch"] / 1e6, v["write_bytes_per_launch"] / 1e6,
              "%.1f" % (v["read_bytes_by_request_size_per_launch"] / 1e6) if "read_bytes_by_request_size_per_launch" in v else "-",
              "%.1f" % (v["write_bytes_by_request_size_per_launch"] / 1e6) if "write_bytes_by_request_size_per_launch" in v else "-", v["launches"], k[:110]))


if __name__ == "__main__":
    main()
