#!/usr/bin/env python
"""Generate tests/golden/nnc_golden.npz from the reference's OWN CPU backend (oracle/_ref/libccv_ref.so, built from
/root/reference by oracle/build_ref.sh).  Run in the build container; the .npz is committed so that the GPU box (which
has no /root/reference) and the plain-C restatement can be checked against reference outputs.

Every case stores: command name + constructor args, hint, flags, input arrays, reference output arrays.  Pools are issued
per image (the reference CPU pool kernels only process image 0 of a batch)."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ccv_amd import nnc  # noqa: E402
from golden_cases import CASES, build_case, run_case  # noqa: E402


def main():
    from oracle_bind import CheckerLib
    R = CheckerLib(os.path.join(ROOT, "oracle", "_ref", "libccv_ref.so"), "reference")
    out = {}
    for name in CASES:
        case = build_case(name)
        res = run_case(R, nnc.CPU_MEMORY, case, backend=nnc.BACKEND_CPU_REF, per_image_pool=True)
        for i, r in enumerate(res):
            if r is not None:
                out["%s/out%d" % (name, i)] = r
    path = os.path.join(ROOT, "tests", "golden", "nnc_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote %s: %d cases, %.1f KB" % (path, len(CASES), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
