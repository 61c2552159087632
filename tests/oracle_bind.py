"""Pick the checker: the reference's own CPU backend when its build is present (oracle/_ref/libccv_ref.so), otherwise the
plain-C restatement (oracle/libnnc_oracle.so).  TEST INFRASTRUCTURE: imported only by tests/, smoke() and bench.py's
cpu_baseline leg."""
import os
from ccv_amd import nnc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def oracle_lib(prefer="reference"):
    """Returns (lib, backend id to force, pools-must-be-issued-per-image)."""
    ref = os.path.join(ROOT, "oracle", "_ref", "libccv_ref.so")
    port = os.path.join(ROOT, "oracle", "libnnc_oracle.so")
    if prefer == "reference" and os.path.exists(ref):
        return nnc.Lib(ref, "reference"), nnc.BACKEND_CPU_REF, True
    if os.path.exists(port):
        return nnc.Lib(port, "oracle"), None, False
    if os.path.exists(ref):
        return nnc.Lib(ref, "reference"), nnc.BACKEND_CPU_REF, True
    raise RuntimeError("no oracle built (run __graft_entry__.build())")
