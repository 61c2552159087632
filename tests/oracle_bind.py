"""Pick the checker: the reference's own CPU backend when its build is present (oracle/_ref/libccv_ref.so), otherwise the
plain-C restatement (oracle/libnnc_oracle.so).  TEST INFRASTRUCTURE: imported only by tests/, smoke() and bench.py's
cpu_baseline leg."""
import ctypes as C
import os
from ccv_amd import nnc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class CheckerLib(nnc.Lib):
    """A CHECKER behind the command interface of ccv_amd.nnc.Lib (same Cmd / Hint / tensor structs, so the drivers in tests/ and ccv_amd/vgg.py run on it
    unchanged): kind "reference" = oracle/_ref/libccv_ref.so, the reference's own lib/nnc compiled here (ccv_nnc_cmd_exec, CPU backends); kind "oracle" =
    oracle/libnnc_oracle.so, the plain-C restatement (nnc_oracle_cmd_exec; CPU tensors only, no device runtime).  Lives in tests/: the product package
    cannot load either."""

    def __init__(self, path, kind):
        assert kind in ("reference", "oracle")
        self.path, self.kind = path, kind
        self.dll = C.CDLL(path, mode=C.RTLD_GLOBAL if kind == "reference" else C.RTLD_LOCAL)
        if kind == "reference":
            self.dll.ccv_nnc_init()
            self._exec = self.dll.ccv_nnc_cmd_exec
        else:
            self._exec = self.dll.nnc_oracle_cmd_exec
        self._exec.restype = C.c_int
        self._exec.argtypes = nnc.Lib._EXEC_ARGS


def oracle_lib(prefer="reference"):
    """Returns (lib, backend id to force, pools-must-be-issued-per-image)."""
    ref = os.path.join(ROOT, "oracle", "_ref", "libccv_ref.so")
    port = os.path.join(ROOT, "oracle", "libnnc_oracle.so")
    if prefer == "reference" and os.path.exists(ref):
        return CheckerLib(ref, "reference"), nnc.BACKEND_CPU_REF, True
    if os.path.exists(port):
        return CheckerLib(port, "oracle"), None, False
    if os.path.exists(ref):
        return CheckerLib(ref, "reference"), nnc.BACKEND_CPU_REF, True
    raise RuntimeError("no oracle built (run __graft_entry__.build())")
