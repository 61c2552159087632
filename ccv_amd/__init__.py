"""ccv_amd: host-side mirror of ccv's nnc command interface over libnnc_mi355x.so (MI355X / gfx950).

The product is the C-ABI shared library built from ccv_amd/csrc (see include/nnc_mi355x.h).  This package is
the thin Python binding used by tests/, bench.py and __graft_entry__.py: ctypes mirrors of the plugin-surface
structs, the CMD_* builders of lib/nnc/cmd/ccv_nnc_cmd_easy.h, and tensor helpers.  There is no CPU fallback:
`load()` raises if the HIP library is missing or no GPU is visible.
"""
from .nnc import *  # noqa: F401,F403
