"""HIP backend vs the committed reference outputs (tests/golden/nnc_golden.npz, produced by the reference's own CPU
backend): every seeded case of tests/golden_cases.py through the C-ABI (nnc_mi355x_cmd_exec) on GPU tensors.
`emu` flavour = the same kernel sources on the CPU HIP emulator (CPU tier); `gpu` flavour = the MI355X.
Tolerances: pooling / relu / scalar bit-exact; conv / GEMM 1e-4 relative (north_star); softmax / SGD a few ulp."""
import os
import numpy as np
import pytest
from ccv_amd import nnc
from golden_cases import CASES, build_case, run_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXACT = ("maxpool", "relu", "scalar_mul")


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "nnc_golden.npz"))


@pytest.mark.parametrize("name", sorted(CASES))
def test_backend_matches_reference_golden(backend, golden, name):
    case = build_case(name)
    if case["fmt"] == "NCHW" and not backend.cmd_ok(nnc.CMD["CONVOLUTION_FORWARD"], nnc.BACKEND_GPU_CUDNN):
        pytest.skip("no conv")
    got = run_case(backend, nnc.GPU_MEMORY, case)
    for i, g in enumerate(got):
        key = "%s/out%d" % (name, i)
        if key not in golden:
            continue
        w = golden[key]
        if name.startswith(EXACT):
            np.testing.assert_array_equal(g, w, err_msg=key)
        elif name.startswith(("conv", "gemm")):
            np.testing.assert_allclose(g, w, rtol=1e-4, atol=2e-5, err_msg=key)
        else:
            np.testing.assert_allclose(g, w, rtol=1e-5, atol=1e-6, err_msg=key)
