"""The reference's OWN GPU integration tests (test/int/nnc/*.tests.c), compiled from where they lie against the
reference's OWN unmodified host linked to libnnc_mi355x.so (oracle/build_ref_host.sh), run one case per process.
tests/golden/ref_int_expected_pass.txt is the set of cases this backend claims; every one must print PASS.
  gpu tier: all of them on the MI355X (oracle/_ref/int/*.gpu)
  CPU tier: the quick ones on the CPU HIP emulator build of the same kernels (oracle/_ref/int/*.emu)"""
import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_int_tests as R  # noqa: E402


def _expected():
    rows = []
    for line in open(os.path.join(ROOT, "tests", "golden", "ref_int_expected_pass.txt")):
        if line.strip():
            s, n = line.rstrip("\n").split("\t", 1)
            rows.append((s, n))
    return rows


EXPECTED = _expected()
# convolutions at the reference tests' sizes (64 x 224 x 224 images) take minutes on the emulator, and so do the twelve attention
# trials of the flash_attn gradient case (B = 32, R = 160 ...; tests/test_attention.py covers the row on the emulator at small sizes)
QUICK = [(s, n) for s, n in EXPECTED if "convolution" not in n and "scaled dot product attention" not in n]


_VERDICTS = {}


def _prefetch(flavor, rows, timeout, env, key):
    """Run a whole list of cases (one process each) a few at a time and keep the verdicts: the cases are independent processes
    dominated by start-up cost, so the tier's wall time drops several-fold; every parametrised test then looks its verdict up."""
    if key in _VERDICTS:
        return _VERDICTS[key]
    from concurrent.futures import ThreadPoolExecutor
    out = {}

    def one(row):
        suite, name = row
        b = os.path.join(R.BIN, "%s.%s" % (suite, flavor))
        if not os.path.exists(b):
            return row, ("MISSING", "")
        return row, R.run_case(b, name, timeout, env)
    with ThreadPoolExecutor(max_workers=int(os.environ.get("NNC_REF_INT_JOBS", "6"))) as ex:
        for row, verdict in ex.map(one, rows):
            out[row] = verdict
    _VERDICTS[key] = out
    return out


def _check(verdicts, suite, name):
    status, detail = verdicts[(suite, name)]
    if status == "MISSING":
        pytest.skip("%s not built (oracle/build_ref_host.sh needs /root/reference)" % suite)
    assert status == "PASS", "%s: %s %s" % (name, status, detail)


@pytest.mark.parametrize("suite,name", QUICK[::4], ids=[n for _, n in QUICK[::4]])
def test_reference_int_case_on_emulator(suite, name):
    _check(_prefetch("emu", QUICK[::4], 120, dict(os.environ, OMP_NUM_THREADS="2"), "emu"), suite, name)


@pytest.mark.gpu
@pytest.mark.parametrize("suite,name", EXPECTED, ids=[n for _, n in EXPECTED])
def test_reference_int_case_on_gpu(suite, name):
    _check(_prefetch("gpu", EXPECTED, 180, dict(os.environ, OMP_NUM_THREADS="8"), "gpu"), suite, name)


def _multidev():
    p = os.path.join(ROOT, "tests", "golden", "ref_int_expected_pass_multidev.txt")
    return [tuple(l.rstrip("\n").split("\t", 1)) for l in open(p) if l.strip()] if os.path.exists(p) else []


MULTIDEV = _multidev()


@pytest.mark.parametrize("suite,name", MULTIDEV, ids=[n for _, n in MULTIDEV])
def test_reference_multi_device_case_on_emulator(suite, name):
    """The reference's multi-GPU tests -- nccl.tests.c (allreduce / broadcast / reduce, blocking and not), parallel.tests.c (its own
    data-parallel graph transformation with all-reduce) and the multi-device dynamic-graph cases -- through the unmodified host,
    on the emulator build with FOUR emulated devices and its in-process stand-in for RCCL: the single-process N-device form of
    section 8(e), which the one-GPU box cannot run."""
    _check(_prefetch("emu", MULTIDEV, 300, dict(os.environ, NNC_EMU_DEVICE_COUNT="4", OMP_NUM_THREADS="2"), "emu4"), suite, name)
