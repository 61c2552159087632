"""Test tiers.
  not gpu : host logic, the C oracle vs the reference's known answers / golden fixtures, C-ABI export check, and --
            on the CPU HIP emulator (tests/emu) -- the real kernel sources against the oracle at small sizes.
  gpu     : the parity tests proper, through the C-ABI of ccv_amd/lib/libnnc_mi355x.so on a real MI355X.
The same test bodies run in both tiers through the `backend` fixture.
"""
import os
import subprocess
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _has_gpu():
    return os.path.exists("/dev/kfd") and os.path.exists(os.path.join(ROOT, "ccv_amd", "lib", "libnnc_mi355x.so"))


def emu_build():
    """(make target, directory) of the emulator build under test: the plain one, or -- tests/test_sanitizers.py re-runs parts of this suite with
    NNC_EMU_BUILD=tsan / asan and the sanitizer's runtime preloaded -- the ThreadSanitizer / AddressSanitizer + UBSan build of the same sources."""
    kind = os.environ.get("NNC_EMU_BUILD", "")
    return ("emu-" + kind, "_build_" + kind) if kind else ("emu", "_build")


def emu_so():
    return os.path.join(ROOT, "tests", "emu", emu_build()[1], "libnnc_mi355x_emu.so")


@pytest.fixture(scope="session")
def emu_lib():
    """The product sources compiled against the CPU HIP emulator (test infrastructure only)."""
    from ccv_amd import nnc
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}  # (the compiler is not run under the sanitizer's runtime)
    r = subprocess.run(["make", "-s", "-j8", "-C", os.path.join(ROOT, "ccv_amd", "csrc"), emu_build()[0]], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    return nnc.load(emu_so())


@pytest.fixture(scope="session")
def gpu_lib():
    from ccv_amd import nnc
    return nnc.load()  # raises if the HIP library or the GPU is missing: never falls back


@pytest.fixture(scope="session")
def ref_lib():
    """The reference's own CPU backend (oracle/_ref/libccv_ref.so), prebuilt by oracle/build_ref.sh."""
    from ccv_amd import nnc
    p = os.path.join(ROOT, "oracle", "_ref", "libccv_ref.so")
    if not os.path.exists(p):
        if os.path.isdir("/root/reference/lib/nnc"):
            subprocess.check_call([os.path.join(ROOT, "oracle", "build_ref.sh")])
        else:
            pytest.skip("oracle/_ref/libccv_ref.so not built and /root/reference absent")
    from oracle_bind import CheckerLib
    return CheckerLib(p, "reference")


def pytest_generate_tests(metafunc):
    # `backend` fixture: "emu" in the CPU tier, "gpu" in the GPU tier.
    if "backend" in metafunc.fixturenames:
        metafunc.parametrize("backend", [pytest.param("emu"), pytest.param("gpu", marks=pytest.mark.gpu)], indirect=True)


@pytest.fixture
def backend(request):
    if request.param == "emu":
        return request.getfixturevalue("emu_lib")
    return request.getfixturevalue("gpu_lib")
