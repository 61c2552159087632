"""The host side of the library under ThreadSanitizer and AddressSanitizer + UBSan (the N > 1 code cannot run on the one-GPU
box, so it is at least run under the sanitizers the reference's own build offers, lib/scheme.mk).  `make emu-tsan` / `make emu-asan` build the emulator
library instrumented (every lane of a kernel is a TSAN fiber and every fiber switch a synchronisation, tests/emu/emu_runtime.cpp: kernels never race with
themselves, what is checked is peephole.cpp / cmd_comm.cpp / device_rt.cpp under the tests' threads); the runs below preload the sanitizer's shared runtime
into an uninstrumented python or into the reference's own test binaries.  Zero reports is the bar.
What the first runs found: the emulator kept ONE current device for the process (HIP's is per thread), and the two "is anything recorded" flags
(g_comm_pending, g_deferred_live) were volatile ints read outside their mutexes -- now atomics."""
import glob
import os
import subprocess
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RT = "/opt/rocm/lib/llvm/lib/clang/22/lib/linux"
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _built(kind):
    rt = os.path.join(RT, "libclang_rt.%s-x86_64.so" % kind)
    if not os.path.exists(rt):
        pytest.skip("no %s runtime in this image" % kind)
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}
    r = subprocess.run(["make", "-s", "-j8", "-C", os.path.join(ROOT, "ccv_amd", "csrc"), "emu-" + kind], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return rt


def _san_env(kind, rt, tmp_path, **extra):
    env = dict(os.environ, NNC_EMU_BUILD=kind, LD_PRELOAD=rt, **extra)
    log = str(tmp_path / ("%s_report" % kind))
    if kind == "tsan":
        # (thread leaks: the reference host never joins the thread its stream callbacks are dispatched on, lib/nnc/ccv_nnc_stream.c _ccv_nnc_async_dispatch)
        env["TSAN_OPTIONS"] = "report_signal_unsafe=0 report_thread_leaks=0 exitcode=66 log_path=%s suppressions=%s" % (log, os.path.join(ROOT, "tests", "tsan.supp"))
    else:
        # (leaks: python and the reference host keep process-lifetime allocations; swapcontext: the emulator's lanes are ucontext fibers)
        env["ASAN_OPTIONS"] = "detect_leaks=0 exitcode=66 log_path=%s detect_stack_use_after_return=0" % log
        env["UBSAN_OPTIONS"] = "print_stacktrace=1 halt_on_error=1 exitcode=66 log_path=%s" % log
    return env, log


def _reports(log):
    out = []
    for f in glob.glob(log + ".*"):
        txt = open(f, errors="replace").read()
        if "WARNING: ThreadSanitizer" in txt or "ERROR: AddressSanitizer" in txt or "runtime error:" in txt:
            out.append(txt[:3000])
    return out


def _pytest_job(kind, rt, tmp_path, files, extra_args=()):
    env, log = _san_env(kind, rt, tmp_path)
    r = subprocess.run([sys.executable, "-m", "pytest", "-m", "not gpu", "-q", "-x", "-p", "no:cacheprovider"] + list(extra_args) + [os.path.join(ROOT, "tests", f) for f in files],
                       capture_output=True, text=True, timeout=3000, env=env, cwd=ROOT)
    return r.returncode, r.stdout[-3000:] + r.stderr[-2000:], _reports(log)


THREADED = ["test_peephole.py", "test_comm_multidev.py", "test_staging_ring.py", "test_bn_cluster.py"]


@pytest.fixture(scope="module")
def runs(tmp_path_factory):
    """Every sanitizer job of this module started at once (they are independent processes; the instrumented convolutions of the reference's two
    data-parallel cases alone take minutes), the tests below look their verdicts up."""
    from concurrent.futures import ThreadPoolExecutor
    import ref_int_tests as R
    from test_ref_int import MULTIDEV
    tsan, asan = _built("tsan"), _built("asan")
    tmp = tmp_path_factory.mktemp("san")
    out = {}
    with ThreadPoolExecutor(max_workers=int(os.environ.get("NNC_SANITIZER_JOBS", "8"))) as ex:
        (tmp / "t").mkdir(); (tmp / "a").mkdir(); (tmp / "i").mkdir()
        out["tsan_pytest"] = ex.submit(_pytest_job, "tsan", tsan, tmp / "t", THREADED, ["-k", "not dispatch_order"])
        # (test_palettize.py: byte streams with ragged tails read through 8-byte loads and expanded into shadow tensors on the wrapper's frame -- ASan + UBSan territory)
        out["asan_pytest"] = ex.submit(_pytest_job, "asan", asan, tmp / "a", ["test_smoke_emu.py", "test_palettize.py"] + THREADED, ["-k", "not dispatch_order"])
        env, log = _san_env("tsan", tsan, tmp / "i", NNC_EMU_DEVICE_COUNT="4", OMP_NUM_THREADS="2", LD_LIBRARY_PATH=os.path.join(ROOT, "tests", "emu", "_build_tsan"))
        cases = []
        for suite, name in MULTIDEV:
            b = os.path.join(R.BIN, "%s.emu" % suite)
            cases.append((name, ex.submit(R.run_case, b, name, 1500, env) if os.path.exists(b) else None))
        out = {k: v.result() for k, v in out.items()}
        out["int_cases"] = [(n, f.result() if f is not None else ("MISSING", "")) for n, f in cases]
        out["int_reports"] = _reports(log)
    return out


@pytest.mark.timeout(2400)
def test_thread_sanitizer_on_the_threaded_host_paths(runs):
    """the look-ahead under foreign-thread flushes, the collectives' queue (two stream contexts, the lock-order regression, the 8-device queue-overflow
    stress), the staging ring, and the cluster kernels' concurrent launches"""
    rc, tail, rep = runs["tsan_pytest"]
    assert rc == 0 and not rep and " passed" in tail, tail + "\n".join(rep)[:4000]


@pytest.mark.timeout(2400)
def test_thread_sanitizer_on_the_reference_multi_device_cases(runs):
    """the reference's own nccl / parallel / multi-device dynamic-graph int cases (its host: coroutine scheduler threads, N stream contexts) on four emulated
    devices, the library instrumented -- among them parallel.tests.c's DP(2 x 16) == single(32) under REQUIRE_TENSOR_EQ"""
    if any(v[0] == "MISSING" for _, v in runs["int_cases"]):
        pytest.skip("reference int binaries not built (oracle/build_ref_host.sh needs /root/reference)")
    bad = [(n, v) for n, v in runs["int_cases"] if v[0] != "PASS"]
    assert not bad and len(runs["int_cases"]) >= 15 and not runs["int_reports"], str(bad) + "\n".join(runs["int_reports"])[:4000]


@pytest.mark.timeout(2400)
def test_address_and_undefined_behaviour_sanitizers(runs):
    """the same library under AddressSanitizer + UBSan: the VGG-style smoke step, the look-ahead, the collectives, the staging ring, the cluster kernels"""
    rc, tail, rep = runs["asan_pytest"]
    assert rc == 0 and not rep and " passed" in tail, tail + "\n".join(rep)[:4000]
