"""bench.py's contract is ONE JSON line on stdout; libraries (RCCL's version banner when a communicator is created) and child
processes write to descriptor 1 too.  claim_stdout() points descriptor 1 at stderr and keeps a private duplicate for the line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = """
import os, sys, ctypes
sys.path.insert(0, %r)
import bench
bench.claim_stdout()
ctypes.CDLL(None).puts(b"banner from a C library")      # C stdio on descriptor 1, buffered until exit like RCCL's
os.write(1, b"raw write to descriptor 1\\n")
print("python print")
bench.emit({"metric": "m", "value": 1.5})
"""


def test_stdout_carries_exactly_the_json_line():
    r = subprocess.run([sys.executable, "-c", SCRIPT % ROOT], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and json.loads(lines[0]) == {"metric": "m", "value": 1.5}, r.stdout
    assert "banner from a C library" in r.stderr and "raw write" in r.stderr and "python print" in r.stderr


def test_gpus_n_without_devices_fails_loudly():
    """`python bench.py --gpus 2` on a box without (enough) GPUs must not print a line for fewer GPUs than asked: non-zero exit, nothing on stdout."""
    if os.path.exists("/dev/kfd"):
        import pytest
        pytest.skip("a GPU box: covered by the gpu-tier self-launch test")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == "", (r.returncode, r.stdout)
    assert "HIP device" in r.stderr or "not built" in r.stderr, r.stderr[-1000:]


import pytest  # noqa: E402


@pytest.mark.gpu
def test_self_launched_ranks_on_this_box():
    """`python bench.py --gpus N` with no launcher starts its own N ranks (one per GPU, RCCL communicator of N, rendezvous over
    ccv_amd/ctl.py).  N = the devices this box has (1 on the test box: the launcher path is forced, the rank goes through the whole N > 1 code -- RCCL
    id hand-over, parameter broadcast, bucketed overlapped all-reduce, the exchange check); asking for one GPU more than there are is refused."""
    from ccv_amd import nnc
    n = nnc.load().device_count()
    env = dict(os.environ, NNC_BENCH_FORCE_SELF_LAUNCH="1", NNC_BENCH_FORCE_COMM="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--batch", "16", "--no-cpu-baseline", "--no-via-host", "--no-alt-leg"],
                       capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["config"]["rccl_ranks"] == n and out["config"]["launcher"].startswith("self")
    assert out["config"]["data_parallel_check"]["ok"], out["config"]["data_parallel_check"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1), "--steps", "1", "--warmup", "0", "--batch", "16"], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and r.stdout.strip() == "" and "visible" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("config", ["resnet50-nchw-bs256", "cifar10-dawn-f16-bs512"])
def test_process_per_gpu_form_of_configs_4_and_5_on_this_box(config):
    """Round 4: configs 4 / 5 at --gpus N run one harness process per GPU (tools/host_resnet_bench.c with HOST_BENCH_WORLD / RANK / DEVICE / COMM_ID).  N = the
    devices this box has; on the one-GPU box NNC_BENCH_FORCE_COMM=1 sends the single rank through the same code (RCCL communicator of one, the reference's
    evaluate / backward / parameter_gradients_map(COMM_ALLREDUCE) / apply_gradients step, the cross-rank barrier around the timed steps)."""
    from ccv_amd import nnc
    n = nnc.load().device_count()
    env = dict(os.environ, NNC_BENCH_FORCE_COMM="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", config, "--gpus", str(n), "--steps", "2", "--warmup", "1", "--batch", "32", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and "one process per GPU" in out["config"]["parallelism"] and ("RCCL ranks %s" % ([n] * n)) in out["config"]["parallelism"], out["config"]["parallelism"]
    assert out["config"]["outputs_finite"] and out["value"] > 0
    assert out["config"]["rccl_ranks"] == n and out["config"]["data_parallel_check"]["ok"], out["config"].get("data_parallel_check")  # (round 5: every config's N > 1 line carries both)


@pytest.mark.gpu
def test_launched_by_torch_distributed_run_as_the_driver_does():
    """The driver's own N > 1 spelling: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`.
    N = the devices this box has; with one device NNC_BENCH_FORCE_COMM=1 takes the rank through the N > 1 code (control plane over ccv_amd/ctl.py keyed by
    MASTER_PORT, RCCL communicator, broadcast, overlapped all-reduce).  Exactly one JSON line on stdout, from rank 0."""
    import importlib.util
    if importlib.util.find_spec("torch") is None:  # (found, NOT imported: torch's wheel brings a second HIP runtime into the process -- ccv_amd/ctl.py -- and this
        pytest.skip("no torch.distributed.run here")  # process has the backend loaded; the launcher runs in a child)
    from ccv_amd import nnc
    n = nnc.load().device_count()
    env = dict(os.environ, NNC_BENCH_FORCE_COMM="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", "29533",
                        os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--batch", "16", "--no-cpu-baseline", "--no-via-host", "--no-alt-leg", "--no-extra-configs"],
                       capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["config"]["rccl_ranks"] == n and out["steps"] == 2 and out["value"] > 0
    assert out["config"]["data_parallel_check"]["ok"], out["config"]["data_parallel_check"]
