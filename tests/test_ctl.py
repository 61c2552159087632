"""The control plane bench.py's ranks use instead of torch.distributed (ccv_amd/ctl.py): real processes, world size 3, the
calls ProcessComm and bench.py make -- id hand-over from rank 0, barriers, gather of per-rank objects, max over the clocks --
and a stale socket name left behind by an earlier (crashed) job."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import os, sys, time
sys.path.insert(0, %(root)r)
from ccv_amd.ctl import LocalControl
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if rank == 0:
    time.sleep(0.3)          # the others are already knocking (on the stale name) when rank 0 starts listening
c = LocalControl(rank, world, timeout=60)
ids = [b"\x01\x02" * 64] if rank == 0 else [None]
c.broadcast_object_list(ids, src=0)
assert ids[0] == b"\x01\x02" * 64
got = [None] * world
c.all_gather_object(got, {"rank": rank, "v": [rank * 1.5] * 3})
assert [g["rank"] for g in got] == list(range(world)) and got[2]["v"] == [3.0] * 3
for _ in range(20):
    c.barrier()
assert c.reduce_max(10.0 + rank) == 10.0 + world - 1
assert "torch" not in sys.modules
c.barrier(); c.destroy_process_group()
print("rank %%d ok" %% rank)
'''


def test_local_control_world3(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER % dict(root=ROOT))
    env = dict(os.environ, MASTER_PORT="29643", TORCHELASTIC_RUN_ID="t1", TMPDIR=str(tmp_path))
    sys.path.insert(0, ROOT)
    from ccv_amd import ctl
    old = dict(os.environ)
    os.environ.update(MASTER_PORT="29643", TORCHELASTIC_RUN_ID="t1", TMPDIR=str(tmp_path))
    try:
        stale = ctl.default_path()
    finally:
        os.environ.clear(); os.environ.update(old)
    s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    s.bind(stale); s.close()   # a name nobody listens on
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), WORLD_SIZE="3", LOCAL_RANK=str(r)), stdout=subprocess.PIPE, text=True) for r in range(3)]
    outs = [p.communicate(timeout=120)[0] for p in procs]
    assert [p.returncode for p in procs] == [0, 0, 0], outs
    assert all("rank %d ok" % r in outs[r] for r in range(3))
    assert not os.path.exists(stale)


def test_local_control_world1_needs_no_socket(tmp_path):
    from ccv_amd.ctl import LocalControl
    c = LocalControl(0, 1, path=str(tmp_path / "never"))
    ids = [b"x"]
    c.broadcast_object_list(ids, src=0)
    c.barrier()
    assert ids == [b"x"] and c.reduce_max(2.5) == 2.5 and not os.path.exists(tmp_path / "never")


DYING = r'''
import os, sys, time
sys.path.insert(0, %(root)r)
from ccv_amd.ctl import LocalControl
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
c = LocalControl(rank, world, timeout=60)
c.barrier()
if rank == int(os.environ["DIES"]):
    os._exit(9)              # a crash: no goodbye
time.sleep(600)              # stands for a rank inside an RCCL collective the dead peer never joins (nothing on the host would time out)
'''


def _run_dying(tmp_path, dies, port):
    import time
    script = tmp_path / "d.py"
    script.write_text(DYING % dict(root=ROOT))
    env = dict(os.environ, MASTER_PORT=port, TORCHELASTIC_RUN_ID="t2", TMPDIR=str(tmp_path), DIES=str(dies))
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), WORLD_SIZE="3", LOCAL_RANK=str(r)), stderr=subprocess.PIPE, text=True) for r in range(3)]
    errs = [p.communicate(timeout=60)[1] for p in procs]
    return [p.returncode for p in procs], errs, time.time() - t0


def test_a_dead_rank_takes_the_job_down_instead_of_hanging_it(tmp_path):
    """Round 5: a rank that dies mid-job leaves its peers in a collective that never completes.  The liveness star + watchdog ends
    every surviving rank within seconds (exit code 70): rank 0 sees rank 2 go, the others see rank 0 go."""
    codes, errs, dt = _run_dying(tmp_path, 2, "29644")
    assert codes == [70, 70, 9] and dt < 30, (codes, errs, dt)
    assert "went away" in errs[0]
    codes, errs, dt = _run_dying(tmp_path, 0, "29645")
    assert codes == [9, 70, 70] and dt < 30, (codes, errs, dt)


def test_the_job_deadline_stops_a_stuck_rank(tmp_path):
    script = tmp_path / "s.py"
    script.write_text(r'''
import os, sys, time
sys.path.insert(0, %r)
from ccv_amd.ctl import LocalControl
c = LocalControl(int(os.environ["RANK"]), 2, timeout=60, deadline_s=2.0)
c.barrier()
time.sleep(600)
''' % ROOT)
    env = dict(os.environ, MASTER_PORT="29646", TORCHELASTIC_RUN_ID="t3", TMPDIR=str(tmp_path))
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), WORLD_SIZE="2"), stderr=subprocess.PIPE, text=True) for r in range(2)]
    errs = [p.communicate(timeout=60)[1] for p in procs]
    assert sorted(p.returncode for p in procs)[-1] == 71 and all(p.returncode in (70, 71) for p in procs), ([p.returncode for p in procs], errs)


GOING = r'''
import os, sys, time
sys.path.insert(0, %(root)r)
from ccv_amd.ctl import LocalControl
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
c = LocalControl(rank, world, timeout=60)
c.barrier()
if rank == 1:
    sys.exit(0)              # done early, and its caller forgot destroy_process_group(): a normal exit, not a death
time.sleep(3)
c.barrier() if False else None
c.destroy_process_group()
'''


def test_a_rank_that_exits_normally_is_not_a_death_and_there_is_no_default_deadline(tmp_path):
    """(a) a rank that finishes and leaves the interpreter without destroy_process_group() says goodbye from an atexit hook -- its peers
    keep running (they used to stop with exit code 70); (b) the job deadline is off unless asked for (it defaulted to 3000 s and killed healthy long jobs)."""
    import time
    from ccv_amd import ctl
    import inspect
    assert '"NNC_MI355X_CTL_DEADLINE_S", "0"' in inspect.getsource(ctl.LocalControl.__init__)
    script = tmp_path / "g.py"
    script.write_text(GOING % dict(root=ROOT))
    env = dict(os.environ, MASTER_PORT="29661", TORCHELASTIC_RUN_ID="t4", TMPDIR=str(tmp_path))
    env.pop("NNC_MI355X_CTL_DEADLINE_S", None)
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), WORLD_SIZE="3", LOCAL_RANK=str(r)), stderr=subprocess.PIPE, text=True) for r in range(3)]
    errs = [p.communicate(timeout=60)[1] for p in procs]
    assert [p.returncode for p in procs] == [0, 0, 0], ([p.returncode for p in procs], errs)
