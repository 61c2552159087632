"""Control plane of the one-process-per-GPU deployment: rendezvous, barrier, object gather -- nothing on the data path.

bench.py's ranks are started by `python -m torch.distributed.run` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT in the
environment) on ONE node.  The data path is RCCL through libnnc_mi355x.so (COMM_* commands); what is left for a control plane
is handing rank 0's RCCL id to the others, barriers around the timed region and the max over the ranks' clocks.  That is a few
hundred bytes per run, done here over a Unix-domain socket so that the worker processes do not have to import torch at all:
torch's wheel carries its OWN copies of libamdhip64 / libhsa-runtime64 / librccl (ROCm 7.0, requested under unversioned names,
so the loader does not share them with /opt/rocm's 7.2 copies our library is linked to), and a process holding two HIP
runtimes aborted in their exit handlers ("double free or corruption") after a correct run on the MI355X.

A dead or stuck rank must not hang the job: a rank that has died leaves its peers inside an RCCL collective that never completes (a device-side wait no
host timeout reaches).  Every rank therefore keeps a second, otherwise silent connection in the same star (the LIVENESS socket) and a watchdog thread on it:
end-of-file without the orderly goodbye byte means the peer's process is gone -- rank 0 sees any rank die, every rank sees rank 0 die, and rank 0 going down
takes the rest with it -- and the watchdog ends the process at once (exit code 70, a line on stderr) instead of waiting on the GPU.  An overall deadline
(NNC_MI355X_CTL_DEADLINE_S seconds, or `deadline_s`; OFF unless asked for -- a healthy job may run for days) covers a rank that is alive but stuck (exit code 71).
A rank that ends normally says goodbye even if its caller forgot destroy_process_group() (an atexit hook): only a process that DIES reads as a death.

The class answers the subset of torch.distributed's module interface ccv_amd.comm.ProcessComm uses (broadcast_object_list,
barrier, all_gather_object), so either can be passed as its `dist`.
"""
import atexit
import os
import pickle
import select
import socket
import struct
import sys
import threading
import time


def _send(sock, obj):
    blob = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
    sock.sendall(struct.pack("<Q", len(blob)) + blob)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("control plane: peer closed the connection")
        buf += chunk
    return bytes(buf)


def _recv(sock):
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    return pickle.loads(_recv_exact(sock, n))


def default_path():
    """One socket per job: the launcher's port and run id name it, so back-to-back runs and concurrent jobs do not meet."""
    port = os.environ.get("MASTER_PORT", "0")
    run = os.environ.get("TORCHELASTIC_RUN_ID", "none")
    return os.path.join(os.environ.get("TMPDIR", "/tmp"), "nnc_mi355x_ctl_%s_%s_%d.sock" % (port, run, os.getuid()))


class LocalControl:
    """Star over a Unix socket: rank 0 listens, every collective is gather-to-0 + fan-out of the gathered list."""

    def __init__(self, rank, world, path=None, timeout=600.0, watchdog=True, deadline_s=None):
        self.rank, self.world = rank, world
        self.path = path or default_path()
        self.peers = []      # rank 0: sockets indexed by rank - 1
        self.sock = None     # others: the connection to rank 0
        self.live = []       # the liveness connections (rank 0: one per peer; others: the one to rank 0)
        self._closing = False
        if world == 1:
            return
        self._connect(rank, world, timeout)
        if watchdog:
            self.live = self._connect_liveness(rank, world, timeout)
            if deadline_s is None:
                deadline_s = float(os.environ.get("NNC_MI355X_CTL_DEADLINE_S", "0") or 0)  # 0 = no deadline (50 minutes by default killed healthy jobs)
            t = threading.Thread(target=self._watch, args=(time.time() + deadline_s if deadline_s > 0 else None,), daemon=True)
            t.start()
            atexit.register(self._goodbye)  # a normal interpreter exit without destroy_process_group() is not a death

    def _connect(self, rank, world, timeout):
        if rank == 0:
            try:
                os.unlink(self.path)
            except FileNotFoundError:
                pass
            srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            srv.bind(self.path)
            srv.listen(world)
            srv.settimeout(timeout)
            by_rank = {}
            while len(by_rank) < world - 1:
                c, _ = srv.accept()
                c.settimeout(timeout)
                r = _recv(c)
                by_rank[int(r)] = c
            srv.close()
            os.unlink(self.path)  # everyone is connected: the name is not needed any more
            self.peers = [by_rank[r] for r in range(1, world)]
        else:
            deadline = time.time() + timeout
            while True:
                s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
                try:
                    s.connect(self.path)
                    break
                except (FileNotFoundError, ConnectionRefusedError):  # rank 0 is not listening yet (or a stale name of an earlier run)
                    s.close()
                    if time.time() > deadline:
                        raise TimeoutError("control plane: rank 0 never listened on %s" % self.path)
                    time.sleep(0.05)
            s.settimeout(timeout)
            _send(s, rank)
            self.sock = s

    # ---- liveness: a second star of connections nobody writes to until the orderly goodbye ------------------------------------------
    def _connect_liveness(self, rank, world, timeout):
        if rank == 0:
            path = self.path + ".live"
            try:
                os.unlink(path)
            except FileNotFoundError:
                pass
            srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            srv.bind(path)
            srv.listen(world)
            srv.settimeout(timeout)
            self.all_gather(path)  # the name exists: the others may connect
            conns = []
            while len(conns) < world - 1:
                c, _ = srv.accept()
                conns.append(c)
            srv.close()
            os.unlink(path)
            return conns
        path = self.all_gather(None)[0]
        s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        s.settimeout(timeout)
        s.connect(path)
        s.settimeout(None)
        return [s]

    def _watch(self, deadline):
        pending = list(self.live)
        while pending and not self._closing:
            wait = 1.0 if deadline is None else max(0.0, min(1.0, deadline - time.time()))
            try:
                ready, _, _ = select.select(pending, [], [], wait)
            except (OSError, ValueError):  # our own sockets were closed underneath: an orderly shutdown on this side
                return
            for c in ready:
                try:
                    b = c.recv(1)
                except OSError:
                    b = b""
                if b == b"Q":      # the peer said goodbye: it is past its last collective
                    pending.remove(c)
                elif not self._closing:
                    sys.stderr.write("[nnc_mi355x ctl] rank %d: a peer process went away without finishing the job -- stopping instead of waiting in a collective it will never join\n" % self.rank)
                    sys.stderr.flush()
                    os._exit(70)
            if deadline is not None and time.time() >= deadline and not self._closing:
                sys.stderr.write("[nnc_mi355x ctl] rank %d: the job's deadline (NNC_MI355X_CTL_DEADLINE_S) passed -- stopping\n" % self.rank)
                sys.stderr.flush()
                os._exit(71)

    # ---- the one primitive ---------------------------------------------------------------------------------------------
    def all_gather(self, obj):
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            out = [obj] + [_recv(p) for p in self.peers]
            for p in self.peers:
                _send(p, out)
            return out
        _send(self.sock, obj)
        return _recv(self.sock)

    # ---- torch.distributed-shaped entry points (what ProcessComm / bench.py call) ---------------------------------------------
    def barrier(self):
        self.all_gather(None)

    def all_gather_object(self, out_list, obj):
        out_list[:] = self.all_gather(obj)

    def broadcast_object_list(self, objs, src=0):
        objs[:] = self.all_gather(list(objs))[src]

    def reduce_max(self, x):
        return max(self.all_gather(float(x)))

    def _goodbye(self):
        if self._closing:
            return
        self._closing = True
        for c in self.live:  # the orderly goodbye: end-of-file WITHOUT it is what the peers' watchdogs read as a death
            try:
                c.sendall(b"Q")
            except OSError:
                pass

    def destroy_process_group(self):
        self._goodbye()
        for p in self.peers:
            p.close()
        if self.sock:
            self.sock.close()
        self.peers, self.sock = [], None
        # (the liveness sockets stay open until the process ends: closing one here would race the peer's read of the goodbye byte)
