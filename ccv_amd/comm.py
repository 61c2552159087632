"""Data-parallel exchange for the one-process-per-GPU deployment.

The reference shards a minibatch over the GPUs of one node and all-reduces every parameter gradient with one NCCL call
per tensor (ccv_nnc_symbolic_graph_parallel.c:545-575).  Here every rank holds a full replica, parameter gradients live
in ONE flat arena (VGGD(flat_grads=True)), and the exchange is a single COMM_ALLREDUCE command over that arena: on xGMI
(7 point-to-point links per GPU) one 444 MB ring/tree collective is bandwidth-bound, 32 separate ones -- most of them a
few hundred bytes -- are latency-bound.  The gradient scale 1/(batch * world) is folded into the SGD command
(bin/nnc/imagenet.c:314-317), so the sum needs no extra pass.

The transport is the COMM_* commands of libnnc_mi355x.so on an RCCL communicator spanning the processes; the three hooks at the top of the
class (_init_transport, _collective, _order) are what the world_size-2 CPU tests of this bucketing logic replace with gloo on CPU tensors
(tests/gloo_comm.py) -- the product package itself imports neither torch nor anything of the checker.
"""
import ctypes as C
import numpy as np
from . import nnc


class ProcessComm:
    def __init__(self, lib, dist, rank, world):
        """dist: anything with broadcast_object_list(list, src) -- only rank 0's RCCL id travels through it."""
        self.lib, self.dist, self.rank, self.world = lib, dist, rank, world
        self._init_transport()
        self._allreduce = nnc.generic_cmd("COMM_ALLREDUCE_FORWARD")
        self._broadcast = nnc.generic_cmd("COMM_BROADCAST_FORWARD")

    # ---- transport hooks -----------------------------------------------------------------------------------------------------
    def _init_transport(self):
        ids = [None]
        if self.rank == 0:
            buf = C.create_string_buffer(128)
            if self.lib.dll.nnc_mi355x_comm_unique_id(buf) != 0:
                raise RuntimeError("ncclGetUniqueId failed")
            ids = [buf.raw]
        self.dist.broadcast_object_list(ids, src=0)
        r = self.lib.dll.nnc_mi355x_comm_init_rank(C.c_char_p(ids[0]), self.rank, self.world)
        if r != 0:
            raise RuntimeError("nnc_mi355x_comm_init_rank failed: %d" % r)

    def _collective(self, cmd, t, stream, op):
        r = self.lib.cmd_exec(cmd, nnc.NO_HINT, 0, [t], [t], stream)
        if r != 0:
            raise RuntimeError("collective failed: %d" % r)

    def _signals(self, net, count):
        return [self.lib.signal_new(net.device) for _ in range(count)]

    def _order(self, first, then, signal):
        """what is enqueued on `then` from here on runs behind what `first` holds now"""
        self.lib.signal_emit(first, signal)
        self.lib.signal_wait(then, signal)

    def broadcast_params(self, net, stream=None):
        """Replicas start from rank 0's weights (_ccv_cnnp_model_copy_tensors, ccv_cnnp_model.c:1451-1452)."""
        for p, _, _ in net.params:
            self._collective(self._broadcast, p, stream, "bcast")

    # ---- overlapped form: the exchange rides a second stream while backward is still running -------------------------------
    # Backward produces the gradients last-layer first.  VGG-D's three fc layers hold 385 of the 444 MB and are finished a
    # few milliseconds into backward, so the arena is cut into buckets of consecutive layers (default: the fc block | the
    # conv block); a bucket's all-reduce is issued on `comm_stream` as soon as the bucket's earliest layer has been
    # enqueued on the compute stream (signal: compute -> comm), and the SGD commands wait for the last one (comm -> compute).
    # Few large collectives, because a ring over point-to-point xGMI links is bandwidth-bound only for large messages.
    def plan_overlap(self, net, comm_stream, bucket_bytes=64 << 20):
        nodes = [(i, n["arena"]) for i, n in enumerate(net.nodes) if "arena" in n]
        self._buckets = []   # (trigger node index, arena tensor slice), in backward order
        hi = None
        for i, (lo_off, hi_off) in reversed(nodes):
            if hi is None:
                hi = hi_off
            if (hi - lo_off) * 4 >= bucket_bytes or i == nodes[0][0]:
                self._buckets.append((i, net.grad_arena.alias((hi - lo_off,), lo_off)))
                hi = None
        self._comm_stream = comm_stream
        sigs = self._signals(net, len(self._buckets) + 1)
        self._sig_ready, self._sig_done = sigs[:-1], sigs[-1]
        self._next = 0

    def after_backward_node(self, net, i, stream):
        """pass as VGGD.backward(after_node=...)"""
        while self._next < len(self._buckets) and self._buckets[self._next][0] >= i:
            trigger, t = self._buckets[self._next]
            if trigger != i:
                break
            self._order(stream, self._comm_stream, self._sig_ready[self._next])
            self._collective(self._allreduce, t, self._comm_stream, "sum")
            self._next += 1

    def finish_overlap(self, stream):
        """every bucket issued; the compute stream (SGD) continues once the comm stream has drained them"""
        assert self._next == len(self._buckets), "backward did not reach every bucket"
        self._order(self._comm_stream, stream, self._sig_done)
        self._next = 0

    def allreduce_grads(self, net, stream=None):
        if getattr(net, "grad_arena", None) is not None:
            self._collective(self._allreduce, net.grad_arena, stream, "sum")
        else:
            for _, dp, _ in net.params:
                self._collective(self._allreduce, dp, stream, "sum")
