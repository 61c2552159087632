"""Data-parallel exchange for the one-process-per-GPU deployment.

The reference shards a minibatch over the GPUs of one node and all-reduces every parameter gradient with one NCCL call
per tensor (ccv_nnc_symbolic_graph_parallel.c:545-575).  Here every rank holds a full replica, parameter gradients live
in ONE flat arena (VGGD(flat_grads=True)), and the exchange is a single COMM_ALLREDUCE command over that arena: on xGMI
(7 point-to-point links per GPU) one 444 MB ring/tree collective is bandwidth-bound, 32 separate ones -- most of them a
few hundred bytes -- are latency-bound.  The gradient scale 1/(batch * world) is folded into the SGD command
(bin/nnc/imagenet.c:314-317), so the sum needs no extra pass.

transport="rccl": the COMM_* commands of libnnc_mi355x.so on an RCCL communicator spanning the processes.
transport="gloo": CPU tensors + torch.distributed (gloo) -- used by the world_size-2 CPU tests of this logic.
"""
import ctypes as C
import numpy as np
from . import nnc


class ProcessComm:
    def __init__(self, lib, dist, rank, world, transport="rccl"):
        self.lib, self.dist, self.rank, self.world, self.transport = lib, dist, rank, world, transport
        if transport == "rccl":
            ids = [None]
            if rank == 0:
                buf = C.create_string_buffer(128)
                if lib.dll.nnc_mi355x_comm_unique_id(buf) != 0:
                    raise RuntimeError("ncclGetUniqueId failed")
                ids = [buf.raw]
            dist.broadcast_object_list(ids, src=0)
            r = lib.dll.nnc_mi355x_comm_init_rank(C.c_char_p(ids[0]), rank, world)
            if r != 0:
                raise RuntimeError("nnc_mi355x_comm_init_rank failed: %d" % r)
        self._allreduce = nnc.generic_cmd("COMM_ALLREDUCE_FORWARD")
        self._broadcast = nnc.generic_cmd("COMM_BROADCAST_FORWARD")

    def _collective(self, cmd, t, stream, op):
        if self.transport == "rccl":
            r = self.lib.cmd_exec(cmd, nnc.NO_HINT, 0, [t], [t], stream)
            if r != 0:
                raise RuntimeError("collective failed: %d" % r)
        else:
            import torch
            base = t.owner if t.owner is not None else t
            x = torch.from_numpy(base.array.reshape(-1))
            if op == "sum":
                self.dist.all_reduce(x, op=self.dist.ReduceOp.SUM)
            else:
                self.dist.broadcast(x, src=0)

    def broadcast_params(self, net, stream=None):
        """Replicas start from rank 0's weights (_ccv_cnnp_model_copy_tensors, ccv_cnnp_model.c:1451-1452)."""
        for p, _, _ in net.params:
            self._collective(self._broadcast, p, stream, "bcast")

    def allreduce_grads(self, net, stream=None):
        if getattr(net, "grad_arena", None) is not None:
            self._collective(self._allreduce, net.grad_arena, stream, "sum")
        else:
            for _, dp, _ in net.params:
                self._collective(self._allreduce, dp, stream, "sum")
