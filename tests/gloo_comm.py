"""TEST INFRASTRUCTURE: ccv_amd/comm.py's bucketing / overlap logic with its three transport hooks replaced by torch.distributed (gloo)
on CPU tensors -- what the world_size-2 CPU tests run (tests/test_data_parallel.py).  Compute in those tests is the oracle; the product's
transport is RCCL behind the COMM_* commands (ccv_amd/comm.py), which needs GPUs."""
import numpy as np
import torch
from ccv_amd.comm import ProcessComm


class GlooProcessComm(ProcessComm):
    def _init_transport(self):
        pass  # the process group is the caller's (dist.init_process_group("gloo", ...))

    def _collective(self, cmd, t, stream, op):
        base = t.owner if t.owner is not None else t
        flat = base.array.reshape(-1)
        first = (t.ptr - base.ptr) // flat.itemsize   # a dense alias covers [first, first + count) of its owner
        x = torch.from_numpy(flat[first:first + int(np.prod(t.dims))])
        if op == "sum":
            self.dist.all_reduce(x, op=self.dist.ReduceOp.SUM)
        else:
            self.dist.broadcast(x, src=0)

    def _signals(self, net, count):
        return [None] * count

    def _order(self, first, then, signal):
        pass  # CPU tensors, blocking collectives: program order is the order
