#!/usr/bin/env python
"""Timing probe of the RCCL leg with a communicator of one (what a 1-GPU box can tell): init, all-reduce latency by size,
and the overlapped VGG-style step against the plain one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ccv_amd import nnc
from ccv_amd.comm import ProcessComm
from ccv_amd.vgg import VGGD


class Solo:
    def broadcast_object_list(self, objs, src=0): return None


L = nnc.load()
t0 = time.time(); comm = ProcessComm(L, Solo(), 0, 1); print("comm init %.2f s" % (time.time() - t0), flush=True)
s, cs = L.stream_new(0), L.stream_new(0)
for n in (1 << 10, 1 << 20, 100 << 20):
    t = L.tensor(nnc.GPU_TENSOR_NHWC(0, nnc.CCV_32F, n))
    for rep in range(3):
        t0 = time.time(); comm._collective(comm._allreduce, t, cs, "sum"); L.stream_wait(cs)
        print("allreduce %d floats: %.2f ms" % (n, 1e3 * (time.time() - t0)), flush=True)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
t0 = time.time(); net = VGGD(L, batch, seed=0, flat_grads=True, sgd=(0, 0.001, 1.0 / batch, 0.0005, 0.9, 0.9)); print("net build %.2f s" % (time.time() - t0), flush=True)
rng = np.random.default_rng(0)
net.set_input(rng.random((batch, 225, 225, 3), dtype=np.float32), rng.integers(0, 1000, batch))
comm.plan_overlap(net, cs)
print("buckets:", [(i, t.dims) for i, t in comm._buckets], flush=True)
for mode in ("plain", "overlap", "plain", "overlap"):
    for it in range(3):
        t0 = time.time()
        net.forward(s)
        if mode == "overlap":
            net.backward(s, after_node=lambda i: comm.after_backward_node(net, i, s)); comm.finish_overlap(s)
        else:
            net.backward(s); comm.allreduce_grads(net, s)
        net.update(s)
        L.stream_wait(s)
        print("%s step %d: %.1f ms" % (mode, it, 1e3 * (time.time() - t0)), flush=True)
