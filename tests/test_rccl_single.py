"""The RCCL leg of the data-parallel exchange on real hardware, as far as a one-GPU box allows: a communicator of size
one built through the same C-ABI calls every rank makes (`nnc_mi355x_comm_unique_id` / `nnc_mi355x_comm_init_rank`), then
COMM_ALLREDUCE / COMM_BROADCAST commands on a stream.  A sum over one rank must return the input bit for bit.  (The
world-size-2 logic is covered on CPU with gloo in test_data_parallel.py; the 8-GPU run is the driver's.)"""
import numpy as np
import pytest
from ccv_amd import nnc
from ccv_amd.comm import ProcessComm


class _SoloDist:
    def broadcast_object_list(self, objs, src=0):
        return None


@pytest.mark.gpu
def test_rccl_world_of_one_allreduce_and_broadcast(gpu_lib):
    L = gpu_lib
    comm = ProcessComm(L, _SoloDist(), 0, 1, transport="rccl")
    s = L.stream_new(0)
    x = np.random.default_rng(0).standard_normal(1 << 20).astype(np.float32)
    t = L.tensor(nnc.GPU_TENSOR_NHWC(0, nnc.CCV_32F, x.size), x)
    comm._collective(comm._allreduce, t, s, "sum")
    comm._collective(comm._broadcast, t, s, "bcast")
    L.stream_wait(s)
    assert np.array_equal(t.numpy(), x)
    L.stream_free(s)
