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


@pytest.fixture(scope="module")
def solo_comm(gpu_lib):
    # one process communicator per process (nnc_mi355x_comm_init_rank refuses a second one), as in a real rank
    return ProcessComm(gpu_lib, _SoloDist(), 0, 1)


@pytest.mark.gpu
def test_rccl_world_of_one_allreduce_and_broadcast(gpu_lib, solo_comm):
    L = gpu_lib
    comm = solo_comm
    s = L.stream_new(0)
    x = np.random.default_rng(0).standard_normal(1 << 20).astype(np.float32)
    t = L.tensor(nnc.GPU_TENSOR_NHWC(0, nnc.CCV_32F, x.size), x)
    comm._collective(comm._allreduce, t, s, "sum")
    comm._collective(comm._broadcast, t, s, "bcast")
    L.stream_wait(s)
    assert np.array_equal(t.numpy(), x)
    L.stream_free(s)


@pytest.mark.gpu
def test_rccl_world_of_one_half_precision(gpu_lib, solo_comm):
    """The CCV_16F COMM rows (comm_gpu_nccl.cu:65-206 registers CCV_32F | CCV_16F) through real RCCL: ncclHalf over a communicator of one
    returns the halves bit for bit; the row is found by the dispatcher for half tensors."""
    L = gpu_lib
    s = L.stream_new(0)
    x = np.random.default_rng(1).standard_normal(1 << 18).astype(np.float16)
    t = L.tensor(nnc.GPU_TENSOR_NHWC(0, nnc.CCV_16F, x.size), x)
    ar = nnc.generic_cmd("COMM_ALLREDUCE_FORWARD")
    assert L.cmd_exec(ar, nnc.NO_HINT, 0, [t], [t], s) == 0
    assert L.cmd_exec(nnc.generic_cmd("COMM_BROADCAST_FORWARD"), nnc.NO_HINT, 0, [t], [t], s) == 0
    L.stream_wait(s)
    assert np.array_equal(t.numpy(), x)
    L.stream_free(s)


@pytest.mark.gpu
def test_overlapped_bucketed_exchange_world_of_one(gpu_lib, solo_comm):
    """bench.py's N > 1 step -- buckets all-reduced on a second HIP stream behind signals while backward runs -- with a
    communicator of one: two steps must leave exactly the parameters of the plain single-GPU step."""
    from ccv_amd.vgg import VGGD
    L = gpu_lib
    layers = [("conv", 32), ("pool",), ("conv", 32), ("pool",), ("fc", 64), ("fc", 10)]
    rng = np.random.default_rng(4)
    x, y = rng.random((8, 31, 31, 3), dtype=np.float32), rng.integers(0, 10, 8)
    res = []
    for overlapped in (False, True):
        net = VGGD(L, 8, input_hw=31, layers=layers, seed=2, flat_grads=overlapped, sgd=(0, 0.01, 1.0 / 8, 0.0005, 0.9, 0.9))
        s = L.stream_new(0)
        net.set_input(x, y)
        if overlapped:
            comm = solo_comm
            cs = L.stream_new(0)
            comm.plan_overlap(net, cs, bucket_bytes=4096)
            assert len(comm._buckets) >= 2
        for _ in range(2):
            net.forward(s)
            if overlapped:
                net.backward(s, after_node=lambda i: comm.after_backward_node(net, i, s))
                comm.finish_overlap(s)
            else:
                net.backward(s)
            net.update(s)
        L.stream_wait(s)
        res.append([p.numpy() for p, _, _ in net.params])
    for a, b in zip(*res):
        assert np.array_equal(a, b)
