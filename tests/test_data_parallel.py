"""N > 1 path on CPU: world_size-2 gloo processes, each running its shard of the minibatch through the command
interface, exchanging the flat gradient arena, applying SGD -- must equal the single-process run on the whole minibatch
(the reference's own criterion: DP(2 x 16) == single(32), test/int/nnc/parallel.tests.c:192-369).
Compute here is the oracle (CPU tensors) and the transport gloo (tests/gloo_comm.py: ccv_amd/comm.py with its transport hooks replaced);
the product form of the same logic is ccv_amd/comm.py on RCCL."""
import os
import subprocess
import sys
import numpy as np
from ccv_amd import nnc
from harness import tensor_eq

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MINI = [("conv", 8), ("pool",), ("conv", 16), ("pool",), ("fc", 32), ("fc", 10)]
WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import torch.distributed as dist
from ccv_amd import nnc
from oracle_vgg import make_vggd
from gloo_comm import GlooProcessComm
from oracle_bind import oracle_lib
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
O, backend, per_image = oracle_lib()
B = 4
net = make_vggd(O, B // world, memory=nnc.CPU_MEMORY, input_hw=19, layers=%(layers)r, seed=3 + rank, backend=backend, pool_per_image=per_image,
           flat_grads=True, sgd=(0, 0.01, 1.0 / B, 0.0005, 0.9, 0.9))
comm = GlooProcessComm(O, dist, rank, world)
comm.broadcast_params(net)   # rank 1 was seeded differently on purpose
comm.plan_overlap(net, None, bucket_bytes=2048)   # several buckets even on this tiny net
assert len(comm._buckets) >= 2
rng = np.random.default_rng(11)
for step in range(2):
    x, y = rng.random((B, 19, 19, 3), dtype=np.float32), rng.integers(0, 10, B)
    s = slice(rank * B // world, (rank + 1) * B // world)
    net.set_input(x[s], y[s])
    net.forward()
    if step == 0:   # the bucketed, backward-interleaved exchange bench.py uses ...
        net.backward(after_node=lambda i: comm.after_backward_node(net, i, None)); comm.finish_overlap(None)
    else:           # ... and the single flat collective: same result
        net.backward(); comm.allreduce_grads(net)
    net.update()
if rank == 0:
    np.savez(sys.argv[1], *[p.numpy() for p, _, _ in net.params])
dist.barrier(); dist.destroy_process_group()
'''


def test_dp2_equals_single(ref_lib, tmp_path):
    from oracle_vgg import make_vggd
    from oracle_bind import oracle_lib
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(root=ROOT, layers=MINI))
    out = tmp_path / "dp.npz"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script), str(out)], env=dict(env, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r))) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    dp = np.load(out)
    O, backend, per_image = oracle_lib()
    B = 4
    net = make_vggd(O, B, memory=nnc.CPU_MEMORY, input_hw=19, layers=MINI, seed=3, backend=backend, pool_per_image=per_image, sgd=(0, 0.01, 1.0 / B, 0.0005, 0.9, 0.9))
    rng = np.random.default_rng(11)
    for step in range(2):
        x, y = rng.random((B, 19, 19, 3), dtype=np.float32), rng.integers(0, 10, B)
        net.set_input(x, y)
        net.step()
    for i, (p, _, _) in enumerate(net.params):
        a, b = dp["arr_%d" % i], p.numpy()
        assert tensor_eq(a, b) or np.allclose(a, b, rtol=2e-5, atol=1e-7), i


def test_comm_commands_single_process_clique(emu_lib):
    """COMM_ALLREDUCE / BROADCAST / REDUCE in the reference's single-process N-device form (nccl.tests.c:14-226) on the
    emulator's 4 fake devices."""
    os.environ["NNC_EMU_DEVICE_COUNT"] = "4"
    L = emu_lib
    n, cnt = 4, 1000
    rng = np.random.default_rng(0)
    src = [rng.random(cnt, dtype=np.float32) for _ in range(n)]
    ts = [L.tensor(nnc.GPU_TENSOR_NHWC(d, nnc.CCV_32F, cnt), src[d]) for d in range(n)]
    assert L.cmd_exec(nnc.generic_cmd("COMM_ALLREDUCE_FORWARD"), nnc.NO_HINT, 0, ts, ts) == 0
    want = src[0] + src[1] + src[2] + src[3]
    for t in ts:
        np.testing.assert_allclose(t.numpy(), want, rtol=1e-6)
    out = [L.tensor(nnc.GPU_TENSOR_NHWC(d, nnc.CCV_32F, cnt)) for d in range(n)]
    assert L.cmd_exec(nnc.generic_cmd("COMM_BROADCAST_FORWARD"), nnc.NO_HINT, 0, [ts[0]], out) == 0
    for t in out:
        np.testing.assert_array_equal(t.numpy(), ts[0].numpy())
    ins = [L.tensor(nnc.GPU_TENSOR_NHWC(d, nnc.CCV_32F, cnt), src[d]) for d in range(n)]
    dst = L.tensor(nnc.GPU_TENSOR_NHWC(0, nnc.CCV_32F, cnt))
    assert L.cmd_exec(nnc.generic_cmd("COMM_REDUCE_FORWARD"), nnc.NO_HINT, 0, ins, [dst]) == 0
    np.testing.assert_allclose(dst.numpy(), want, rtol=1e-6)
