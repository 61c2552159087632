"""End-to-end: one VGG-style training step (forward + backward + SGD) through the command interface, backend under
test vs the reference's own CPU backend, same weights / inputs.  Small spatial size so the oracle finishes in seconds;
the full-size configuration is exercised by bench.py and checked there through the loss value."""
import numpy as np
from ccv_amd import nnc
from ccv_amd.vgg import VGGD, VGG_D

MINI = [("conv", 8), ("conv", 8), ("pool",), ("conv", 16), ("conv", 16), ("pool",), ("fc", 32), ("fc", 10)]


def _run(lib, memory, backend, batch, hw, layers, pool_per_image, steps=2, fuse_relu=False):
    from oracle_vgg import make_vggd
    net = make_vggd(lib, batch, memory=memory, input_hw=hw, layers=layers, seed=1, backend=backend, pool_per_image=pool_per_image, fuse_relu=fuse_relu)
    rng = np.random.default_rng(5)
    out = []
    for s in range(steps):
        net.set_input(rng.random((batch, hw, hw, 3), dtype=np.float32), rng.integers(0, net.classes, batch))
        net.step()
        out.append((net.loss.numpy(), net.softmax.numpy()))
    params = [p.numpy() for p, _, _ in net.params]
    grads = [d.numpy() for _, d, _ in net.params]
    return out, params, grads


def test_vgg_mini_step_matches_reference(backend, ref_lib):
    got = _run(backend, nnc.GPU_MEMORY, None, 3, 23, MINI, False)
    want = _run(ref_lib, nnc.CPU_MEMORY, nnc.BACKEND_CPU_REF, 3, 23, MINI, True)
    for (l1, s1), (l2, s2) in zip(got[0], want[0]):
        np.testing.assert_allclose(l1, l2, rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(s1, s2, rtol=1e-4, atol=1e-6)
    for a, b in zip(got[2], want[2]):
        np.testing.assert_allclose(a, b, rtol=1e-3, atol=1e-5)
    for a, b in zip(got[1], want[1]):
        np.testing.assert_allclose(a, b, rtol=1e-4, atol=1e-6)


def test_vgg_mini_step_with_fused_relu_is_the_same_step(backend):
    """The convolutions rectifying in their own epilogue (NNC_MI355X_CONV_ALGO_FUSE_RELU) against conv followed by the in-place
    RELU_FORWARD: max(0, .) is exact, so losses, gradients and updated parameters are bit-identical."""
    a = _run(backend, nnc.GPU_MEMORY, None, 3, 23, MINI, False, fuse_relu=True)
    b = _run(backend, nnc.GPU_MEMORY, None, 3, 23, MINI, False, fuse_relu=False)
    for (l1, s1), (l2, s2) in zip(a[0], b[0]):
        assert np.array_equal(l1, l2) and np.array_equal(s1, s2)
    for x, y in zip(a[1] + a[2], b[1] + b[2]):
        assert np.array_equal(x, y)


def test_vgg_d_layer_table():
    from ccv_amd.vgg import vgg_d_flops_per_image
    fwd, both = vgg_d_flops_per_image()
    assert abs(fwd / 1e9 - 29.39) < 0.01 and abs(both / 1e9 - 88.01) < 0.01  # SURVEY.md section 8 / BASELINE.md section 2
