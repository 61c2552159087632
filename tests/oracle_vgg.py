"""TEST INFRASTRUCTURE: the VGG-D command driver (ccv_amd/vgg.py) pointed at the checker -- the reference's own CPU backend
(oracle/_ref/libccv_ref.so) or the C restatement (oracle/libnnc_oracle.so), CPU tensors.

The reference's CPU pooling loops walk ONE image of a batch (lib/nnc/cmd/pool/ccv_nnc_max_pool_cpu_ref.c:37-63 has no batch loop; the GPU
backend being replaced pools the whole batch), so every driver of the oracle issues the pooling commands image by image.  That is a property
of the checker, not of the product: it lives here, and only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import it."""
from ccv_amd.vgg import VGGD


class OracleVGGD(VGGD):
    def _pool(self, cmd, hint, ins, outs, stream, tag, hook):
        for i in range(self.batch):
            vi = [self._img(t, i) for t in ins]
            vo = [self._img(t, i) for t in outs]
            self._exec(cmd, hint, 0, vi, vo, stream, tag, hook)


def make_vggd(lib, *args, pool_per_image=False, **kw):
    """VGGD for a library that pools whole batches, OracleVGGD for one that does not (oracle_bind.oracle_lib()'s third value)."""
    return (OracleVGGD if pool_per_image else VGGD)(lib, *args, **kw)
