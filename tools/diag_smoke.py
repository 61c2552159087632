"""GPU diagnostic: smoke()'s mini network on the FIRST step of a process, snapshots around pool_bwd/4 and relu_bwd/3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from ccv_amd import nnc
from oracle_vgg import make_vggd
from oracle_bind import oracle_lib

L = nnc.load()
O, backend, per_image = oracle_lib()
mini = [("conv", 16), ("conv", 16), ("pool",), ("conv", 32), ("pool",), ("fc", 64), ("fc", 10)]
for trial in range(2):
    rng = np.random.default_rng(trial)
    x, y = rng.random((4, 33, 33, 3), dtype=np.float32), rng.integers(0, 10, 4)
    snaps = {}
    def run(lib, mem, be, ppi, key):
        net = make_vggd(lib, 4, memory=mem, input_hw=33, layers=mini, seed=2 + trial, backend=be, pool_per_image=ppi)
        net.set_input(x, y)
        if mem == nnc.GPU_MEMORY: lib.stream_wait(None)
        n3, n4 = net.nodes[3], net.nodes[4]
        def hook(ev, tag):
            if mem == nnc.GPU_MEMORY: lib.stream_wait(None)
            if tag in ("pool_bwd/4", "relu_bwd/3", "conv_bwd/3") and (ev == "begin" or tag != "pool_bwd/4" or True):
                snaps[(key, tag, ev)] = dict(g4=net.grads[id(n4["b"])].numpy().copy(), g3=net.grads[id(n3["b"])].numpy().copy(), a3=n3["b"].numpy().copy(), b4=n4["b"].numpy().copy())
        net.step(None, hook)
        return net
    gnet = run(L, nnc.GPU_MEMORY, None, False, "gpu")
    rnet = run(O, nnc.CPU_MEMORY, backend, per_image, "ref")
    print("trial", trial)
    for tag, ev in (("pool_bwd/4", "begin"), ("pool_bwd/4", "end"), ("relu_bwd/3", "end")):
        if ("ref", tag, ev) not in snaps: continue
        g, r = snaps[("gpu", tag, ev)], snaps[("ref", tag, ev)]
        # the ref's per-image pool issues several commands under one tag: its last 'end' snapshot is the complete one
        for k in ("g4", "g3", "a3", "b4"):
            d = np.abs(g[k] - r[k])
            print("  %-11s %-5s %-3s max err %.3g  wrong %d of %d" % (tag, ev, k, d.max(), (d > 1e-5).sum(), d.size))
        if tag == "pool_bwd/4" and ev == "end":
            d = np.abs(g["g3"] - r["g3"])
            idx = np.argwhere(d > 1e-5)[:12]
            for n_, y_, x_, c_ in idx:
                print("    g3[%d,%d,%d,%d] gpu %.6g ref %.6g  a3 %.6g  window outs:" % (n_, y_, x_, c_, g["g3"][n_, y_, x_, c_], r["g3"][n_, y_, x_, c_], g["a3"][n_, y_, x_, c_]),
                      [(oy, ox, float(g["b4"][n_, oy, ox, c_]), float(g["g4"][n_, oy, ox, c_])) for oy in range(7) for ox in range(7) if 2 * oy <= y_ <= 2 * oy + 2 and 2 * ox <= x_ <= 2 * ox + 2])
