"""VGG-D training step as a sequence of nnc commands -- the workload BASELINE.json's metric is quoted on.

Layer table: the reference's `vgg_d_params` (bin/vgg_models.inc:361-838): 225x225x3 input, conv1_1 with border 0
(-> 223), every other conv 3x3 / stride 1 / border 1, five 3x3 stride-2 border-0 max-pools (223 -> 111 -> 55 -> 27 ->
13 -> 6), fc 18432 -> 4096 -> 4096 -> 1000, ReLU (in place) after every conv / fc but the last.  The forward command
sequence is the one `ccv_nnc_simple_graph` builds in test/int/nnc/graph.vgg.d.tests.c:14-90 (CONVOLUTION_FORWARD,
MAX_POOL_FORWARD, GEMM_FORWARD with TRANSPOSE(0,1) weights, in-place RELU_FORWARD); the backward + update sequence is
what ccv_nnc_symbolic_graph_minimize derives for it: SOFTMAX_CROSSENTROPY, *_BACKWARD per node in reverse order, one
SGD_FORWARD per parameter tensor (32 tensors, 111.09 M parameters).

The driver only issues commands through the library object it is given; the checker's side of the parity tests (the same sequence
on CPU tensors, with the per-image pooling the reference's CPU loops need) is a subclass under tests/ (tests/oracle_vgg.py).
"""
import numpy as np
from . import nnc

# (kind, out_channels) ; conv = 3x3 stride 1, pool = 3x3 stride 2 border 0
VGG_D = [("conv", 64), ("conv", 64), ("pool",), ("conv", 128), ("conv", 128), ("pool",),
         ("conv", 256), ("conv", 256), ("conv", 256), ("pool",), ("conv", 512), ("conv", 512), ("conv", 512), ("pool",),
         ("conv", 512), ("conv", 512), ("conv", 512), ("pool",), ("fc", 4096), ("fc", 4096), ("fc", 1000)]


def vgg_d_flops_per_image(input_hw=225, layers=VGG_D, in_channels=3):
    """(forward, forward+backward) FLOP per image, 1 MAC = 2 FLOP; no dgrad for the first conv."""
    h = input_hw
    c = in_channels
    fwd = bwd = 0
    first = True
    for l in layers:
        if l[0] == "conv":
            oh = h - 2 if first else h
            f = 2 * oh * oh * l[1] * 9 * c
            fwd += f
            bwd += f if first else 2 * f
            h, c, first = oh, l[1], False
        elif l[0] == "pool":
            h = (h - 3) // 2 + 1
        else:
            k = h * h * c if h else c
            f = 2 * k * l[1]
            fwd += f
            bwd += 2 * f
            h, c = 0, l[1]
    return fwd, fwd + bwd


def hash_unit(n, stream):
    """[0, 1) floats 0 .. n-1 of stream `stream`: the counter hash of tools/host_vgg_bench.c (splitmix64 finaliser), so that the
    driver through the reference host and this one run on identical parameters / images / labels."""
    with np.errstate(over="ignore"):
        h = (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) ^ np.uint64(((stream + 1) * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF)
        h ^= h >> np.uint64(30); h *= np.uint64(0xBF58476D1CE4E5B9)
        h ^= h >> np.uint64(27); h *= np.uint64(0x94D049BB133111EB)
        h ^= h >> np.uint64(31)
    return (h >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)


class VGGD:
    def __init__(self, lib, batch, memory=nnc.GPU_MEMORY, device=0, input_hw=225, layers=VGG_D, classes=None, seed=0, backend=None,
                 sgd=(0, 0.001, None, 0.0005, 0.9, 0.9), train=True, init="numpy", flat_grads=False, fuse_relu=False):
        self.lib, self.batch, self.memory, self.device, self.backend = lib, batch, memory, device, backend
        # fuse_relu: the convolutions write max(0, .) themselves (NNC_MI355X_CONV_ALGO_FUSE_RELU, include/nnc_mi355x.h) and the in-place
        # RELU_FORWARD behind each of them is not issued -- this driver knows the ReLU is the convolution's only consumer; on the way back
        # the command that writes the gradient of a rectified map (convolution or max-pool backward, both read that map) masks it, and the
        # RELU_BACKWARD is not issued either.  Same numbers either way (tests/test_vgg_step.py).
        self.fuse_relu = fuse_relu
        self.conv_fwd_backend = None  # bench.py's CPU leg: route CONVOLUTION_FORWARD to another backend of the same library (CPU_OPT)
        self.layers = list(layers)
        self.train = train
        rng = np.random.default_rng(seed)
        self._hash_stream = 0
        F = nnc.CCV_32F
        mk = lambda *dims: lib.tensor(nnc.tensor_param(memory, nnc.NHWC, F, dims, device))
        self.x = mk(batch, input_hw, input_hw, 3)
        self.nodes = []   # forward nodes: dict(kind, cmd, hint, inputs, outputs, ...)
        self.params = []  # (w, dw, mom)
        h = w_ = input_hw
        c = 3
        cur = self.x
        first = True
        self.acts = [self.x]
        for l in self.layers:
            if l[0] == "conv":
                k = l[1]
                border = 0 if first else 1
                oh, ow = h + 2 * border - 2, w_ + 2 * border - 2
                out = mk(batch, oh, ow, k)
                wt, bt = self._param(rng, (k, 3, 3, c), 9 * c, init), self._param(rng, (k,), 0, init)
                self.nodes.append(dict(kind="conv", cmd=nnc.CMD_CONVOLUTION_FORWARD(1, k, 3, 3, c), bcmd=nnc.CMD_CONVOLUTION_BACKWARD(1, k, 3, 3, c),
                                       hint=nnc.HINT((1, 1), (border, border)), a=cur, w=wt, bias=bt, b=out, first=first, relu=True,
                                       flops=2.0 * batch * oh * ow * k * 9 * c))
                cur, h, w_, c, first = out, oh, ow, k, False
            elif l[0] == "pool":
                oh, ow = (h - 3) // 2 + 1, (w_ - 3) // 2 + 1
                out = mk(batch, oh, ow, c)
                self.nodes.append(dict(kind="pool", cmd=nnc.CMD_MAX_POOL_FORWARD(3, 3), bcmd=nnc.CMD_MAX_POOL_BACKWARD(3, 3),
                                       hint=nnc.HINT((2, 2), (0, 0)), a=cur, b=out))
                cur, h, w_ = out, oh, ow
            else:
                k = l[1]
                fan_in = h * w_ * c if h else c
                a2 = cur if len(cur.dims) == 2 else self._flat(cur, batch, fan_in)
                out = mk(batch, k)
                wt, bt = self._param(rng, (k, fan_in), fan_in, init), self._param(rng, (k,), 0, init)
                last = l is self.layers[-1]
                self.nodes.append(dict(kind="fc", cmd=nnc.CMD_GEMM_FORWARD(nnc.NO_TRANSPOSE, nnc.TRANSPOSE(0, 1)), bcmd=nnc.CMD_GEMM_BACKWARD(nnc.NO_TRANSPOSE, nnc.TRANSPOSE(0, 1)),
                                       hint=nnc.NO_HINT, a=a2, w=wt, bias=bt, b=out, first=False, relu=not last, flops=2.0 * batch * fan_in * k))
                cur, h, w_, c = out, 0, 0, k
        self.classes = c
        self.logits = cur
        self.label = lib.tensor(nnc.tensor_param(memory, nnc.NHWC, F, (batch,), device))
        self.loss = mk(batch)
        self.softmax = mk(batch, self.classes)
        if train:
            # gradient tensors: one per activation (in-place relu shares them), dw/dbias + momentum per parameter
            for n in self.nodes:
                n["g_b"] = None
            self.grads = {}
            for n in self.nodes:
                for key in ("a", "b"):
                    t = n[key]
                    if id(t) not in self.grads and t is not self.x and not (key == "a" and n.get("first")):
                        base = t.owner if t.owner is not None else t
                        if id(base) not in self.grads:
                            self.grads[id(base)] = lib.tensor(nnc.tensor_param(memory, nnc.NHWC, F, base.dims, device))
                        gb = self.grads[id(base)]
                        self.grads[id(t)] = gb if t.owner is None else self._flat(gb, *t.dims)
            scale = sgd[2] if sgd[2] is not None else 1.0 / batch
            self.sgd_cmd = nnc.CMD_SGD_FORWARD(sgd[0], sgd[1], scale, sgd[3], sgd[4], sgd[5])
            # flat_grads: every parameter gradient is a contiguous slice of ONE arena, so data-parallel training exchanges
            # gradients with a single large collective (ccv_amd/comm.py) instead of one per tensor.
            self.grad_arena = None
            if flat_grads:
                total = sum((int(np.prod(n[key].dims)) + 31) // 32 * 32 for n in self.nodes if n["kind"] in ("conv", "fc") for key in ("w", "bias"))
                self.grad_arena = lib.tensor(nnc.tensor_param(memory, nnc.NHWC, F, (total,), device))
                self._exec(nnc.CMD_SET_FORWARD(0), nnc.NO_HINT, 0, [], [self.grad_arena], None, "set0")
            arena_off = 0
            for n in self.nodes:
                if n["kind"] in ("conv", "fc"):
                    for key in ("w", "bias"):
                        p = n[key]
                        if flat_grads:
                            dp = self._flat_at(self.grad_arena, arena_off, *p.dims)
                            n.setdefault("arena", [arena_off, arena_off])
                            arena_off += (int(np.prod(p.dims)) + 31) // 32 * 32
                            n["arena"][1] = arena_off
                        else:
                            dp = lib.tensor(nnc.tensor_param(memory, nnc.NHWC, F, p.dims, device))
                        mom = lib.tensor(nnc.tensor_param(memory, nnc.NHWC, F, p.dims, device))
                        n["d" + key], n["m" + key] = dp, mom
                        self.params.append((p, dp, mom))
            # momentum starts at zero (the reference zeroes saved_aux with CMD_SET_FORWARD(0), ccv_cnnp_model.c:1360-1375)
            zero = nnc.CMD_SET_FORWARD(0)
            for _, dp, mom in self.params:
                self._exec(zero, nnc.NO_HINT, 0, [], [mom], None, "set0")
        self.fwd_flops = sum(n.get("flops", 0) for n in self.nodes)
        self.step_flops = sum(n.get("flops", 0) * (3 if not n.get("first") else 2) for n in self.nodes if "flops" in n)

    def _flat(self, t, *dims):
        return self._flat_at(t, 0, *dims)

    def _flat_at(self, t, offset, *dims):
        return t.alias(dims, offset)

    def _param(self, rng, dims, fan_in, init):
        # weights ~ U(-1, 1) * sqrt(6 / fan_in) (keeps activations O(1) through 16 ReLU layers); biases small positive
        if init == "hash":  # the counter hash of tools/host_vgg_bench.c: parameter tensor number = stream id, same float arithmetic
            u = hash_unit(int(np.prod(dims)), self._hash_stream).reshape(dims)
            self._hash_stream += 1
            if fan_in:
                arr = ((u - np.float32(0.5)) * np.float32(2) * np.float32(np.sqrt(np.float32(6.0) / np.float32(fan_in)))).astype(np.float32)
            else:
                arr = (u * np.float32(0.01)).astype(np.float32)
        elif fan_in:
            arr = ((rng.random(dims, dtype=np.float32) - 0.5) * 2 * np.sqrt(6.0 / fan_in)).astype(np.float32)
        else:
            arr = (rng.random(dims, dtype=np.float32) * 0.01).astype(np.float32)
        return self.lib.tensor(nnc.tensor_param(self.memory, nnc.NHWC, nnc.CCV_32F, dims, self.device), arr)

    def set_input(self, images, labels):
        if self.memory == nnc.GPU_MEMORY:
            self.x.upload(images)
            self.label.upload(np.asarray(labels, dtype=np.float32))
        else:
            self.x.array[...] = images
            self.label.array[...] = np.asarray(labels, dtype=np.float32)

    def _exec(self, cmd, hint, flags, ins, outs, stream, tag=None, hook=None):
        c = cmd
        if self.backend is not None:
            c = nnc.Cmd()
            nnc.C.memmove(nnc.C.byref(c), nnc.C.byref(cmd), nnc.C.sizeof(c))
            c.backend = self.backend
            if self.conv_fwd_backend is not None and cmd.cmd == nnc.CMD["CONVOLUTION_FORWARD"]:
                c.backend = self.conv_fwd_backend
        if hook:
            hook("begin", tag)
        r = self.lib.cmd_exec(c, hint, flags, ins, outs, stream)
        if hook:
            hook("end", tag)
        if r != 0:
            raise RuntimeError("command %s failed with %d" % (tag, r))

    def _img(self, t, i):
        d = t.dims[1:]
        return t.alias(d, i * int(np.prod(d)))

    def _pool(self, cmd, hint, ins, outs, stream, tag, hook):
        return self._exec(cmd, hint, 0, ins, outs, stream, tag, hook)

    def forward(self, stream=None, hook=None):
        relu = nnc.CMD_RELU_FORWARD()
        for i, n in enumerate(self.nodes):
            if n["kind"] == "pool":
                self._pool(n["cmd"], n["hint"], [n["a"]], [n["b"]], stream, "pool_fwd/%d" % i, hook)
            else:
                fused = self.fuse_relu and n["kind"] == "conv" and n["relu"]
                cmd = n["cmd"]
                if fused:
                    cmd = nnc.Cmd(); nnc.C.memmove(nnc.C.byref(cmd), nnc.C.byref(n["cmd"]), nnc.C.sizeof(cmd))
                    cmd.algorithm = nnc.CONV_ALGO_FUSE_RELU | (0xff if n["cmd"].algorithm < 0 else n["cmd"].algorithm)
                self._exec(cmd, n["hint"], 0, [n["a"], n["w"], n["bias"]], [n["b"]], stream, "%s_fwd/%d" % (n["kind"], i), hook)
                if n["relu"] and not fused:
                    self._exec(relu, nnc.NO_HINT, 0, [n["b"]], [n["b"]], stream, "relu_fwd/%d" % i, hook)
        self._exec(nnc.CMD_SOFTMAX_CROSSENTROPY_FORWARD(), nnc.NO_HINT, 0, [self.logits, self.label], [self.loss, self.softmax], stream, "softmax_ce_fwd", hook)

    def backward(self, stream=None, hook=None, after_node=None):
        """after_node(i): called once node i's backward command has been enqueued (its parameter gradients are then ordered on
        `stream`) -- the data-parallel exchange hangs its bucketed all-reduce on this (ccv_amd/comm.py)."""
        relub = nnc.CMD_RELU_BACKWARD()
        g_logits = self.grads[id(self.logits)]
        self._exec(nnc.CMD_SOFTMAX_CROSSENTROPY_BACKWARD(), nnc.NO_HINT, 0, [None, None, None, self.label, None, self.softmax], [g_logits], stream, "softmax_ce_bwd", hook)
        masked = set()  # fuse_relu: gradients whose producer already applied the ReLU backward of the map they belong to
        for i in range(len(self.nodes) - 1, -1, -1):
            n = self.nodes[i]
            gb = self.grads[id(n["b"])]
            if n["kind"] == "pool":
                cmd = n["bcmd"]
                prev = self.nodes[i - 1] if i > 0 else None
                if self.fuse_relu and prev is not None and prev["kind"] == "conv" and prev["relu"] and prev["b"] is n["a"]:
                    # the pooled map IS the ReLU's output: max-pool backward masks by it as it writes (NNC_MI355X_POOL_ALGO_FUSE_RELU_BACKWARD)
                    cmd = nnc.Cmd(); nnc.C.memmove(nnc.C.byref(cmd), nnc.C.byref(n["bcmd"]), nnc.C.sizeof(cmd))
                    cmd.algorithm = nnc.POOL_ALGO_FUSE_RELU_BACKWARD
                    masked.add(id(n["a"]))
                self._pool(cmd, n["hint"], [gb, n["a"], n["b"]], [self.grads[id(n["a"])]], stream, "pool_bwd/%d" % i, hook)
                continue
            if n["relu"] and id(n["b"]) not in masked:
                self._exec(relub, nnc.NO_HINT, 0, [gb, None, n["b"]], [gb], stream, "relu_bwd/%d" % i, hook)
            h = None if n["first"] else self.grads[id(n["a"])]
            cmd = n["bcmd"]
            prev = self.nodes[i - 1] if i > 0 else None
            if self.fuse_relu and h is not None and n["kind"] == "conv" and prev["kind"] == "conv" and prev["relu"] and prev["b"] is n["a"]:
                # the convolution's input IS the previous ReLU's output: the data gradient is masked by it as it is written
                cmd = nnc.Cmd(); nnc.C.memmove(nnc.C.byref(cmd), nnc.C.byref(n["bcmd"]), nnc.C.sizeof(cmd))
                cmd.algorithm = nnc.CONV_ALGO_FUSE_RELU | (0xff if n["bcmd"].algorithm < 0 else n["bcmd"].algorithm)
                masked.add(id(n["a"]))
            self._exec(cmd, n["hint"], 0, [gb, n["a"], n["w"]], [h, n["dw"], n["dbias"]], stream, "%s_bwd/%d" % (n["kind"], i), hook)
            if after_node is not None:
                after_node(i)

    def update(self, stream=None, hook=None):
        for j, (p, dp, mom) in enumerate(self.params):
            self._exec(self.sgd_cmd, nnc.NO_HINT, 0, [dp, p, mom], [p, mom], stream, "sgd/%d" % j, hook)

    def step(self, stream=None, hook=None):
        self.forward(stream, hook)
        self.backward(stream, hook)
        self.update(stream, hook)
