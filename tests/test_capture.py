"""HIP-graph capture of a compiled schedule (ccv_amd/csrc/device_rt.cpp "HIP-graph capture", include/nnc_mi355x.h): the host records
ONE step -- enqueue-only calls on its stream; the schedule's other streams join through the signals it emits and waits for, as in the reference's run loop
(lib/nnc/ccv_nnc_graph_run.c:581-675, :707-726 fork, :819-839 join) -- and replays it with one runtime call per step.  What is checked here, on the CPU emulator
(whose capturing streams RECORD instead of running, refuse what the real runtime refuses during a capture, and replay the recorded nodes) and on the MI355X:
  * a step of convolution + in-place ReLU (the look-ahead's fold), pooling, cluster batch norm with running statistics, filter gradient and momentum SGD over
    three streams: warm-up + capture + 2 replays == 3 steps issued directly, bit for bit -- every stateful tensor included;
  * nothing executes while the step is recorded;
  * DROPOUT draws a fresh mask per replay (the seed the host drew during the recording is a kernel argument);
  * device memory freed during the capture / while the graph lives is set aside, and comes back with nnc_mi355x_graph_free;
  * a scratch buffer may grow inside the capture (no wait for the recording stream);
  * a step that does not join its streams back is refused, and the library goes on working."""
import numpy as np
import pytest
from ccv_amd import nnc
from harness import make_tensors

F = np.float32


@pytest.fixture
def lib(backend):
    return backend


class Step:
    """The tensors and streams of a small training step, and the step itself as the reference's schedule would issue it."""

    def __init__(self, lib, seed=5, n=4, hw=12, c=8, k=16):
        rng = np.random.default_rng(seed)
        self.lib, self.n, self.c, self.k = lib, n, c, k
        r = lambda *s: rng.standard_normal(s).astype(F)
        self.hint = nnc.Hint()
        for ax in range(2):
            self.hint.stride.dim[ax] = 1
            self.hint.border.begin[ax] = 1
            self.hint.border.end[ax] = 1
        self.phint = nnc.Hint()
        for ax in range(2):
            self.phint.stride.dim[ax] = 2
        G = nnc.GPU_MEMORY
        (self.x, self.w, self.b, self.y, self.gy, self.h, self.dw, self.db, self.mw, self.mb) = make_tensors(lib, G, [
            r(n, hw, hw, c), 0.2 * r(k, 3, 3, c), 0.1 * r(k), np.zeros((n, hw, hw, k), F), r(n, hw, hw, k), np.zeros((n, hw, hw, c), F),
            np.zeros((k, 3, 3, c), F), np.zeros(k, F), np.zeros((k, 3, 3, c), F), np.zeros(k, F)])
        (self.p,) = make_tensors(lib, G, [np.zeros((n, hw // 2, hw // 2, k), F)])
        # batch norm on an NCHW tensor, cluster kernels (several workgroups per channel: the hand-over words and their epochs are exercised)
        self.z, self.zy = make_tensors(lib, G, [r(6, 5, 14, 14), np.zeros((6, 5, 14, 14), F)], "NCHW")
        self.scale, self.bias, self.mean, self.var, self.smean, self.sistd = make_tensors(lib, G, [rng.random(5, dtype=F) + F(1), r(5), r(5), rng.random(5, dtype=F) + F(0.5), np.zeros(5, F), np.zeros(5, F)], "NCHW")
        self.A, self.X, self.Y = lib.stream_new(0), lib.stream_new(0), lib.stream_new(0)
        self.S, self.Sx, self.Sy = lib.signal_new(0), lib.signal_new(0), lib.signal_new(0)

    def issue(self):
        lib, A, X, Y = self.lib, self.A, self.X, self.Y
        conv, relu = nnc.CMD_CONVOLUTION_FORWARD(1, self.k, 3, 3, self.c), nnc.CMD_RELU_FORWARD()
        back = nnc.CMD_CONVOLUTION_BACKWARD(1, self.k, 3, 3, self.c)
        sgd = nnc.CMD_SGD_FORWARD(0, 0.01, 0.5, 0.0005, 0.9, 0.9)
        bn = nnc.CMD_BATCH_NORM_FORWARD(1e-4, 0, 0.9, 0, 2, 3)
        assert lib.cmd_exec(conv, self.hint, 0, [self.x, self.w, self.b], [self.y], A) == 0
        lib.signal_emit(A, self.S)                  # stream Y forks here: the batch norm runs next to the rest of the step
        lib.signal_wait(Y, self.S)
        assert lib.cmd_exec(bn, nnc.NO_HINT, 0, [self.z, self.scale, self.bias, self.mean, self.var], [self.zy, self.mean, self.var, self.smean, self.sistd], Y) == 0
        lib.signal_emit(Y, self.Sy)
        assert lib.cmd_exec(relu, nnc.NO_HINT, 0, [self.y], [self.y], A) == 0   # in place behind the convolution: folded by the look-ahead
        assert lib.cmd_exec(nnc.CMD_MAX_POOL_FORWARD(2, 2), self.phint, 0, [self.y], [self.p], A) == 0
        assert lib.cmd_exec(back, self.hint, 0, [self.gy, self.x, self.w], [self.h, self.dw, self.db], A) == 0
        lib.signal_emit(A, self.S)                  # (the same signal again, as the schedule reuses its signals from node to node)
        lib.signal_wait(X, self.S)
        assert lib.cmd_exec(sgd, nnc.NO_HINT, 0, [self.dw, self.w, self.mw], [self.w, self.mw], X) == 0
        assert lib.cmd_exec(sgd, nnc.NO_HINT, 0, [self.db, self.b, self.mb], [self.b, self.mb], X) == 0
        lib.signal_emit(X, self.Sx)
        lib.signal_wait(A, self.Sx)                 # both side streams join back
        lib.signal_wait(A, self.Sy)

    def state(self):
        self.lib.stream_wait(self.A)
        return [t.numpy().copy() for t in (self.y, self.p, self.h, self.dw, self.db, self.w, self.b, self.mw, self.mb, self.zy, self.mean, self.var, self.smean, self.sistd)]

    def close(self):
        for s in (self.A, self.X, self.Y):
            self.lib.stream_free(s)
        for s in (self.S, self.Sx, self.Sy):
            self.lib.signal_free(s)


NAMES = ("y", "pool", "h", "dw", "db", "w", "bias", "momentum w", "momentum b", "bn y", "running mean", "running var", "saved mean", "saved inv std")


@pytest.mark.parametrize("form", ["folded", "streams"])
def test_a_captured_step_replayed_equals_the_step_issued_directly(lib, form):
    """Both forms of a capture (device_rt.cpp): the step's side streams folded into the recording stream (default), or kept as HIP streams that join it."""
    lib.capture_keep_streams(form == "streams")
    old = lib.tune_get("BN_CLUSTER")
    lib.tune_set("BN_CLUSTER", 16)  # chunks per workgroup forced down: several workgroups per channel at this size
    n0 = lib.dll.nnc_mi355x_debug_bn_cluster_launches()
    try:
        direct = Step(lib)
        for _ in range(3):
            direct.issue()
        want = direct.state()
        direct.close()
        assert lib.dll.nnc_mi355x_debug_bn_cluster_launches() == n0 + 3  # (the batch norm of the step is a cluster launch)
        cap = Step(lib)
        cap.issue()                                  # warm-up: scratch buffers, hand-over areas, autotune
        after_one = cap.state()
        assert lib.capture_begin(cap.A) == 0
        cap.issue()
        graph = lib.capture_end(cap.A)
        assert graph, "the capture was refused"
        assert lib.graph_node_count(graph) >= 8      # tick + convolution (+ ReLU folded) + pool + gradient kernels + 2 updates + area clear + batch norm
        for a, b, name in zip(cap.state(), after_one, NAMES):
            assert np.array_equal(a, b), "recording the step executed something: " + name
        for _ in range(2):
            assert lib.graph_launch(graph, cap.A) == 0
        got = cap.state()
        lib.graph_free(graph)
        cap.close()
    finally:
        lib.tune_set("BN_CLUSTER", old)
        lib.capture_keep_streams(0)
    for a, b, name in zip(got, want, NAMES):
        assert np.array_equal(a, b), name
    assert not np.array_equal(got[5], after_one[5])  # (the parameters did move with every replay)


def test_dropout_draws_a_fresh_mask_per_replay(lib):
    n = 1 << 14
    (a, b, mask) = make_tensors(lib, nnc.GPU_MEMORY, [np.ones(n, F), np.zeros(n, F), np.zeros(n // 4, F)])  # (the mask: a byte per element in an opaque fp32 tensor, as the reference sizes it)
    s = lib.stream_new(0)
    try:
        cmd = nnc.CMD_DROPOUT_FORWARD(0.5)
        assert lib.cmd_exec(cmd, nnc.NO_HINT, 0, [a], [b, mask], s) == 0
        lib.stream_wait(s)
        assert lib.capture_begin(s) == 0
        assert lib.cmd_exec(cmd, nnc.NO_HINT, 0, [a], [b, mask], s) == 0
        graph = lib.capture_end(s)
        assert graph
        masks = []
        for _ in range(3):
            assert lib.graph_launch(graph, s) == 0
            lib.stream_wait(s)
            m, out = mask.numpy().view(np.uint8).copy(), b.numpy()
            assert np.array_equal(out, np.where(m != 0, F(0), F(2)))
            assert 0.45 < m.mean() < 0.55
            masks.append(m)
        lib.graph_free(graph)
        assert not np.array_equal(masks[0], masks[1]) and not np.array_equal(masks[1], masks[2]) and not np.array_equal(masks[0], masks[2])
    finally:
        lib.stream_free(s)


def test_memory_freed_under_a_capture_is_set_aside_until_the_graph_is_freed(lib):
    n = 3 << 20  # a size nothing else in the suite uses
    s = lib.stream_new(0)
    try:
        (old,) = make_tensors(lib, nnc.GPU_MEMORY, [np.zeros(n, F)])     # allocated before the capture, named by it, freed while the graph lives
        (t,) = make_tensors(lib, nnc.GPU_MEMORY, [np.zeros(n, F)])
        assert lib.cmd_exec(nnc.CMD_SET_FORWARD(1.0), nnc.NO_HINT, 0, [], [t], s) == 0
        lib.stream_wait(s)
        parked0 = lib.pool_parked_bytes()
        assert lib.capture_begin(s) == 0
        from harness import _param
        tmp = lib.tensor(_param(nnc.GPU_MEMORY, "NHWC", np.zeros(n, F)))  # allocated AND freed inside the capture (no upload: a blocking copy is NULL-stream work, which the runtime refuses while a stream records): the recorded kernel still names it
        assert lib.cmd_exec(nnc.CMD_SET_FORWARD(3.0), nnc.NO_HINT, 0, [], [tmp], s) == 0
        assert lib.cmd_exec(nnc.CMD_SET_FORWARD(4.0), nnc.NO_HINT, 0, [], [old], s) == 0
        assert lib.cmd_exec(nnc.CMD_SET_FORWARD(2.0), nnc.NO_HINT, 0, [], [t], s) == 0
        tmp_ptr = tmp.ptr
        tmp.free()
        assert lib.pool_parked_bytes() - parked0 >= 4 * n
        graph = lib.capture_end(s)
        assert graph
        old_ptr = old.ptr
        old.free()                                                        # after the capture, the graph alive: set aside as well
        assert lib.pool_parked_bytes() - parked0 >= 8 * n
        (other,) = make_tensors(lib, nnc.GPU_MEMORY, [np.full(n, 7, F)])  # the same size: must be a block of its own
        assert other.ptr not in (tmp_ptr, old_ptr)
        assert lib.graph_launch(graph, s) == 0                            # writes 3 into tmp's block, 4 into old's
        lib.stream_wait(s)
        assert (other.numpy() == F(7)).all() and (t.numpy() == F(2)).all()
        (late,) = make_tensors(lib, nnc.GPU_MEMORY, [np.zeros(16, F)])    # allocated after the capture ended: no graph can name it
        late.free()
        assert lib.pool_parked_bytes() - parked0 == 8 * n
        lib.graph_free(graph)
        assert lib.pool_parked_bytes() == parked0
        (again,) = make_tensors(lib, nnc.GPU_MEMORY, [np.zeros(n, F)])   # ... and the blocks are in circulation again
        assert again.ptr in (tmp_ptr, old_ptr)
        for x in (t, other, again):
            x.free()
    finally:
        lib.stream_free(s)


def test_a_scratch_buffer_may_grow_inside_a_capture(lib):
    """The convolution's scratch request is served by the stream's grow-only workspace; growing it outside a capture waits for the stream (queued kernels may
    still read the old buffer) -- a recording stream cannot be waited for and has nothing queued: the old buffer is set aside instead."""
    rng = np.random.default_rng(9)
    s = lib.stream_new(0)
    hint = nnc.Hint()
    for ax in range(2):
        hint.stride.dim[ax] = 1
        hint.border.begin[ax] = 1
        hint.border.end[ax] = 1

    def conv(n, hw, c, k):
        x, w, b = rng.standard_normal((n, hw, hw, c)).astype(F), (0.1 * rng.standard_normal((k, 3, 3, c))).astype(F), np.zeros(k, F)
        ts = make_tensors(lib, nnc.GPU_MEMORY, [x, w, b, np.zeros((n, hw, hw, k), F)])
        return ts, x, w

    try:
        small, _, _ = conv(1, 8, 8, 8)
        cmd = nnc.CMD_CONVOLUTION_FORWARD(1, 8, 3, 3, 8)
        cmd.algorithm = 1  # Winograd through HBM: transformed tiles in the workspace
        assert lib.cmd_exec(cmd, hint, 0, small[:3], small[3:], s) == 0
        lib.stream_wait(s)
        big, x, w = conv(2, 24, 32, 32)
        cmdb = nnc.CMD_CONVOLUTION_FORWARD(1, 32, 3, 3, 32)
        cmdb.algorithm = 1
        assert lib.capture_begin(s) == 0
        assert lib.cmd_exec(cmdb, hint, 0, big[:3], big[3:], s) == 0
        graph = lib.capture_end(s)
        assert graph
        assert lib.graph_launch(graph, s) == 0
        lib.stream_wait(s)
        got = big[3].numpy()
        lib.graph_free(graph)
        # against the same command issued directly
        (out2,) = make_tensors(lib, nnc.GPU_MEMORY, [np.zeros_like(got)])
        assert lib.cmd_exec(cmdb, hint, 0, big[:3], [out2], s) == 0
        lib.stream_wait(s)
        assert np.array_equal(got, out2.numpy()) and np.abs(got).max() > 0
    finally:
        lib.stream_free(s)


def test_a_step_that_leaves_a_stream_unjoined_is_refused(emu_lib, capfd):
    """(Emulator only: on the MI355X the runtime leaves the unjoined stream in its recording state after the refused hipStreamEndCapture -- it can be neither
    synchronised nor used again; the first GPU run of this test stopped in that stream's hipStreamSynchronize.)"""
    lib = emu_lib
    (t, u) = make_tensors(lib, nnc.GPU_MEMORY, [np.zeros(256, F), np.zeros(256, F)])
    A, B = lib.stream_new(0), lib.stream_new(0)
    S = lib.signal_new(0)
    lib.capture_keep_streams(1)                                # (in the folded form B's work is on the recording stream: nothing to join)
    try:
        assert lib.capture_begin(A) == 0
        assert lib.capture_begin(B) == -1                      # one capture at a time
        lib.signal_emit(A, S)
        lib.signal_wait(B, S)
        assert lib.cmd_exec(nnc.CMD_SET_FORWARD(1.0), nnc.NO_HINT, 0, [], [u], B) == 0   # ... and nobody waits for B
        assert lib.capture_end(A) is None
        assert "capture_end" in capfd.readouterr().err
        # the library goes on: a well-formed capture right behind it
        assert lib.capture_begin(A) == 0
        assert lib.cmd_exec(nnc.CMD_SET_FORWARD(5.0), nnc.NO_HINT, 0, [], [t], A) == 0
        graph = lib.capture_end(A)
        assert graph
        assert (t.numpy() == 0).all()
        assert lib.graph_launch(graph, A) == 0
        lib.stream_wait(A)
        assert (t.numpy() == F(5)).all()
        lib.graph_free(graph)
    finally:
        lib.capture_keep_streams(0)
        lib.stream_free(A)
        lib.stream_free(B)
        lib.signal_free(S)


def test_side_streams_that_wait_for_each_other_are_recorded_in_the_folded_form(lib):
    """The reference's schedules run on the graph's OWN streams, none of which is the caller's (recording) stream: its stream 0 and the side streams wait for
    each other's signals.  ROCm 7.2's hipStreamEndCapture recurses without end on that pattern (device_rt.cpp "HIP-graph capture"; the emulator models the
    runtime's bookkeeping and refuses), so the default form folds the step's streams into the recording one.  Origin O; M and N wait for each other."""
    (t, u) = make_tensors(lib, nnc.GPU_MEMORY, [np.zeros(1024, F), np.zeros(1024, F)])
    O, M, N = lib.stream_new(0), lib.stream_new(0), lib.stream_new(0)
    S0, Sm, Sn, Se = (lib.signal_new(0) for _ in range(4))

    def step():
        lib.signal_emit(O, S0)
        lib.signal_wait(M, S0)
        assert lib.cmd_exec(nnc.CMD_SET_FORWARD(2.0), nnc.NO_HINT, 0, [], [t], M) == 0
        lib.signal_emit(M, Sm)
        lib.signal_wait(N, Sm)                                                   # N waits for M ...
        assert lib.cmd_exec(nnc.CMD_SCALAR_MUL_FORWARD(3.0), nnc.NO_HINT, 0, [t], [u], N) == 0
        lib.signal_emit(N, Sn)
        lib.signal_wait(M, Sn)                                                   # ... and M for N
        assert lib.cmd_exec(nnc.CMD_SCALAR_MUL_FORWARD(5.0), nnc.NO_HINT, 0, [u], [t], M) == 0
        lib.signal_emit(M, Se)
        lib.signal_wait(O, Se)

    try:
        assert lib.capture_begin(O) == 0
        step()
        graph = lib.capture_end(O)
        assert graph and lib.graph_node_count(graph) == 4                        # tick + three kernels, one chain
        assert (t.numpy() == 0).all()
        assert lib.graph_launch(graph, O) == 0
        lib.stream_wait(O)
        assert (t.numpy() == F(30)).all() and (u.numpy() == F(6)).all()
        lib.graph_free(graph)
        step()                                                                   # and the same streams work as streams again afterwards
        lib.stream_wait(O)
        assert (t.numpy() == F(30)).all()
    finally:
        for s_ in (O, M, N):
            lib.stream_free(s_)
        for g_ in (S0, Sm, Sn, Se):
            lib.signal_free(g_)


def test_the_kept_form_is_refused_for_mutually_waiting_side_streams_on_the_emulator(emu_lib, capfd):
    """What the MI355X's runtime does with that pattern is a stack overflow inside hipStreamEndCapture; the emulator, which keeps the same books, says so."""
    lib = emu_lib
    (t,) = make_tensors(lib, nnc.GPU_MEMORY, [np.zeros(64, F)])
    O, M, N = lib.stream_new(0), lib.stream_new(0), lib.stream_new(0)
    S0, Sm, Sn, Se = (lib.signal_new(0) for _ in range(4))
    lib.capture_keep_streams(1)
    try:
        assert lib.capture_begin(O) == 0
        lib.signal_emit(O, S0); lib.signal_wait(M, S0)
        assert lib.cmd_exec(nnc.CMD_SET_FORWARD(2.0), nnc.NO_HINT, 0, [], [t], M) == 0
        lib.signal_emit(M, Sm); lib.signal_wait(N, Sm)
        lib.signal_emit(N, Sn); lib.signal_wait(M, Sn)
        lib.signal_emit(M, Se); lib.signal_wait(O, Se)
        assert lib.capture_end(O) is None
        assert "recurses without end" in capfd.readouterr().err
    finally:
        lib.capture_keep_streams(0)
        for s_ in (O, M, N):
            lib.stream_free(s_)
        for g_ in (S0, Sm, Sn, Se):
            lib.signal_free(g_)
