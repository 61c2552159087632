"""TEST INFRASTRUCTURE -- the checker for CCV_NNC_LSTM_FORWARD / BACKWARD (ccv_amd/csrc/cmd_lstm.cpp); never imported by the product path.

PARITY UNPINNED.  The reference has no CPU implementation of this command (lib/nnc/cmd/rnn/ccv_nnc_lstm_cpu_ref.c is an empty file) and its
tests hold no values for it (test/int/nnc/lstm.tests.c runs the command and asserts nothing): the algorithm lives in a third-party dependency
that is not in /root/reference -- cuDNN (>= 8.1, the _v8 RNN API; lib/nnc/cmd/rnn/gpu/ccv_nnc_lstm_gpu_cudnn.cu:10) -- which the reference calls
as cudnnRNNForward / cudnnRNNBackwardData_v8 / cudnnRNNBackwardWeights_v8 with CUDNN_LSTM, CUDNN_LINEAR_INPUT, CUDNN_RNN_DOUBLE_BIAS (or NO_BIAS),
CUDNN_UNIDIRECTIONAL / CUDNN_BIDIRECTIONAL, an optional recurrent projection and CUDNN_RNN_PADDED_IO_ENABLED (:29, :109, :173).  This file restates
cuDNN's published LSTM in float64:

    i = sigmoid(W_i x + R_i h' + bW_i + bR_i)      f = sigmoid(W_f x + R_f h' + bW_f + bR_f)
    g = tanh   (W_g x + R_g h' + bW_g + bR_g)      o = sigmoid(W_o x + R_o h' + bW_o + bR_o)
    c = f * c' + i * g          h = o * tanh(c)     with a projection: h = W_p (o * tanh(c))

(linear-layer ids 0..3 = input matrices of i, f, g, o; 4..7 = recurrent ones; 8 = projection), the weight space packed as every pseudo-layer's
matrices (layer-major, forward direction then backward) followed by every pseudo-layer's two bias sets -- the sizes the reference's own test helper
computes (lstm.tests.c:14-21) --, a layer's input = the previous layer's outputs of both directions side by side, dropout between layers, padded
sequences (zeros in y past an item's end, final states taken at each item's own last step).  What anchors it instead of the reference's golden vectors
(there are none): (1) the gradients returned by `backward` equal central differences of `forward`; (2) outputs, final states and every gradient equal
torch.nn.LSTM's on the CPU in float64 to 1e-10 -- an independent implementation that takes cuDNN's own parameters (weight_ih / weight_hh / weight_hr / bias_ih /
bias_hh per layer and direction, gates i f g o; its CUDA path hands exactly these to cudnnRNNForward) -- with layers, both directions, the projection and
packed sequences (tests/lstm_torch_check.py, run by tests/test_lstm.py::test_oracle_matches_an_independent_lstm).  Still "unpinned" in the strict sense: neither is the reference itself."""
import numpy as np


def weight_count(I, H, P, L, D, bias):
    proj = P != H
    n = 0
    for l in range(L):
        inp = I if l == 0 else D * P
        n += D * (4 * H * inp + 4 * H * P + (P * H if proj else 0))
    return n + (L * D * 8 * H if bias else 0)


def unpack(w, I, H, P, L, D, bias):
    """-> per pseudo-layer dict of views into the flat weight vector (W [4H, in], R [4H, P], Wp [P, H] or None, bW [4H], bR [4H] or None)"""
    proj = P != H
    out, at = [], 0
    for l in range(L):
        inp = I if l == 0 else D * P
        for d in range(D):
            W = w[at:at + 4 * H * inp].reshape(4 * H, inp); at += 4 * H * inp
            R = w[at:at + 4 * H * P].reshape(4 * H, P); at += 4 * H * P
            Wp = None
            if proj:
                Wp = w[at:at + P * H].reshape(P, H); at += P * H
            out.append(dict(W=W, R=R, Wp=Wp, bW=None, bR=None))
    if bias:
        for q in out:
            q["bW"] = w[at:at + 4 * H]; at += 4 * H
            q["bR"] = w[at:at + 4 * H]; at += 4 * H
    return out


def _sig(x):
    return 1.0 / (1.0 + np.exp(-x))


def forward(x, w, H, P=0, L=1, bias=True, bidirectional=False, hx=None, cx=None, lens=None, masks=None):
    """x [T, B, I] (sequence-major), w flat.  masks: per layer l < L - 1 an array [T, B, D * P] of dropout scales (0 or 1 / (1 - p)), or None.
    -> y [T, B, D * P], hy [L * D, B, P], cy [L * D, B, H], tape (what backward needs)"""
    x = np.asarray(x, np.float64)
    w = np.asarray(w, np.float64)
    T, B, I = x.shape
    P = P or H
    D = 2 if bidirectional else 1
    lens = np.full(B, T, np.int64) if lens is None else np.asarray(lens, np.int64)
    parts = unpack(w, I, H, P, L, D, bias)
    hy = np.zeros((L * D, B, P))
    cy = np.zeros((L * D, B, H))
    tape = []
    inp = x
    for l in range(L):
        out = np.zeros((T, B, D * P))
        for d in range(D):
            p = l * D + d
            q = parts[p]
            h = np.zeros((B, P)) if hx is None else np.array(hx[p], np.float64)
            c = np.zeros((B, H)) if cx is None else np.array(cx[p], np.float64)
            steps = []
            for s in range(T):
                t = T - 1 - s if d else s
                valid = (t < lens)[:, None]
                a = inp[t] @ q["W"].T + h @ q["R"].T
                if bias:
                    a = a + q["bW"] + q["bR"]
                i, f, g, o = _sig(a[:, :H]), _sig(a[:, H:2 * H]), np.tanh(a[:, 2 * H:3 * H]), _sig(a[:, 3 * H:])
                cn = f * c + i * g
                tc = np.tanh(cn)
                raw = o * tc
                hn = raw @ q["Wp"].T if q["Wp"] is not None else raw
                steps.append(dict(t=t, valid=valid, i=i, f=f, g=g, o=o, tc=tc, raw=raw, hprev=h, cprev=c, x=inp[t]))
                out[t, :, d * P:(d + 1) * P] = np.where(valid, hn, 0.0)
                h = np.where(valid, hn, h)
                c = np.where(valid, cn, c)
            hy[p], cy[p] = h, c
            tape.append(steps)
        if l < L - 1 and masks is not None and masks[l] is not None:
            out = out * masks[l]
        inp = out
    return inp, hy, cy, dict(tape=tape, parts=parts, dims=(T, B, I, H, P, L, D, bias), lens=lens, masks=masks)


def backward(dy, tape, dhy=None, dcy=None):
    """-> dx [T, B, I], dhx [L * D, B, P], dcx [L * D, B, H], dw flat (the same packing as w)"""
    T, B, I, H, P, L, D, bias = tape["dims"]
    parts = tape["parts"]
    masks = tape["masks"]
    dparts = [dict(W=np.zeros_like(q["W"]), R=np.zeros_like(q["R"]), Wp=None if q["Wp"] is None else np.zeros_like(q["Wp"]), b=np.zeros(4 * H)) for q in parts]
    dhx = np.zeros((L * D, B, P))
    dcx = np.zeros((L * D, B, H))
    dout = np.asarray(dy, np.float64)
    for l in range(L - 1, -1, -1):
        inp = I if l == 0 else D * P
        din = np.zeros((T, B, inp))
        for d in range(D):
            p = l * D + d
            q, dq = parts[p], dparts[p]
            dh = np.zeros((B, P)) if dhy is None else np.array(dhy[p], np.float64)
            dc = np.zeros((B, H)) if dcy is None else np.array(dcy[p], np.float64)
            for st in reversed(tape["tape"][p]):
                t, valid = st["t"], st["valid"]
                dhn = np.where(valid, dh + dout[t, :, d * P:(d + 1) * P], 0.0)
                if q["Wp"] is not None:
                    dq["Wp"] += dhn.T @ st["raw"]
                    draw = dhn @ q["Wp"]
                else:
                    draw = dhn
                dct = np.where(valid, dc, 0.0) + draw * st["o"] * (1 - st["tc"] ** 2)
                da = np.concatenate([dct * st["g"] * st["i"] * (1 - st["i"]), dct * st["cprev"] * st["f"] * (1 - st["f"]),
                                     dct * st["i"] * (1 - st["g"] ** 2), draw * st["tc"] * st["o"] * (1 - st["o"])], axis=1)
                da = np.where(valid, da, 0.0)
                dq["W"] += da.T @ st["x"]
                dq["R"] += da.T @ st["hprev"]
                dq["b"] += da.sum(0)
                din[t] += da @ q["W"]
                dh = np.where(valid, da @ q["R"], dh)
                dc = np.where(valid, dct * st["f"], dc)
            dhx[p], dcx[p] = dh, dc
        if l > 0 and masks is not None and masks[l - 1] is not None:
            din = din * masks[l - 1]
        dout = din
    flat = []
    for dq in dparts:
        flat += [dq["W"].ravel(), dq["R"].ravel()] + ([dq["Wp"].ravel()] if dq["Wp"] is not None else [])
    if bias:
        for dq in dparts:
            flat += [dq["b"], dq["b"]]
    return dout, dhx, dcx, np.concatenate(flat)
