"""oracle/lstm_numpy.py against torch.nn.LSTM on the CPU (float64) -- an independent implementation whose parameters are cuDNN's (weight_ih / weight_hh /
weight_hr / bias_ih / bias_hh per layer and direction, gates i f g o; its CUDA path hands exactly these to cudnnRNNForward), packed sequences for the
per-item lengths: outputs, final states and every gradient to 1e-10.  The reference holds no LSTM values to pin the oracle to; this is the anchor there is.
Run by tests/test_lstm.py::test_oracle_matches_an_independent_lstm, one case per process: python tests/lstm_torch_check.py <case>   (exit 77: no torch)"""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import lstm_numpy as oracle  # noqa: E402

try:
    import torch
except Exception:  # noqa: BLE001
    sys.exit(77)

CASES = [
    # T, B, I, H, P, L, bidirectional, lens
    (5, 3, 7, 8, 0, 2, False, None),
    (4, 3, 6, 8, 0, 2, True, [4, 2, 3]),
    (5, 2, 6, 12, 5, 2, True, [3, 5]),      # recurrent projection (weight_hr), both directions, own lengths
    (3, 4, 5, 6, 0, 1, False, [3, 1, 2, 3]),
]
T, B, I, H, P, L, bidir, lens = CASES[int(sys.argv[1])]
D, Pe = (2 if bidir else 1), (P or H)
torch.manual_seed(7)
net = torch.nn.LSTM(I, H, num_layers=L, bias=True, batch_first=False, bidirectional=bidir, proj_size=P).double()
mats, biases = [], []
for l in range(L):
    for sfx in ([""] + (["_reverse"] if bidir else [])):
        names = ["weight_ih_l%d%s" % (l, sfx), "weight_hh_l%d%s" % (l, sfx)] + (["weight_hr_l%d%s" % (l, sfx)] if P else [])
        mats += [getattr(net, n) for n in names]
        biases += [getattr(net, "bias_ih_l%d%s" % (l, sfx)), getattr(net, "bias_hh_l%d%s" % (l, sfx))]
w = np.concatenate([p.detach().numpy().ravel() for p in mats + biases])
assert w.size == oracle.weight_count(I, H, Pe, L, D, True)
rng = np.random.default_rng(5)
x, hx, cx = rng.standard_normal((T, B, I)), rng.standard_normal((L * D, B, Pe)) * 0.5, rng.standard_normal((L * D, B, H)) * 0.5
gy, ghy, gcy = rng.standard_normal((T, B, D * Pe)), rng.standard_normal((L * D, B, Pe)), rng.standard_normal((L * D, B, H))
xt, hxt, cxt = (torch.tensor(a, requires_grad=True) for a in (x, hx, cx))
if lens is None:
    yt, (hyt, cyt) = net(xt, (hxt, cxt))
else:
    packed = torch.nn.utils.rnn.pack_padded_sequence(xt, torch.tensor(lens), enforce_sorted=False)
    out, (hyt, cyt) = net(packed, (hxt, cxt))
    yt, _ = torch.nn.utils.rnn.pad_packed_sequence(out, total_length=T)
((yt * torch.tensor(gy)).sum() + (hyt * torch.tensor(ghy)).sum() + (cyt * torch.tensor(gcy)).sum()).backward()
y, hy, cy, tape = oracle.forward(x, w, H, Pe, L, True, bidir, hx, cx, lens)
dx, dhx, dcx, dw = oracle.backward(gy, tape, ghy, gcy)
for got, want in ((y, yt), (hy, hyt), (cy, cyt), (dx, xt.grad), (dhx, hxt.grad), (dcx, cxt.grad)):
    np.testing.assert_allclose(got, want.detach().numpy(), atol=1e-10)
np.testing.assert_allclose(dw, np.concatenate([p.grad.numpy().ravel() for p in mats + biases]), atol=1e-10)
print("ok")
