import ctypes as C, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from ccv_amd import nnc
L = nnc.load(); st = L.stream_new(0)
L.dll.nnc_mi355x_event_elapsed_ms.restype = C.c_float; L.dll.nnc_mi355x_event_new.restype = C.c_void_p
L.dll.nnc_mi355x_event_record.argtypes = [C.c_void_p, C.c_void_p]; L.dll.nnc_mi355x_event_elapsed_ms.argtypes = [C.c_void_p, C.c_void_p]
def rate(nbytes, fn, reps=8):
    for _ in range(3): fn()
    L.stream_wait(st)
    e0, e1 = L.dll.nnc_mi355x_event_new(), L.dll.nnc_mi355x_event_new()
    L.dll.nnc_mi355x_event_record(e0, st)
    for _ in range(reps): fn()
    L.dll.nnc_mi355x_event_record(e1, st)
    ms = L.dll.nnc_mi355x_event_elapsed_ms(e0, e1) / reps
    return ms, nbytes / ms / 1e9
n = 256 * 223 * 223 * 64
x = L.tensor(nnc.tensor_param(nnc.GPU_MEMORY, nnc.NHWC, nnc.CCV_32F, (256, 223, 223, 64), 0))
y = L.tensor(nnc.tensor_param(nnc.GPU_MEMORY, nnc.NHWC, nnc.CCV_32F, (256, 223, 223, 64), 0))
for rnd in range(3):
    print("fill (write only, %.2f GB): %.3f ms %.2f TB/s" % ((4.0 * n / 1e9,) + rate(4.0 * n, lambda: L.cmd_exec(nnc.CMD_SET_FORWARD(1.5), nnc.NO_HINT, 0, [], [x], st))))
    print("copy (read + write): %.3f ms %.2f TB/s" % rate(8.0 * n, lambda: L.cmd_exec(nnc.generic_cmd("DATA_TRANSFER_FORWARD"), nnc.NO_HINT, 0, [x], [y], st)))
