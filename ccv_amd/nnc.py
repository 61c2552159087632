"""ctypes mirror of the nnc plugin surface (struct layouts: include/nnc_mi355x.h; reference: lib/nnc/ccv_nnc.h,
lib/nnc/ccv_nnc_tfb.h) and of the command builders in lib/nnc/cmd/ccv_nnc_cmd_easy.h.

The same structs drive three libraries through the same call signature
(ccv_nnc_cmd_exec, lib/nnc/ccv_nnc_cmd.c:651):
  * libnnc_mi355x.so          -- the product (entry point nnc_mi355x_cmd_exec)
  * oracle/_ref/libccv_ref.so -- the reference's own CPU backend (tests only; entry point ccv_nnc_cmd_exec)
  * tests/emu/_build/libnnc_mi355x_emu.so -- the product sources on the CPU HIP emulator (tests only)
"""
import ctypes as C
import os
import numpy as np

MAX_DIM_ALLOC = 12

# ---- constants (include/nnc_mi355x.h) -------------------------------------------------------------------------
NCHW, NHWC, CHWN = 0x01, 0x02, 0x04
CPU_MEMORY, GPU_MEMORY = 0x1, 0x2
TENSOR_VIEW = 0x01000000
CCV_8U, CCV_32S, CCV_32F, CCV_64S, CCV_64F, CCV_16F = 0x01000, 0x02000, 0x04000, 0x08000, 0x10000, 0x20000
CCV_QX = 0x40000  # palettized: CCV_QX | qbits << 8 | palette datatype >> 12, elements per block in info.reserved (lib/nnc/ccv_nnc_easy.h:210-218)
ACCUMULATE_OUTPUT, ZERO_MEMORY_ALLOC = 0x01, 0x02
EXEC_SUCCESS, EXEC_INVALID, EXEC_NO_KERNEL, EXEC_OOM = 0, -1, -2, -3
STREAM_CONTEXT_CPU, STREAM_CONTEXT_GPU = 0x1, 0x2
NO_BACKEND = 0
BACKEND_CPU_OPT, BACKEND_CPU_REF = 0x46deb194, 0x3d9883e5
BACKEND_GPU_CUBLAS, BACKEND_GPU_CUDNN, BACKEND_GPU_NCCL, BACKEND_GPU_REF = 0x9b8cfed, 0x854b679a, 0x7afed9c7, 0x5f19790a

CMD = dict(
    SCALED_DOT_PRODUCT_ATTENTION_FORWARD=0x284ed926, SCALED_DOT_PRODUCT_ATTENTION_BACKWARD=0x284ed927, LSTM_FORWARD=0xc5cb998c, LSTM_BACKWARD=0xc5cb998d,
    CMUL_FORWARD=0xead486e6, CMUL_BACKWARD=0xead486e7, NMS_FORWARD=0xdba26106, NMS_BACKWARD=0xdba26107,
    ROI_ALIGN_FORWARD=0xfef55168, ROI_ALIGN_BACKWARD=0xfef55169, COMPRESSION_LSSC_FORWARD=0x17ea8f72, COMPRESSION_LSSC_BACKWARD=0x17ea8f73,
    ADD_FORWARD=0x58fb3664, ADD_BACKWARD=0x58fb3665, DROPOUT_FORWARD=0x7f2dc3e4, DROPOUT_BACKWARD=0x7f2dc3e5,
    AVERAGE_POOL_FORWARD=0x51267ab8, AVERAGE_POOL_BACKWARD=0x51267ab9,
    BATCH_NORM_FORWARD=0x5419819c, BATCH_NORM_BACKWARD=0x5419819d,
    CLAMP_FORWARD=0x2640d854, CLAMP_BACKWARD=0x2640d855,
    COMM_ALLREDUCE_FORWARD=0x75c8d340, COMM_BROADCAST_FORWARD=0x830eee, COMM_REDUCE_FORWARD=0x3434ead8,
    CONVOLUTION_FORWARD=0x254d05f4, CONVOLUTION_BACKWARD=0x254d05f5, CONVOLUTION_TRANSPOSE_FORWARD=0xd691f78e,
    DATATYPE_CONVERSION_FORWARD=0xd873e38c, DATA_TRANSFER_FORWARD=0x12d21e1a, DATA_TRANSFER_BACKWARD=0x12d21e1b,
    EWDIV_FORWARD=0x1cd2fa18, EWDIV_BACKWARD=0x1cd2fa19, EWEXP_FORWARD=0xd784b170, EWEXP_BACKWARD=0xd784b171,
    EWLOG_FORWARD=0xf4191bf2, EWLOG_BACKWARD=0xf4191bf3, EWPROD_FORWARD=0xee07e8fe, EWPROD_BACKWARD=0xee07e8ff,
    EWSQRT_FORWARD=0x8870a61e, EWSQRT_BACKWARD=0x8870a61f, EWSUM_FORWARD=0xe21a2c4c, EWSUM_BACKWARD=0xe21a2c4d,
    FORMAT_TRANSFORM_FORWARD=0xe4a2b192, FORMAT_TRANSFORM_BACKWARD=0xe4a2b193,
    GEMM_FORWARD=0x7e87d00c, GEMM_BACKWARD=0x7e87d00d,
    MAX_POOL_FORWARD=0x7bec9360, MAX_POOL_BACKWARD=0x7bec9361,
    MUL_FORWARD=0x24721a46, MUL_BACKWARD=0x24721a47,
    REDUCE_MEAN_FORWARD=0xf23556c6, REDUCE_MEAN_BACKWARD=0xf23556c7,
    REDUCE_SUM_FORWARD=0x52970f06, REDUCE_SUM_BACKWARD=0x52970f07,
    RELU_FORWARD=0xc51eaa80, RELU_BACKWARD=0xc51eaa81,
    SCALAR_MUL_FORWARD=0x8b4d86aa, SCALAR_MUL_BACKWARD=0x8b4d86ab,
    SET_FORWARD=0x2b070804, SET_BACKWARD=0x2b070805,
    SGD_FORWARD=0xe650ad26,
    SOFTMAX_CROSSENTROPY_FORWARD=0xc26b7b5e, SOFTMAX_CROSSENTROPY_BACKWARD=0xc26b7b5f,
    TRANSPOSE_FORWARD=0xb4d506e0, TRANSPOSE_BACKWARD=0xb4d506e1,
    # element-wise / optimizer / loss rows of SURVEY.md section 8(f).1
    RANDOM_UNIFORM_FORWARD=0xa0cd1d5e, RANDOM_NORMAL_FORWARD=0x7062c8b4,
    GROUP_NORM_FORWARD=0x17deb074, GROUP_NORM_BACKWARD=0x17deb075,
    LAYER_NORM_FORWARD=0xbed3c264, LAYER_NORM_BACKWARD=0xbed3c265, RMSNORM_FORWARD=0x6889e9d0, RMSNORM_BACKWARD=0x6889e9d1,
    ADAM_FORWARD=0xe30099dc, ADAM_BACKWARD=0xe30099dd, ADAMW_FORWARD=0x4f5d4870, ADAMW_BACKWARD=0x4f5d4871,
    ARGMAX_FORWARD=0x68af2804, ARGMAX_BACKWARD=0x68af2805, ARGMIN_FORWARD=0xeb8747f2, ARGMIN_BACKWARD=0xeb8747f3,
    BINARY_CROSSENTROPY_FORWARD=0xcd2107ec, BINARY_CROSSENTROPY_BACKWARD=0xcd2107ed,
    CATEGORICAL_CROSSENTROPY_FORWARD=0x1eb327a2, CATEGORICAL_CROSSENTROPY_BACKWARD=0x1eb327a3,
    GELU_FORWARD=0xb1527ab8, GELU_BACKWARD=0xb1527ab9, INDEX_SELECT_FORWARD=0x7ee7771e,
    INDEX_SELECT_BACKWARD=0x7ee7771f, LAMB_FORWARD=0x450edb1a, LAMB_BACKWARD=0x450edb1b,
    LEAKY_RELU_FORWARD=0x507144e0, LEAKY_RELU_BACKWARD=0x507144e1, MAX_FORWARD=0xdf6f014c, MAX_BACKWARD=0xdf6f014d,
    MIN_FORWARD=0x972fbd26, MIN_BACKWARD=0x972fbd27, MSE_FORWARD=0x6904a9a2, MSE_BACKWARD=0x6904a9a3,
    PAD_FORWARD=0xd8aaca60, PAD_BACKWARD=0xd8aaca61, REDUCE_MAX_FORWARD=0x80f1a506, REDUCE_MAX_BACKWARD=0x80f1a507,
    REDUCE_MIN_FORWARD=0x6785ef96, REDUCE_MIN_BACKWARD=0x6785ef97, REDUCE_NORM2_FORWARD=0xb3034e16,
    REDUCE_NORM2_BACKWARD=0xb3034e17, RMSPROP_FORWARD=0x9c886b1c, RMSPROP_BACKWARD=0x9c886b1d,
    SIGMOID_FORWARD=0xf2f69650, SIGMOID_BACKWARD=0xf2f69651, SIGMOID_BINARY_CROSSENTROPY_FORWARD=0xd9e0e4a,
    SIGMOID_BINARY_CROSSENTROPY_BACKWARD=0xd9e0e4b, SMOOTH_L1_FORWARD=0x4e428e, SMOOTH_L1_BACKWARD=0x4e428f,
    SOFTMAX_FORWARD=0xc969a252, SOFTMAX_BACKWARD=0xc969a253, SWISH_FORWARD=0x583d90c2, SWISH_BACKWARD=0x583d90c3,
    TANH_FORWARD=0x6a62be30, TANH_BACKWARD=0x6a62be31, UPSAMPLE_FORWARD=0x73875556, UPSAMPLE_BACKWARD=0x73875557,
)

_DT_NP = {CCV_32F: np.float32, CCV_32S: np.int32, CCV_64F: np.float64, CCV_16F: np.float16, CCV_8U: np.uint8, CCV_64S: np.int64}
_NP_DT = {np.dtype(v): k for k, v in _DT_NP.items()}


# ---- struct mirrors -----------------------------------------------------------------------------------------
class TensorParam(C.Structure):
    _fields_ = [("type", C.c_int), ("format", C.c_int), ("datatype", C.c_int), ("reserved", C.c_int), ("dim", C.c_int * MAX_DIM_ALLOC)]


class TensorStruct(C.Structure):
    _fields_ = [("type", C.c_int), ("refcount", C.c_int), ("data", C.c_void_p), ("dataof", C.c_long), ("alias_ref", C.c_size_t),
                ("data_size", C.c_uint64), ("sig", C.c_uint64), ("info", TensorParam)]


class TensorViewStruct(C.Structure):
    _fields_ = TensorStruct._fields_ + [("contiguous", C.c_int), ("off", C.c_long), ("stride", C.c_int * MAX_DIM_ALLOC)]


class _Size(C.Structure):
    _fields_ = [("dim", C.c_int * MAX_DIM_ALLOC)]


class _Conv(C.Structure):
    _fields_ = [("count", C.c_int), ("groups", C.c_int), ("dilation", C.c_int * MAX_DIM_ALLOC)]


class _ConvTranspose(C.Structure):  # ccv_nnc.h:121-126: the convolution block + output_padding
    _fields_ = [("count", C.c_int), ("groups", C.c_int), ("dilation", C.c_int * MAX_DIM_ALLOC), ("output_padding", C.c_int)]


class _Bnorm(C.Structure):
    _fields_ = [("axis", C.c_int * MAX_DIM_ALLOC), ("count", C.c_int), ("epsilon", C.c_float), ("is_test", C.c_int), ("momentum", C.c_float)]


class _Sgd(C.Structure):
    _fields_ = [("nesterov", C.c_int), ("rate", C.c_float), ("scale", C.c_float), ("decay", C.c_float), ("momentum", C.c_float), ("dampening", C.c_float)]


class _Blas(C.Structure):
    _fields_ = [("transpose_a", C.c_int * 2), ("transpose_b", C.c_int * 2), ("a", C.c_float * 3), ("flags", C.c_int)]


class _LabelSmoothing(C.Structure):
    _fields_ = [("trim0", C.c_float), ("trim1", C.c_float)]


class _Reduce(C.Structure):
    _fields_ = [("axis", C.c_int * MAX_DIM_ALLOC), ("count", C.c_int)]


class _Transpose(C.Structure):
    _fields_ = [("axis", C.c_int * 2)]


class _Clamp(C.Structure):
    _fields_ = [("min", C.c_float), ("max", C.c_float)]


class _Gelu(C.Structure):
    _fields_ = [("tanh", C.c_int)]


class _LeakyRelu(C.Structure):
    _fields_ = [("negative_slope", C.c_float)]


class _Adam(C.Structure):
    _fields_ = [("step", C.c_int), ("rate", C.c_float), ("scale", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("decay", C.c_float), ("epsilon", C.c_float), ("amsgrad", C.c_int)]


class _Rmsprop(C.Structure):
    _fields_ = [("rate", C.c_float), ("scale", C.c_float), ("decay", C.c_float), ("alpha", C.c_float), ("momentum", C.c_float), ("epsilon", C.c_float)]


class _Lnorm(C.Structure):
    _fields_ = [("axis", C.c_int * MAX_DIM_ALLOC), ("count", C.c_int), ("epsilon", C.c_float), ("elementwise_affine", C.c_int)]


class _Gnorm(C.Structure):
    _fields_ = [("group_axis", C.c_int), ("reduce_axis", C.c_int * MAX_DIM_ALLOC), ("reduce_count", C.c_int), ("groups", C.c_int), ("epsilon", C.c_float), ("elementwise_affine", C.c_int)]


class _Upsample(C.Structure):
    _fields_ = [("type", C.c_int), ("width_scale", C.c_float), ("height_scale", C.c_float), ("align_corners", C.c_int)]


class _Pad(C.Structure):
    _fields_ = [("type", C.c_int), ("end", C.c_int * MAX_DIM_ALLOC)]


class _F1(C.Structure):   # binary_crossentropy.pos_weight / smooth_l1.beta
    _fields_ = [("v", C.c_float)]


class _I1(C.Structure):   # mse.reduce_op
    _fields_ = [("v", C.c_int)]


class _Rnn(C.Structure):  # ccv_nnc.h:127-136
    _fields_ = [("hidden_size", C.c_int), ("proj_size", C.c_int), ("num_layers", C.c_int), ("bias", C.c_int), ("batch_first", C.c_int), ("bidirectional", C.c_int), ("dropout", C.c_float), ("is_test", C.c_int)]


class _Dropout(C.Structure):  # ccv_nnc.h:235-238
    _fields_ = [("p", C.c_float), ("entirety", C.c_int)]


class _CmdUnion(C.Union):
    _fields_ = [("convolution", _Conv), ("convolution_transpose", _ConvTranspose), ("bnorm", _Bnorm), ("sgd", _Sgd), ("blas", _Blas), ("label_smoothing", _LabelSmoothing),
                ("reduce", _Reduce), ("transpose", _Transpose), ("clamp", _Clamp), ("gelu", _Gelu), ("leaky_relu", _LeakyRelu),
                ("adam", _Adam), ("rmsprop", _Rmsprop), ("f1", _F1), ("i1", _I1), ("pad", _Pad), ("upsample", _Upsample), ("lnorm", _Lnorm), ("gnorm", _Gnorm), ("rnn", _Rnn), ("dropout", _Dropout), ("_widest", C.c_char * 68), ("userdata", C.c_void_p)]


class CmdParam(C.Structure):
    _anonymous_ = ("u",)
    _fields_ = [("size", _Size), ("u", _CmdUnion)]


class Cmd(C.Structure):
    _fields_ = [("cmd", C.c_uint32), ("backend", C.c_uint32), ("algorithm", C.c_int), ("info", CmdParam), ("isa", C.c_void_p), ("data", C.c_void_p)]


class _Stride(C.Structure):
    _fields_ = [("dim", C.c_int * MAX_DIM_ALLOC)]


class _Border(C.Structure):
    _fields_ = [("begin", C.c_int * MAX_DIM_ALLOC), ("end", C.c_int * MAX_DIM_ALLOC)]


class Hint(C.Structure):
    _fields_ = [("stride", _Stride), ("border", _Border)]


class BackendRegistry(C.Structure):
    _fields_ = [("tensor_formats", C.c_int), ("tensor_datatypes", C.c_int), ("tensor_memory", C.c_int), ("algorithms", C.c_int),
                ("exec", C.c_void_p), ("autotune", C.c_void_p), ("aux", C.c_void_p)]


assert C.sizeof(TensorParam) == 64 and C.sizeof(TensorStruct) == 112 and C.sizeof(TensorViewStruct) == 176
assert C.sizeof(CmdParam) == 120 and C.sizeof(Cmd) == 152 and C.sizeof(Hint) == 144 and C.sizeof(BackendRegistry) == 40

NO_HINT = Hint()


# ---- command builders (lib/nnc/cmd/ccv_nnc_cmd_easy.h) --------------------------------------------------------
def _cmd(name, size=(1, 1, 1), backend=NO_BACKEND):
    c = Cmd()
    c.cmd = CMD[name]
    c.backend = backend
    c.algorithm = -1
    for i, v in enumerate(size):
        c.info.size.dim[i] = v
    return c


def CMD_CONVOLUTION_FORWARD(groups, count, *size, dilation=None):
    c = _cmd("CONVOLUTION_FORWARD", size)
    c.info.convolution.count, c.info.convolution.groups = count, groups
    if dilation:
        c.info.convolution.dilation[0], c.info.convolution.dilation[1] = dilation
    return c


def CMD_CONVOLUTION_TRANSPOSE_FORWARD(groups, count, output_padding, *size, dilation=None):
    """CMD_CONVOLUTION_TRANSPOSE_FORWARD(_groups, _count, _output_padding, kh, kw, c) (lib/nnc/cmd/ccv_nnc_cmd_easy.h:58)."""
    c = _cmd("CONVOLUTION_TRANSPOSE_FORWARD", size)
    t = c.info.convolution_transpose
    t.count, t.groups, t.output_padding = count, groups, output_padding
    if dilation:
        t.dilation[0], t.dilation[1] = dilation
    return c


def CMD_CONVOLUTION_BACKWARD(groups, count, *size, dilation=None):
    c = CMD_CONVOLUTION_FORWARD(groups, count, *size, dilation=dilation)
    c.cmd = CMD["CONVOLUTION_BACKWARD"]
    return c


def _gemm(name, ta, tb):
    c = _cmd(name)
    c.info.blas.a[0] = c.info.blas.a[1] = 1
    c.info.blas.transpose_a[0], c.info.blas.transpose_a[1] = ta
    c.info.blas.transpose_b[0], c.info.blas.transpose_b[1] = tb
    return c


NO_TRANSPOSE = (0, 0)


def TRANSPOSE(x, y):
    return (x, y)


def CMD_GEMM_FORWARD(ta=NO_TRANSPOSE, tb=NO_TRANSPOSE):
    return _gemm("GEMM_FORWARD", ta, tb)


def CMD_GEMM_BACKWARD(ta=NO_TRANSPOSE, tb=NO_TRANSPOSE):
    return _gemm("GEMM_BACKWARD", ta, tb)


def CMD_MAX_POOL_FORWARD(rows, cols): return _cmd("MAX_POOL_FORWARD", (rows, cols, 1))
def CMD_MAX_POOL_BACKWARD(rows, cols): return _cmd("MAX_POOL_BACKWARD", (rows, cols, 1))
def CMD_AVERAGE_POOL_FORWARD(rows, cols): return _cmd("AVERAGE_POOL_FORWARD", (rows, cols, 1))
def CMD_AVERAGE_POOL_BACKWARD(rows, cols): return _cmd("AVERAGE_POOL_BACKWARD", (rows, cols, 1))
def CMD_RELU_FORWARD(): return _cmd("RELU_FORWARD", (0, 0, 0))
def CMD_RELU_BACKWARD(): return _cmd("RELU_BACKWARD", (0, 0, 0))
def CMD_EWSUM_FORWARD(): return _cmd("EWSUM_FORWARD", (0, 0, 0))
def CMD_EWSUM_BACKWARD(): return _cmd("EWSUM_BACKWARD", (0, 0, 0))
def CMD_DATA_TRANSFER_FORWARD(): return _cmd("DATA_TRANSFER_FORWARD", (0, 0, 0))
def CMD_FORMAT_TRANSFORM_FORWARD(): return _cmd("FORMAT_TRANSFORM_FORWARD", (0, 0, 0))


def CMD_SOFTMAX_CROSSENTROPY_FORWARD(trim0=0.0, trim1=1.0):
    c = _cmd("SOFTMAX_CROSSENTROPY_FORWARD")
    c.info.label_smoothing.trim0, c.info.label_smoothing.trim1 = trim0, trim1
    return c


def CMD_SOFTMAX_CROSSENTROPY_BACKWARD(trim0=0.0, trim1=1.0):
    c = CMD_SOFTMAX_CROSSENTROPY_FORWARD(trim0, trim1)
    c.cmd = CMD["SOFTMAX_CROSSENTROPY_BACKWARD"]
    return c


def CMD_SGD_FORWARD(nesterov, rate, scale, decay, momentum, dampening):
    c = _cmd("SGD_FORWARD")
    s = c.info.sgd
    s.nesterov, s.rate, s.scale, s.decay, s.momentum, s.dampening = nesterov, rate, scale, decay, momentum, dampening
    return c


def CMD_DROPOUT_FORWARD(p, entirety=0):
    c = _cmd("DROPOUT_FORWARD", (0, 0, 0))
    c.info.dropout.p, c.info.dropout.entirety = p, entirety
    return c


def _blas_a(name, *a):
    c = _cmd(name)
    for i, v in enumerate(a):
        c.info.blas.a[i] = v
    return c


def CMD_SET_FORWARD(val): return _blas_a("SET_FORWARD", val)
def CMD_SCALAR_MUL_FORWARD(a): return _blas_a("SCALAR_MUL_FORWARD", a)
def CMD_SCALAR_MUL_BACKWARD(a): return _blas_a("SCALAR_MUL_BACKWARD", a)
def CMD_ADD_FORWARD(p, q): return _blas_a("ADD_FORWARD", p, q)
def CMD_ADD_BACKWARD(p, q): return _blas_a("ADD_BACKWARD", p, q)
def CMD_MUL_FORWARD(p): return _blas_a("MUL_FORWARD", p)
def CMD_MUL_BACKWARD(p): return _blas_a("MUL_BACKWARD", p)


def CMD_BATCH_NORM_FORWARD(epsilon, is_test, momentum, *axis):
    c = _cmd("BATCH_NORM_FORWARD")
    b = c.info.bnorm
    b.epsilon, b.is_test, b.momentum, b.count = epsilon, is_test, momentum, len(axis)
    for i, a in enumerate(axis):
        b.axis[i] = a
    return c


def CMD_BATCH_NORM_BACKWARD(epsilon, is_test, momentum, *axis):
    c = CMD_BATCH_NORM_FORWARD(epsilon, is_test, momentum, *axis)
    c.cmd = CMD["BATCH_NORM_BACKWARD"]
    return c


def _reduce(name, *axis):
    c = _cmd(name)
    c.info.reduce.count = len(axis)
    for i, a in enumerate(axis):
        c.info.reduce.axis[i] = a
    return c


def CMD_REDUCE_SUM_FORWARD(*axis): return _reduce("REDUCE_SUM_FORWARD", *axis)
def CMD_REDUCE_SUM_BACKWARD(*axis): return _reduce("REDUCE_SUM_BACKWARD", *axis)
def CMD_REDUCE_MEAN_FORWARD(*axis): return _reduce("REDUCE_MEAN_FORWARD", *axis)
def CMD_REDUCE_MEAN_BACKWARD(*axis): return _reduce("REDUCE_MEAN_BACKWARD", *axis)


def CMD_GELU_FORWARD(tanh=0):
    c = _cmd("GELU_FORWARD", (0, 0, 0)); c.info.gelu.tanh = tanh; return c
def CMD_GELU_BACKWARD(tanh=0):
    c = _cmd("GELU_BACKWARD", (0, 0, 0)); c.info.gelu.tanh = tanh; return c
def CMD_LEAKY_RELU_FORWARD(slope):
    c = _cmd("LEAKY_RELU_FORWARD", (0, 0, 0)); c.info.leaky_relu.negative_slope = slope; return c
def CMD_LEAKY_RELU_BACKWARD(slope):
    c = _cmd("LEAKY_RELU_BACKWARD", (0, 0, 0)); c.info.leaky_relu.negative_slope = slope; return c


def CMD_ADAM_FORWARD(step, rate, beta1, beta2, decay, epsilon, amsgrad=0, scale=1.0, decoupled=False):
    """CMD_ADAM_FORWARD / CMD_ADAMW_FORWARD(step, rate, beta1, beta2, decay, epsilon, amsgrad) (ccv_nnc_easy.h macro order; scale = 1)."""
    c = _cmd("ADAMW_FORWARD" if decoupled else "ADAM_FORWARD", (0, 0, 0))
    a = c.info.adam
    a.step, a.rate, a.scale, a.beta1, a.beta2, a.decay, a.epsilon, a.amsgrad = step, rate, scale, beta1, beta2, decay, epsilon, amsgrad
    return c


def CMD_RMSPROP_FORWARD(rate, decay, alpha, momentum, epsilon, scale=1.0):
    c = _cmd("RMSPROP_FORWARD", (0, 0, 0))
    r = c.info.rmsprop
    r.rate, r.scale, r.decay, r.alpha, r.momentum, r.epsilon = rate, scale, decay, alpha, momentum, epsilon
    return c


def _f1(name, v):
    c = _cmd(name, (0, 0, 0)); c.info.f1.v = v; return c
def CMD_MSE_FORWARD(reduce_op=0):     # 0 = CCV_NNC_MSE_REDUCE_MEAN, 1 = CCV_NNC_MSE_REDUCE_SUM
    c = _cmd("MSE_FORWARD", (0, 0, 0)); c.info.i1.v = reduce_op; return c
def CMD_MSE_BACKWARD(reduce_op=0):
    c = _cmd("MSE_BACKWARD", (0, 0, 0)); c.info.i1.v = reduce_op; return c
def CMD_SMOOTH_L1_FORWARD(beta): return _f1("SMOOTH_L1_FORWARD", beta)
def CMD_SMOOTH_L1_BACKWARD(beta): return _f1("SMOOTH_L1_BACKWARD", beta)
def CMD_BINARY_CROSSENTROPY_FORWARD(pos_weight=1.0): return _f1("BINARY_CROSSENTROPY_FORWARD", pos_weight)
def CMD_BINARY_CROSSENTROPY_BACKWARD(pos_weight=1.0): return _f1("BINARY_CROSSENTROPY_BACKWARD", pos_weight)
def CMD_CATEGORICAL_CROSSENTROPY_FORWARD(trim0=0.0, trim1=1.0):
    c = _cmd("CATEGORICAL_CROSSENTROPY_FORWARD", (0, 0, 0)); c.info.label_smoothing.trim0, c.info.label_smoothing.trim1 = trim0, trim1; return c
def CMD_CATEGORICAL_CROSSENTROPY_BACKWARD(trim0=0.0, trim1=1.0):
    c = _cmd("CATEGORICAL_CROSSENTROPY_BACKWARD", (0, 0, 0)); c.info.label_smoothing.trim0, c.info.label_smoothing.trim1 = trim0, trim1; return c


def CMD_NORM(name, epsilon, affine, *axis):
    """CMD_LAYER_NORM_*(epsilon, elementwise_affine, axis...) / CMD_RMSNORM_*(epsilon, axis...): rmsnorm shares lnorm's leading fields"""
    c = _cmd(name, (0, 0, 0))
    for i, a in enumerate(axis):
        c.info.lnorm.axis[i] = a
    c.info.lnorm.count, c.info.lnorm.epsilon, c.info.lnorm.elementwise_affine = len(axis), epsilon, affine
    return c


def CMD_GROUP_NORM(name, group_axis, groups, epsilon, affine, *reduce_axis):
    """CMD_GROUP_NORM_*(group_axis, groups, epsilon, elementwise_affine, reduce axes...) (ccv_nnc_easy.h)"""
    c = _cmd(name, (0, 0, 0))
    gn = c.info.gnorm
    gn.group_axis, gn.groups, gn.epsilon, gn.elementwise_affine, gn.reduce_count = group_axis, groups, epsilon, affine, len(reduce_axis)
    for i, a in enumerate(reduce_axis):
        gn.reduce_axis[i] = a
    return c


def CMD_UPSAMPLE(name, up_type, width_scale, height_scale, align_corners):
    """CMD_UPSAMPLE_*(type, width_scale, height_scale, align_corners): 0 nearest, 1 bilinear"""
    c = _cmd(name, (0, 0, 0))
    u = c.info.upsample
    u.type, u.width_scale, u.height_scale, u.align_corners = up_type, width_scale, height_scale, align_corners
    return c


def CMD_PAD(name, pad_type, begin, end):
    """CMD_PAD_FORWARD(type, (begin...), (end...)): begin goes to info.size.dim, end to info.pad.end (ccv_nnc_easy.h)"""
    c = _cmd(name, tuple(begin))
    c.info.pad.type = pad_type
    for i, e in enumerate(end):
        c.info.pad.end[i] = e
    return c


CONV_ALGO_FUSE_RELU = 0x100  # NNC_MI355X_CONV_ALGO_FUSE_RELU (include/nnc_mi355x.h)
POOL_ALGO_FUSE_RELU_BACKWARD = 0x100  # NNC_MI355X_POOL_ALGO_FUSE_RELU_BACKWARD
BNORM_ALGO_FUSE_RELU = 0x100  # NNC_MI355X_BNORM_ALGO_FUSE_RELU
EWSUM_ALGO_FUSE_RELU = 0x100  # NNC_MI355X_EWSUM_ALGO_FUSE_RELU
EWSUM_ALGO_FUSE_RELU_BACKWARD = 0x200  # NNC_MI355X_EWSUM_ALGO_FUSE_RELU_BACKWARD: the last input is the mask map


def _lstm(name, hidden_size, proj_size, num_layers, bias, batch_first, bidirectional, dropout, is_test):
    c = _cmd(name, (0, 0, 0))
    c.info.rnn.hidden_size, c.info.rnn.proj_size, c.info.rnn.num_layers, c.info.rnn.bias = hidden_size, proj_size, num_layers, bias
    c.info.rnn.batch_first, c.info.rnn.bidirectional, c.info.rnn.dropout, c.info.rnn.is_test = batch_first, bidirectional, dropout, is_test
    return c


def CMD_LSTM_FORWARD(*a): return _lstm("LSTM_FORWARD", *a)    # lib/nnc/cmd/rnn/ccv_nnc_lstm.c:84
def CMD_LSTM_BACKWARD(*a): return _lstm("LSTM_BACKWARD", *a)  # :86


def generic_cmd(name, size=(0, 0, 0)):
    return _cmd(name, size)


def HINT(stride=(1, 1), border=(0, 0), border_end=None):
    """HINT((sy, sx), (by, bx)) as in lib/nnc/ccv_nnc_easy.h."""
    h = Hint()
    for i, s in enumerate(stride):
        h.stride.dim[i] = s
    for i, b in enumerate(border):
        h.border.begin[i] = b
    for i, b in enumerate(border_end if border_end is not None else border):
        h.border.end[i] = b
    return h


def hint_auto(cmd, a_dim_hw, b_dim_hw):
    """ccv_nnc_hint_auto (lib/nnc/ccv_nnc_cmd.c:178-217) for the 2 spatial dims: the stride/border mapping a onto b."""
    h = Hint()
    for i in range(2):
        a, b, k = a_dim_hw[i], b_dim_hw[i], cmd.info.size.dim[i]
        stride = (a + b // 2) // b
        border = (b - 1) * stride - a + k
        begin = int((border + 1) / 2)  # C division truncates toward zero
        h.stride.dim[i], h.border.begin[i], h.border.end[i] = stride, begin, border - begin
    return h


# ---- tensors ------------------------------------------------------------------------------------------------
def tensor_param(memory, fmt, datatype, dims, device=0):
    p = TensorParam()
    p.type = memory | (device << 8)
    p.format, p.datatype = fmt, datatype
    for i, d in enumerate(dims):
        p.dim[i] = d
    return p


def CPU_TENSOR_NHWC(datatype, *dims): return tensor_param(CPU_MEMORY, NHWC, datatype, dims)
def CPU_TENSOR_NCHW(datatype, *dims): return tensor_param(CPU_MEMORY, NCHW, datatype, dims)
def GPU_TENSOR_NHWC(device, datatype, *dims): return tensor_param(GPU_MEMORY, NHWC, datatype, dims, device)
def GPU_TENSOR_NCHW(device, datatype, *dims): return tensor_param(GPU_MEMORY, NCHW, datatype, dims, device)


def param_dims(p):
    out = []
    for i in range(MAX_DIM_ALLOC):
        if p.dim[i] == 0:
            break
        out.append(p.dim[i])
    return tuple(out)


class Tensor:
    """A ccv_nnc_tensor_t (or tensor view) plus the memory behind it."""

    def __init__(self, lib, params, array=None, view_of=None, strides=None, offset=0):
        self.lib = lib
        self.dims = param_dims(params)
        self.datatype = params.datatype
        self.np_dtype = np.dtype(_DT_NP[params.datatype & 0xFF000])
        self.memory = params.type & 0x3
        self.device = (params.type & 0xfff00) >> 8
        self.owner = None
        self._dptr = None
        if view_of is not None and strides is None:  # ccv_nnc_tensor(ptr, params, 0): a plain dense tensor over part of another one
            self.struct = TensorStruct()
            self.owner = view_of.owner if view_of.owner is not None else view_of
            self.ptr = view_of.ptr + offset * self.np_dtype.itemsize
            self.struct.type = params.type
            self.struct.data = self.ptr
        elif view_of is not None:  # ccv_nnc_tensor_view_new: same memory, explicit element strides
            self.struct = TensorViewStruct()
            self.owner = view_of
            base = view_of.ptr + offset * self.np_dtype.itemsize
            self.struct.type = params.type | TENSOR_VIEW
            self.struct.contiguous = 0
            self.struct.off = offset * self.np_dtype.itemsize
            for i, s in enumerate(strides):
                self.struct.stride[i] = s
            self.struct.data = base
            self.ptr = base
        else:
            self.struct = TensorStruct()
            n = int(np.prod(self.dims)) if self.dims else 0
            nbytes = max(n * self.np_dtype.itemsize, 16)
            self.nbytes = n * self.np_dtype.itemsize
            if self.memory == CPU_MEMORY:
                self.array = np.zeros(self.dims, dtype=self.np_dtype) if array is None else np.array(array, dtype=self.np_dtype, order="C", copy=True).reshape(self.dims)
                self.ptr = self.array.ctypes.data
            else:
                self._dptr = lib.malloc(self.device, (nbytes + 127) & ~127)  # GPU tensors round to 128 B (ccv_nnc_easy.h:238-244)
                if not self._dptr:
                    raise MemoryError("device allocation of %d bytes failed" % nbytes)
                self.ptr = self._dptr
                if array is not None:
                    self.upload(array)
            self.struct.type = params.type
            self.struct.data = self.ptr
        self.struct.info = params
        self.struct.refcount = 1

    def upload(self, array):
        a = np.ascontiguousarray(array, dtype=self.np_dtype).reshape(self.dims)
        self.lib.memcpy(self.ptr, GPU_MEMORY | (self.device << 8), a.ctypes.data, CPU_MEMORY, a.nbytes)

    def numpy(self):
        """Contents as a fresh numpy array (device tensors are copied back, blocking)."""
        if self.owner is not None:  # contiguous views only: slice the owner's flat contents
            flat = self.owner.numpy().reshape(-1)
            off = (self.ptr - self.owner.ptr) // self.np_dtype.itemsize
            return flat[off:off + int(np.prod(self.dims))].reshape(self.dims).copy()
        if self.memory == CPU_MEMORY:
            return self.array.copy()
        out = np.empty(self.dims, dtype=self.np_dtype)
        if out.nbytes:
            self.lib.memcpy(out.ctypes.data, CPU_MEMORY, self.ptr, GPU_MEMORY | (self.device << 8), out.nbytes)
        return out

    def alias(self, dims, offset=0):
        """A dense tensor of `dims` over this tensor's memory starting at element `offset` (no view flag)."""
        p = TensorParam()
        C.memmove(C.byref(p), C.byref(self.struct.info), C.sizeof(p))
        for i in range(MAX_DIM_ALLOC):
            p.dim[i] = dims[i] if i < len(dims) else 0
        return Tensor(self.lib, p, view_of=self, strides=None, offset=offset)

    def view(self, dims, strides, offset=0, fmt=None):
        p = TensorParam()
        C.memmove(C.byref(p), C.byref(self.struct.info), C.sizeof(p))
        for i in range(MAX_DIM_ALLOC):
            p.dim[i] = dims[i] if i < len(dims) else 0
        if fmt is not None:
            p.format = fmt
        return Tensor(self.lib, p, view_of=self, strides=strides, offset=offset)

    def free(self):
        if self._dptr:
            self.lib.free(self.device, self._dptr)
            self._dptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    @property
    def c(self):
        return C.cast(C.pointer(self.struct), C.POINTER(TensorStruct))


def tensor_palettize(params, qbits, number_in_blocks):
    """ccv_nnc_tensor_palettize (lib/nnc/ccv_nnc_easy.h:210-218): the parameters of the palettized form of a 16F / 32F / 64F tensor."""
    assert params.datatype in (CCV_16F, CCV_32F, CCV_64F) and 4 <= qbits <= 8
    p = TensorParam()
    C.memmove(C.byref(p), C.byref(params), C.sizeof(p))
    p.datatype = ((params.datatype >> 12) & 0xff) | CCV_QX | ((qbits << 8) & 0xf00)
    p.reserved = number_in_blocks
    return p


class PalettizedTensor(Tensor):
    """A CCV_QX tensor: `stream` is the byte stream the host's ccv_nnc_palettize writes (per block: the palette, then the indices)."""

    def __init__(self, lib, params, stream):
        assert (params.datatype & 0xFF000) == CCV_QX
        self.lib = lib
        self.dims = param_dims(params)
        self.datatype = params.datatype
        self.np_dtype = np.dtype(np.uint8)
        self.memory = params.type & 0x3
        self.device = (params.type & 0xfff00) >> 8
        self.owner = None
        self._dptr = None
        stream = np.ascontiguousarray(stream, dtype=np.uint8).reshape(-1)
        self.nbytes = stream.nbytes
        self.struct = TensorStruct()
        if self.memory == CPU_MEMORY:
            self.array = stream.copy()
            self.ptr = self.array.ctypes.data
        else:
            self._dptr = lib.malloc(self.device, (max(stream.nbytes, 16) + 127) & ~127)
            if not self._dptr:
                raise MemoryError("device allocation of %d bytes failed" % stream.nbytes)
            self.ptr = self._dptr
            lib.memcpy(self.ptr, GPU_MEMORY | (self.device << 8), stream.ctypes.data, CPU_MEMORY, stream.nbytes)
        self.struct.type = params.type
        self.struct.data = self.ptr
        self.struct.info = params
        self.struct.refcount = 1

    def numpy(self):
        out = np.empty(self.nbytes, dtype=np.uint8)
        if self.memory == CPU_MEMORY:
            return self.array.copy()
        self.lib.memcpy(out.ctypes.data, CPU_MEMORY, self.ptr, GPU_MEMORY | (self.device << 8), out.nbytes)
        return out


def _tensor_array(tensors):
    arr = (C.POINTER(TensorStruct) * max(1, len(tensors)))()
    for i, t in enumerate(tensors):
        arr[i] = t.c if t is not None else None
    return arr


class Lib:
    """A loaded implementation of the command interface."""
    _EXEC_ARGS = [Cmd, Hint, C.c_int, C.POINTER(C.POINTER(TensorStruct)), C.c_int, C.POINTER(C.POINTER(TensorStruct)), C.c_int, C.c_void_p]

    def __init__(self, path, kind="mi355x"):
        # This package loads the MI355X backend (or the CPU emulator build of the same sources in the test tier) and nothing else: the checker
        # libraries -- the reference's CPU backend, the C restatement -- are loaded by tests/oracle_bind.py (CheckerLib), outside the product.
        if kind != "mi355x":
            raise ValueError("ccv_amd.nnc.Lib loads libnnc_mi355x.so only; checker libraries: tests/oracle_bind.py")
        self.path, self.kind = path, kind
        self.dll = C.CDLL(path, mode=C.RTLD_GLOBAL)
        self._bind()
        self._exec.restype = C.c_int
        self._exec.argtypes = self._EXEC_ARGS

    def _bind(self):
        d = self.dll
        if True:
            self._exec = d.nnc_mi355x_cmd_exec
            d.nnc_mi355x_malloc.restype = C.c_void_p
            d.nnc_mi355x_malloc.argtypes = [C.c_int, C.c_size_t]
            d.nnc_mi355x_free.argtypes = [C.c_int, C.c_void_p]
            d.nnc_mi355x_memcpy.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_size_t]
            d.nnc_mi355x_stream_context_new.restype = C.c_void_p
            d.nnc_mi355x_stream_context_new.argtypes = [C.c_int]
            d.nnc_mi355x_stream_context_free.argtypes = [C.c_void_p]
            d.nnc_mi355x_stream_context_wait.argtypes = [C.c_void_p]
            d.nnc_mi355x_stream_signal_new.restype = C.c_void_p
            d.nnc_mi355x_stream_signal_new.argtypes = [C.c_int]
            d.nnc_mi355x_stream_signal_free.argtypes = [C.c_void_p]
            d.ccv_nnc_stream_compat_emit_signal.argtypes = [C.c_void_p, C.c_void_p]
            d.ccv_nnc_stream_compat_wait_signal.argtypes = [C.c_void_p, C.c_void_p]
            d.nnc_mi355x_event_new.restype = C.c_void_p
            d.nnc_mi355x_event_record.argtypes = [C.c_void_p, C.c_void_p]
            d.nnc_mi355x_event_elapsed_ms.restype = C.c_float
            d.nnc_mi355x_event_elapsed_ms.argtypes = [C.c_void_p, C.c_void_p]
            d.nnc_mi355x_event_free.argtypes = [C.c_void_p]
            d.nnc_mi355x_last_kernel_name.restype = C.c_char_p
            d.nnc_mi355x_registry_name.restype = C.c_char_p
            d.nnc_mi355x_registry_get.argtypes = [C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(BackendRegistry)]
            d.nnc_mi355x_cmd_ok.argtypes = [C.c_uint32, C.c_uint32]
            d.nnc_mi355x_version.restype = C.c_char_p
            d.nnc_mi355x_lstm_reserve_space_size.restype = C.c_size_t
            d.nnc_mi355x_lstm_reserve_space_size.argtypes = [Cmd, C.c_int, C.c_int, C.c_int, C.c_int]
            d.nnc_mi355x_depalettize.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
            d.nnc_mi355x_palettized_bytes.restype = C.c_size_t
            d.nnc_mi355x_palettized_bytes.argtypes = [C.c_int, C.c_size_t, C.c_int, C.c_int]
            d.nnc_mi355x_profile_get.argtypes = [C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int)]
            d.nnc_mi355x_capture_begin.argtypes = [C.c_void_p]
            d.nnc_mi355x_capture_end.restype = C.c_void_p
            d.nnc_mi355x_capture_end.argtypes = [C.c_void_p]
            d.nnc_mi355x_graph_launch.argtypes = [C.c_void_p, C.c_void_p]
            d.nnc_mi355x_graph_node_count.argtypes = [C.c_void_p]
            d.nnc_mi355x_graph_free.argtypes = [C.c_void_p]
            d.nnc_mi355x_debug_pool_parked_bytes.restype = C.c_long

    # device runtime (product / emulator only)
    def malloc(self, device, size): return self.dll.nnc_mi355x_malloc(device, size)
    def free(self, device, ptr): self.dll.nnc_mi355x_free(device, ptr)
    def memcpy(self, dst, dst_type, src, src_type, n): self.dll.nnc_mi355x_memcpy(dst, dst_type, src, src_type, n)
    def device_count(self): return self.dll.nnc_mi355x_device_count()
    def set_device(self, d): self.dll.nnc_mi355x_set_device(d)
    def stream_new(self, device=0): return self.dll.nnc_mi355x_stream_context_new(STREAM_CONTEXT_GPU | (device << 8))
    def stream_free(self, s): self.dll.nnc_mi355x_stream_context_free(s)
    def stream_wait(self, s): self.dll.nnc_mi355x_stream_context_wait(s)
    # ccv_nnc_stream_signal_new / ccv_nnc_stream_context_emit_signal / _wait_signal (lib/nnc/ccv_nnc_stream.c:304-342)
    def signal_new(self, device=0): return self.dll.nnc_mi355x_stream_signal_new(STREAM_CONTEXT_GPU | (device << 8))
    def signal_free(self, sig): self.dll.nnc_mi355x_stream_signal_free(sig)
    def signal_emit(self, stream, sig): self.dll.ccv_nnc_stream_compat_emit_signal(stream, sig)
    def signal_wait(self, stream, sig): self.dll.ccv_nnc_stream_compat_wait_signal(stream, sig)
    def cmd_ok(self, cmd, backend): return bool(self.dll.nnc_mi355x_cmd_ok(cmd, backend))
    # HIP-graph capture of a step (include/nnc_mi355x.h "HIP-graph capture"): record the enqueue-only calls made on `stream` between begin and end, replay them
    def capture_begin(self, stream): return self.dll.nnc_mi355x_capture_begin(stream)
    def capture_end(self, stream): return self.dll.nnc_mi355x_capture_end(stream)
    def graph_launch(self, graph, stream): return self.dll.nnc_mi355x_graph_launch(graph, stream)
    def graph_node_count(self, graph): return self.dll.nnc_mi355x_graph_node_count(graph)
    def graph_free(self, graph): self.dll.nnc_mi355x_graph_free(graph)
    def capture_keep_streams(self, on): self.dll.nnc_mi355x_capture_keep_streams(int(on))
    def pool_parked_bytes(self): return self.dll.nnc_mi355x_debug_pool_parked_bytes()

    def depalettize(self, src, datatype, input_length, qbits, number_in_blocks, dst, output_length, stream=None):
        """ccv_nnc_depalettize of device memory (lib/nnc/ccv_nnc_palettize.c:958-966): src / dst are Tensors (or raw device pointers)."""
        sp, dp = getattr(src, "ptr", src), getattr(dst, "ptr", dst)
        return self.dll.nnc_mi355x_depalettize(sp, datatype, input_length, qbits, number_in_blocks, dp, output_length, stream)

    def palettized_bytes(self, datatype, count, qbits, number_in_blocks): return self.dll.nnc_mi355x_palettized_bytes(datatype, count, qbits, number_in_blocks)

    def profile_enable(self, on): self.dll.nnc_mi355x_profile_enable(int(on))
    def force_tile(self, wm, wn): self.dll.nnc_mi355x_debug_force_tile(int(wm), int(wn))
    def force_splits(self, n): self.dll.nnc_mi355x_debug_force_splits(int(n))

    def tune_set(self, name, value):
        self.dll.nnc_mi355x_tune_set.argtypes = [C.c_char_p, C.c_long]
        if self.dll.nnc_mi355x_tune_set(name.encode(), int(value)) != 0:
            raise KeyError(name)

    def tune_get(self, name):
        self.dll.nnc_mi355x_tune_get.argtypes = [C.c_char_p]
        self.dll.nnc_mi355x_tune_get.restype = C.c_long
        return self.dll.nnc_mi355x_tune_get(name.encode())

    def profile_records(self):
        """[(name, flops, bytes, ms, (M, N, K, Z, splits))] for every contraction launch since profile_enable(1)."""
        out = []
        for i in range(self.dll.nnc_mi355x_profile_count()):
            name = C.create_string_buffer(256)
            fl, by, ms, dims = C.c_double(), C.c_double(), C.c_float(), (C.c_int * 5)()
            self.dll.nnc_mi355x_profile_get(i, name, 256, C.byref(fl), C.byref(by), C.byref(ms), dims)
            out.append((name.value.decode(), fl.value, by.value, ms.value, tuple(dims)))
        return out

    def registry(self):
        rows = []
        for i in range(self.dll.nnc_mi355x_registry_count()):
            c, b, r = C.c_uint32(), C.c_uint32(), BackendRegistry()
            self.dll.nnc_mi355x_registry_get(i, C.byref(c), C.byref(b), C.byref(r))
            rows.append((self.dll.nnc_mi355x_registry_name(i).decode(), c.value, b.value, r))
        return rows

    def tensor(self, params, array=None):
        return Tensor(self, params, array)

    def cmd_exec(self, cmd, hint, flags, inputs, outputs, stream=None):
        """ccv_nnc_cmd_exec(cmd, hint, flags, inputs, input_size, outputs, output_size, stream_context)."""
        ia, oa = _tensor_array(inputs), _tensor_array(outputs)
        return self._exec(cmd, hint, flags, ia, len(inputs), oa, len(outputs), stream)


_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libnnc_mi355x.so")
_lib = None


def load(path=None):
    """Load the MI355X backend.  Fails loudly (no CPU fallback) when the HIP library or a GPU is missing."""
    global _lib
    if path is None and _lib is not None:
        return _lib
    p = path or LIB_PATH
    # multi-process GPU work (RCCL between ranks, device memory shared across processes): this image's host driver only supports dmabuf IPC; the
    # variable must be in the environment before the HIP runtime starts (it is exported on the boxes; kept here for environments built by hand)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not os.path.exists(p):
        raise RuntimeError("libnnc_mi355x.so not built: %s (run `python -c 'import __graft_entry__ as g; g.build()'`)" % p)
    lib = Lib(p, "mi355x")
    if path is None:
        if lib.device_count() <= 0:
            raise RuntimeError("libnnc_mi355x.so loaded but no HIP device is visible; this backend has no CPU fallback")
        _lib = lib
    return lib
