/*
 * nnc_mi355x.h -- C-ABI of libnnc_mi355x.so, the MI355X (gfx950) compute backend for
 * ccv's nnc tensor engine.
 *
 * This header is the DROP-IN BOUNDARY.  It declares, in plain C (no HIP, no torch types):
 *
 *   1. ABI mirrors of the reference's plugin-surface structs.  They are re-stated here
 *      (field order / sizes identical, only the members this path touches are named) so the
 *      library builds without the reference tree.  tests/test_abi.py compiles a
 *      sizeof/offsetof probe against the real headers whenever /root/reference is present.
 *        ccv_nnc_tensor_param_t / ccv_nnc_tensor_t / ccv_nnc_tensor_view_t
 *                                         <- lib/nnc/ccv_nnc_tfb.h:79-111
 *        ccv_nnc_cmd_param_t / ccv_nnc_hint_t / ccv_nnc_cmd_t / exec_f / autotune_f
 *                                         <- lib/nnc/ccv_nnc.h:111-323
 *        ccv_nnc_cmd_backend_registry_t   <- lib/nnc/ccv_nnc_internal.h:34-42
 *        ccv_nnc_stream_context_s / ccv_nnc_stream_signal_s (base part the host allocates)
 *                                         <- lib/nnc/_ccv_nnc_stream.h:17-46
 *
 *   2. The backend registration entry points.  The reference host calls one function per
 *      (command, backend slot) from its generated _ccv_nnc_cmd_init()
 *      (lib/nnc/cmd/ccv_nnc_cmd.inc:464-, :944-).  The slot names (GPU_REF / GPU_CUDNN /
 *      GPU_CUBLAS / GPU_NCCL) are the reference registry's names for "the GPU backends";
 *      nothing from those vendor libraries is used: every exec function launches a
 *      hand-written gfx950 HIP kernel from this library.
 *
 *   3. The device "compat" ABI the unmodified host .c files call for memory, streams,
 *      events and workspace (lib/nnc/gpu/ccv_nnc_compat.h:23-59).  Native names are
 *      nnc_mi355x_*; the reference-spelled aliases (cumalloc, ...) live in
 *      ccv_amd/csrc/ref_host_abi.cpp and simply forward.
 *
 *   4. A standalone dispatch (nnc_mi355x_cmd_exec) mirroring ccv_nnc_cmd_exec
 *      (lib/nnc/ccv_nnc_cmd.c:651-693) for callers that do not link the reference host
 *      (our tests, bench.py, smoke()).
 */
#ifndef NNC_MI355X_H
#define NNC_MI355X_H

#include <stddef.h>
#include <stdint.h>
#include <sys/types.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ constants ---- */
/* lib/nnc/ccv_nnc_tfb.h:26-58 */
enum { CCV_TENSOR_FORMAT_NCHW = 0x01, CCV_TENSOR_FORMAT_NHWC = 0x02, CCV_TENSOR_FORMAT_CHWN = 0x04 };
enum { CCV_TENSOR_CPU_MEMORY = 0x1, CCV_TENSOR_GPU_MEMORY = 0x2 };
enum { CCV_COMPUTE_DEVICE_ANY = 0xfff00 };
#define CCV_TENSOR_GET_MEMORY(type) ((type) & 0x3)
#define CCV_TENSOR_GET_DEVICE(type) ((type) & 0xfff00)
#define CCV_TENSOR_GET_DEVICE_ID(type) (CCV_TENSOR_GET_DEVICE(type) >> 8)
enum { CCV_TENSOR_VIEW = 0x01000000, CCV_TENSOR_MULTIVIEW = 0x02000000, CCV_TENSOR_PINNED_MEM = 0x04000000 };
/* lib/ccv.h:46-52,73 */
enum { CCV_8U = 0x01000, CCV_32S = 0x02000, CCV_32F = 0x04000, CCV_64S = 0x08000, CCV_64F = 0x10000, CCV_16F = 0x20000, CCV_QX = 0x40000 };
#define CCV_GET_DATA_TYPE(x) ((x) & 0xFF000)
/* lib/nnc/ccv_nnc.h:69-79 */
enum { CCV_NNC_ACCUMULATE_OUTPUT = 0x01, CCV_NNC_ZERO_MEMORY_ALLOC = 0x02 };
enum { CCV_NNC_EXEC_SUCCESS = 0, CCV_NNC_EXEC_INVALID = -1, CCV_NNC_EXEC_NO_KERNEL = -2, CCV_NNC_EXEC_OOM = -3 };
/* lib/nnc/ccv_nnc.h:928-933 */
enum { CCV_STREAM_CONTEXT_CPU = 0x1, CCV_STREAM_CONTEXT_GPU = 0x2 };
#define CCV_STREAM_GET_CONTEXT(type) ((type) & 0x3)
#define CCV_STREAM_GET_DEVICE_ID(type) CCV_TENSOR_GET_DEVICE_ID(type)

#define CCV_NNC_MAX_DIM_ALLOC (12)
#define CCV_NNC_MAX_DIM (2)

/* Command ids on this path (lib/nnc/cmd/ccv_nnc_cmd.h; SHA-256 prefixes, must not change). */
enum {
	CCV_NNC_NOOP = 0,
	CCV_NNC_ADD_FORWARD = 0x58fb3664, CCV_NNC_ADD_BACKWARD = 0x58fb3665,
	CCV_NNC_AVERAGE_POOL_FORWARD = 0x51267ab8, CCV_NNC_AVERAGE_POOL_BACKWARD = 0x51267ab9,
	CCV_NNC_BATCH_NORM_FORWARD = 0x5419819c, CCV_NNC_BATCH_NORM_BACKWARD = 0x5419819d,
	CCV_NNC_CLAMP_FORWARD = 0x2640d854, CCV_NNC_CLAMP_BACKWARD = 0x2640d855,
	CCV_NNC_COMM_ALLREDUCE_FORWARD = 0x75c8d340, CCV_NNC_COMM_ALLREDUCE_BACKWARD = 0x75c8d341,
	CCV_NNC_COMM_BROADCAST_FORWARD = 0x830eee, CCV_NNC_COMM_BROADCAST_BACKWARD = 0x830eef,
	CCV_NNC_COMM_REDUCE_FORWARD = 0x3434ead8, CCV_NNC_COMM_REDUCE_BACKWARD = 0x3434ead9,
	CCV_NNC_CONVOLUTION_FORWARD = 0x254d05f4, CCV_NNC_CONVOLUTION_BACKWARD = 0x254d05f5,
	CCV_NNC_DATATYPE_CONVERSION_FORWARD = 0xd873e38c, CCV_NNC_DATATYPE_CONVERSION_BACKWARD = 0xd873e38d,
	CCV_NNC_DATA_TRANSFER_FORWARD = 0x12d21e1a, CCV_NNC_DATA_TRANSFER_BACKWARD = 0x12d21e1b,
	CCV_NNC_DROPOUT_FORWARD = 0x7f2dc3e4, CCV_NNC_DROPOUT_BACKWARD = 0x7f2dc3e5,
	CCV_NNC_EWDIV_FORWARD = 0x1cd2fa18, CCV_NNC_EWDIV_BACKWARD = 0x1cd2fa19,
	CCV_NNC_EWEXP_FORWARD = 0xd784b170, CCV_NNC_EWEXP_BACKWARD = 0xd784b171,
	CCV_NNC_EWLOG_FORWARD = 0xf4191bf2, CCV_NNC_EWLOG_BACKWARD = 0xf4191bf3,
	CCV_NNC_EWPROD_FORWARD = 0xee07e8fe, CCV_NNC_EWPROD_BACKWARD = 0xee07e8ff,
	CCV_NNC_EWSQRT_FORWARD = 0x8870a61e, CCV_NNC_EWSQRT_BACKWARD = 0x8870a61f,
	CCV_NNC_EWSUM_FORWARD = 0xe21a2c4c, CCV_NNC_EWSUM_BACKWARD = 0xe21a2c4d,
	CCV_NNC_FORMAT_TRANSFORM_FORWARD = 0xe4a2b192, CCV_NNC_FORMAT_TRANSFORM_BACKWARD = 0xe4a2b193,
	CCV_NNC_GEMM_FORWARD = 0x7e87d00c, CCV_NNC_GEMM_BACKWARD = 0x7e87d00d,
	CCV_NNC_MAX_POOL_FORWARD = 0x7bec9360, CCV_NNC_MAX_POOL_BACKWARD = 0x7bec9361,
	CCV_NNC_MUL_FORWARD = 0x24721a46, CCV_NNC_MUL_BACKWARD = 0x24721a47,
	CCV_NNC_REDUCE_MEAN_FORWARD = 0xf23556c6, CCV_NNC_REDUCE_MEAN_BACKWARD = 0xf23556c7,
	CCV_NNC_REDUCE_SUM_FORWARD = 0x52970f06, CCV_NNC_REDUCE_SUM_BACKWARD = 0x52970f07,
	CCV_NNC_RELU_FORWARD = 0xc51eaa80, CCV_NNC_RELU_BACKWARD = 0xc51eaa81,
	CCV_NNC_SCALAR_MUL_FORWARD = 0x8b4d86aa, CCV_NNC_SCALAR_MUL_BACKWARD = 0x8b4d86ab,
	CCV_NNC_SET_FORWARD = 0x2b070804, CCV_NNC_SET_BACKWARD = 0x2b070805,
	CCV_NNC_SGD_FORWARD = 0xe650ad26, CCV_NNC_SGD_BACKWARD = 0xe650ad27,
	CCV_NNC_SOFTMAX_CROSSENTROPY_FORWARD = 0xc26b7b5e, CCV_NNC_SOFTMAX_CROSSENTROPY_BACKWARD = 0xc26b7b5f,
	CCV_NNC_TRANSPOSE_FORWARD = 0xb4d506e0, CCV_NNC_TRANSPOSE_BACKWARD = 0xb4d506e1,
	CCV_NNC_RANDOM_UNIFORM_FORWARD = 0xa0cd1d5e, CCV_NNC_RANDOM_UNIFORM_BACKWARD = 0xa0cd1d5f,
	CCV_NNC_RANDOM_NORMAL_FORWARD = 0x7062c8b4, CCV_NNC_RANDOM_NORMAL_BACKWARD = 0x7062c8b5,
	CCV_NNC_LAYER_NORM_FORWARD = 0xbed3c264, CCV_NNC_LAYER_NORM_BACKWARD = 0xbed3c265,
	CCV_NNC_RMSNORM_FORWARD = 0x6889e9d0, CCV_NNC_RMSNORM_BACKWARD = 0x6889e9d1,
	CCV_NNC_GROUP_NORM_FORWARD = 0x17deb074, CCV_NNC_GROUP_NORM_BACKWARD = 0x17deb075,
	CCV_NNC_ADAM_FORWARD = 0xe30099dc, CCV_NNC_ADAM_BACKWARD = 0xe30099dd,
	CCV_NNC_ADAMW_FORWARD = 0x4f5d4870, CCV_NNC_ADAMW_BACKWARD = 0x4f5d4871,
	CCV_NNC_ARGMAX_FORWARD = 0x68af2804, CCV_NNC_ARGMAX_BACKWARD = 0x68af2805,
	CCV_NNC_ARGMIN_FORWARD = 0xeb8747f2, CCV_NNC_ARGMIN_BACKWARD = 0xeb8747f3,
	CCV_NNC_BINARY_CROSSENTROPY_FORWARD = 0xcd2107ec, CCV_NNC_BINARY_CROSSENTROPY_BACKWARD = 0xcd2107ed,
	CCV_NNC_CATEGORICAL_CROSSENTROPY_FORWARD = 0x1eb327a2, CCV_NNC_CATEGORICAL_CROSSENTROPY_BACKWARD = 0x1eb327a3,
	CCV_NNC_GELU_FORWARD = 0xb1527ab8, CCV_NNC_GELU_BACKWARD = 0xb1527ab9,
	CCV_NNC_INDEX_SELECT_FORWARD = 0x7ee7771e, CCV_NNC_INDEX_SELECT_BACKWARD = 0x7ee7771f,
	CCV_NNC_LAMB_FORWARD = 0x450edb1a, CCV_NNC_LAMB_BACKWARD = 0x450edb1b,
	CCV_NNC_LEAKY_RELU_FORWARD = 0x507144e0, CCV_NNC_LEAKY_RELU_BACKWARD = 0x507144e1,
	CCV_NNC_MAX_FORWARD = 0xdf6f014c, CCV_NNC_MAX_BACKWARD = 0xdf6f014d,
	CCV_NNC_MIN_FORWARD = 0x972fbd26, CCV_NNC_MIN_BACKWARD = 0x972fbd27,
	CCV_NNC_MSE_FORWARD = 0x6904a9a2, CCV_NNC_MSE_BACKWARD = 0x6904a9a3,
	CCV_NNC_PAD_FORWARD = 0xd8aaca60, CCV_NNC_PAD_BACKWARD = 0xd8aaca61,
	CCV_NNC_REDUCE_MAX_FORWARD = 0x80f1a506, CCV_NNC_REDUCE_MAX_BACKWARD = 0x80f1a507,
	CCV_NNC_REDUCE_MIN_FORWARD = 0x6785ef96, CCV_NNC_REDUCE_MIN_BACKWARD = 0x6785ef97,
	CCV_NNC_REDUCE_NORM2_FORWARD = 0xb3034e16, CCV_NNC_REDUCE_NORM2_BACKWARD = 0xb3034e17,
	CCV_NNC_RMSPROP_FORWARD = 0x9c886b1c, CCV_NNC_RMSPROP_BACKWARD = 0x9c886b1d,
	CCV_NNC_MASKED_FILL_FORWARD = 0x7f992d84, CCV_NNC_MASKED_FILL_BACKWARD = 0x7f992d85,
	CCV_NNC_REDUCE_ISNAN_FORWARD = 0xee0a4ade, CCV_NNC_REDUCE_ISNAN_BACKWARD = 0xee0a4adf,
	CCV_NNC_CONVOLUTION_TRANSPOSE_FORWARD = 0xd691f78e, CCV_NNC_CONVOLUTION_TRANSPOSE_BACKWARD = 0xd691f78f,
	CCV_NNC_CMUL_FORWARD = 0xead486e6, CCV_NNC_CMUL_BACKWARD = 0xead486e7,
	CCV_NNC_COMPRESSION_LSSC_FORWARD = 0x17ea8f72, CCV_NNC_COMPRESSION_LSSC_BACKWARD = 0x17ea8f73,
	CCV_NNC_NMS_FORWARD = 0xdba26106, CCV_NNC_NMS_BACKWARD = 0xdba26107,
	CCV_NNC_ROI_ALIGN_FORWARD = 0xfef55168, CCV_NNC_ROI_ALIGN_BACKWARD = 0xfef55169,
	CCV_NNC_SCALED_DOT_PRODUCT_ATTENTION_FORWARD = 0x284ed926, CCV_NNC_SCALED_DOT_PRODUCT_ATTENTION_BACKWARD = 0x284ed927,
	CCV_NNC_SIGMOID_FORWARD = 0xf2f69650, CCV_NNC_SIGMOID_BACKWARD = 0xf2f69651,
	CCV_NNC_SIGMOID_BINARY_CROSSENTROPY_FORWARD = 0xd9e0e4a, CCV_NNC_SIGMOID_BINARY_CROSSENTROPY_BACKWARD = 0xd9e0e4b,
	CCV_NNC_SMOOTH_L1_FORWARD = 0x4e428e, CCV_NNC_SMOOTH_L1_BACKWARD = 0x4e428f,
	CCV_NNC_SOFTMAX_FORWARD = 0xc969a252, CCV_NNC_SOFTMAX_BACKWARD = 0xc969a253,
	CCV_NNC_SWISH_FORWARD = 0x583d90c2, CCV_NNC_SWISH_BACKWARD = 0x583d90c3,
	CCV_NNC_TANH_FORWARD = 0x6a62be30, CCV_NNC_TANH_BACKWARD = 0x6a62be31,
	CCV_NNC_UPSAMPLE_FORWARD = 0x73875556, CCV_NNC_UPSAMPLE_BACKWARD = 0x73875557,
	CCV_NNC_LSTM_FORWARD = 0xc5cb998c, CCV_NNC_LSTM_BACKWARD = 0xc5cb998d,
};

/* Backend slot ids (lib/nnc/cmd/ccv_nnc_backend.h). */
enum {
	CCV_NNC_NO_BACKEND = 0,
	CCV_NNC_BACKEND_CPU_OPT = 0x46deb194,
	CCV_NNC_BACKEND_CPU_REF = 0x3d9883e5,
	CCV_NNC_BACKEND_GPU_CUBLAS = 0x9b8cfed,
	CCV_NNC_BACKEND_GPU_CUDNN = 0x854b679a,
	CCV_NNC_BACKEND_GPU_NCCL = 0x7afed9c7,
	CCV_NNC_BACKEND_GPU_REF = 0x5f19790a,
	CCV_NNC_BACKEND_MPS = 0xb2f325e2,
};

/* ---------------------------------------------------------------- ABI mirrors ---- */
typedef struct { short v; } ccv_float16_t;

typedef union ccv_numeric_data_u {
	char* i8;
	unsigned char* u8;
	int* i32;
	ccv_float16_t* f16;
	float* f32;
	int64_t* i64;
	uint64_t* u64;
	double* f64;
	void* ptr;
} ccv_numeric_data_t;

typedef struct { /* 64 bytes */
	int type;     /* memory | device id << 8 */
	int format;   /* CCV_TENSOR_FORMAT_* */
	int datatype; /* CCV_32F ... */
	int reserved;
	int dim[CCV_NNC_MAX_DIM_ALLOC]; /* zero terminated */
} ccv_nnc_tensor_param_t;

typedef struct { /* 112 bytes */
	int type;
	int refcount;
	ccv_numeric_data_t data; /* raw device pointer for GPU tensors */
	off_t dataof;
	uintptr_t alias_ref;
	uint64_t data_size;
	uint64_t sig;
	ccv_nnc_tensor_param_t info;
} ccv_nnc_tensor_t;

typedef struct { /* 176 bytes; valid when (type & CCV_TENSOR_VIEW) */
	int type;
	int refcount;
	ccv_numeric_data_t data;
	off_t dataof;
	uintptr_t alias_ref;
	uint64_t data_size;
	uint64_t sig;
	ccv_nnc_tensor_param_t info;
	int contiguous;
	off_t off;
	int stride[CCV_NNC_MAX_DIM_ALLOC]; /* in elements */
} ccv_nnc_tensor_view_t;

#define CCV_IS_TENSOR_VIEW(x) ((*(const int*)(x)) & CCV_TENSOR_VIEW)
#define CCV_IS_TENSOR_CONTIGUOUS(x) (!CCV_IS_TENSOR_VIEW(x) || (((const ccv_nnc_tensor_view_t*)(x))->contiguous == 1))

typedef struct { /* 120 bytes */
	struct { int dim[CCV_NNC_MAX_DIM_ALLOC]; } size;
	union {
		struct { int count; int groups; int dilation[CCV_NNC_MAX_DIM_ALLOC]; } convolution;
		struct { int reserved; } pool;
		struct { int axis[CCV_NNC_MAX_DIM_ALLOC]; int count; float epsilon; int is_test; float momentum; } bnorm;
		struct { int nesterov; float rate; float scale; float decay; float momentum; float dampening; } sgd;
		struct { int transpose_a[2]; int transpose_b[2]; float a[3]; int flags; } blas;
		struct { float trim0; float trim1; } label_smoothing;
		struct { int axis[CCV_NNC_MAX_DIM_ALLOC]; int count; } reduce;
		struct { int axis[2]; } transpose;
		struct { float p; int entirety; } dropout;
		struct { float min; float max; } clamp;
		struct { int tanh; } gelu;
		struct { int type; float width_scale; float height_scale; int align_corners; } upsample;
		struct { int axis[CCV_NNC_MAX_DIM_ALLOC]; int count; float epsilon; int elementwise_affine; } lnorm;
		struct { int axis[CCV_NNC_MAX_DIM_ALLOC]; int count; float epsilon; } rmsnorm;
		struct { int group_axis; int reduce_axis[CCV_NNC_MAX_DIM_ALLOC]; int reduce_count; int groups; float epsilon; int elementwise_affine; } gnorm;
		struct { int type; int end[CCV_NNC_MAX_DIM_ALLOC]; } pad;
		struct { float pos_weight; } binary_crossentropy;
		struct { float beta; } smooth_l1;
		struct { int reduce_op; } mse;
		struct { float negative_slope; } leaky_relu;
		struct { int step; float rate; float scale; float beta1; float beta2; float decay; float epsilon; int amsgrad; } adam;
		struct { float rate; float scale; float decay; float alpha; float momentum; float epsilon; } rmsprop;
		struct { int step; float rate; float scale; float beta1; float beta2; float decay; float epsilon; } lamb;
		struct { float iou_threshold; } nms;
		struct { float scale; int is_causal; int flags; int deterministic; } scaled_dot_product_attention;
		struct { int hidden_size; int proj_size; int num_layers; int bias; int batch_first; int bidirectional; float dropout; int is_test; } rnn; /* ccv_nnc.h:127-136 */
		char _widest[68]; /* gnorm is the widest member in the reference (68 B) */
		void* userdata;
	};
} ccv_nnc_cmd_param_t;

typedef struct { /* 144 bytes */
	struct { int dim[CCV_NNC_MAX_DIM_ALLOC]; } stride;
	struct { int begin[CCV_NNC_MAX_DIM_ALLOC]; int end[CCV_NNC_MAX_DIM_ALLOC]; } border;
} ccv_nnc_hint_t;

typedef struct ccv_nnc_stream_context_s ccv_nnc_stream_context_t;
typedef struct ccv_nnc_stream_signal_s ccv_nnc_stream_signal_t;
typedef struct ccv_nnc_cmd_vtab_s ccv_nnc_cmd_vtab_t;

typedef struct ccv_nnc_cmd_s { /* 152 bytes */
	uint32_t cmd;
	uint32_t backend;
	int algorithm;
	ccv_nnc_cmd_param_t info;
	ccv_nnc_cmd_vtab_t* isa;
	void* data;
} ccv_nnc_cmd_t;

typedef int (*ccv_nnc_cmd_exec_f)(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context);
typedef int (*ccv_nnc_cmd_autotune_f)(const ccv_nnc_cmd_t cmd, const size_t max_workspace_size, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context);

typedef struct { /* 40 bytes */
	int tensor_formats;
	int tensor_datatypes;
	int tensor_memory;
	int algorithms;
	ccv_nnc_cmd_exec_f exec;
	ccv_nnc_cmd_autotune_f autotune;
	void* aux;
} ccv_nnc_cmd_backend_registry_t;

/* Base part of a stream context: allocated by the HOST (ccv_nnc_stream_context_new,
 * lib/nnc/ccv_nnc_stream.c:27-37) and grown in place by nnc_mi355x_init_stream_context().
 * Only `type` is read by this library; the rest belongs to the host's coroutine scheduler. */
struct ccv_nnc_stream_context_s { /* 88 bytes */
	int type;
	void* _host_private[10];
};
struct ccv_nnc_stream_signal_s { /* 16 bytes */
	int type;
	ccv_nnc_stream_context_t* emit_context;
};

typedef void (*ccv_nnc_callback_f)(void* const callback_context); /* lib/nnc/ccv_nnc.h:1005 */
typedef struct { ccv_nnc_callback_f fn; void* callback_context; } ccv_nnc_async_callback_t; /* _ccv_nnc_stream.h:57-60 */
typedef void (*ccv_nnc_async_callback_f)(ccv_nnc_async_callback_t* const async);
typedef void (*nnc_mi355x_mem_pressure_f)(int device_id, void* const context); /* cump_f, compat.h:33 */

/* ---------------------------------------------------- 3. device compat ABI -------- */
/* Each entry names the reference symbol it replaces (lib/nnc/gpu/ccv_nnc_compat.h:LINE). */
void* nnc_mi355x_malloc(int device, size_t size);                 /* cumalloc   :24 */
void  nnc_mi355x_free(int device, void* ptr);                     /* cufree     :25 */
void  nnc_mi355x_set_device(int device);                          /* cudevice   :26 */
void  nnc_mi355x_memcpy(void* dest, const int dest_type, const void* src, const int src_type, size_t n); /* cumemcpy :27, blocking */
void* nnc_mi355x_host_alloc(size_t size);                         /* cuhostalloc:28 */
void  nnc_mi355x_host_free(void* ptr);                            /* cuhostfree :29 */
int   nnc_mi355x_host_register(void* ptr, size_t size);           /* curegister :30 */
void  nnc_mi355x_host_unregister(void* ptr);                      /* cuunregister:31 */
int   nnc_mi355x_register_mem_pressure(int device_id, nnc_mi355x_mem_pressure_f func, void* const context); /* curegmp :34 */
void  nnc_mi355x_unregister_mem_pressure(const int id);           /* cuunregmp  :35 */
void  nnc_mi355x_set_profiler(int state);                         /* cusetprofiler:36 (roctx range on/off) */
int   nnc_mi355x_device_count(void);                              /* ccv_nnc_gpu_device_count :59 */

/* Stream contexts / signals: same names as the reference because the host calls exactly these. */
ccv_nnc_stream_context_t* ccv_nnc_init_stream_context(ccv_nnc_stream_context_t* const stream_context);   /* :39 (reallocs) */
void  ccv_nnc_deinit_stream_context(ccv_nnc_stream_context_t* const stream_context);                     /* :43 */
void  ccv_nnc_synchronize_stream_context(const ccv_nnc_stream_context_t* const stream_context);          /* :40 */
void* ccv_nnc_stream_compat_get_workspace(const ccv_nnc_stream_context_t* const stream_context, const size_t workspace_size, const int mem); /* :44 */
void  ccv_nnc_stream_compat_drain(ccv_nnc_stream_context_t* const stream_context);                       /* :45 */
void  ccv_nnc_stream_compat_add_callback(ccv_nnc_stream_context_t* const stream, const ccv_nnc_callback_f callback, const ccv_nnc_async_callback_f async_callback, void* const callback_context); /* :41 */
ccv_nnc_stream_signal_t* ccv_nnc_init_stream_signal(ccv_nnc_stream_signal_t* const signal);              /* :46 (reallocs) */
void  ccv_nnc_deinit_stream_signal(ccv_nnc_stream_signal_t* const signal);                               /* :49 */
void  ccv_nnc_stream_compat_emit_signal(const ccv_nnc_stream_context_t* const stream, const ccv_nnc_stream_signal_t* const signal); /* :47 */
void  ccv_nnc_stream_compat_wait_signal(const ccv_nnc_stream_context_t* const stream, const ccv_nnc_stream_signal_t* const signal); /* :48 */
int   ccv_nnc_stream_context_get_device(const ccv_nnc_stream_context_t* const stream_context);           /* :95 */
/* Returns the hipStream_t (as void*) a context launches on; NULL context = per-thread default context of the current device. */
void* nnc_mi355x_stream_context_get_stream(const ccv_nnc_stream_context_t* const stream_context);        /* ccv_nnc_stream_context_get_stream :96 */

/* Standalone constructors for callers without the reference host (mirror ccv_nnc_stream_context_new/free,
 * ccv_nnc_stream_signal_new/free: lib/nnc/ccv_nnc_stream.c:27,222,304,342). */
ccv_nnc_stream_context_t* nnc_mi355x_stream_context_new(const int type);
void nnc_mi355x_stream_context_free(ccv_nnc_stream_context_t* const stream_context);
void nnc_mi355x_stream_context_wait(const ccv_nnc_stream_context_t* const stream_context);
ccv_nnc_stream_signal_t* nnc_mi355x_stream_signal_new(const int type);
void nnc_mi355x_stream_signal_free(ccv_nnc_stream_signal_t* const signal);

/* --------------------------------------------- 2. backend registration ------------ */
/* Generic enumeration of every (cmd, backend-slot) row implemented by this library. */
int nnc_mi355x_registry_count(void);
/* Fills *cmd, *backend, *registry for row i; returns 0 on success, -1 if i is out of range. */
int nnc_mi355x_registry_get(int i, uint32_t* cmd, uint32_t* backend, ccv_nnc_cmd_backend_registry_t* registry);
/* Symbolic name of row i, e.g. "CCV_NNC_CONVOLUTION_FORWARD/CCV_NNC_BACKEND_GPU_CUDNN". */
const char* nnc_mi355x_registry_name(int i);

/* 4. Standalone dispatch: same contract as ccv_nnc_cmd_exec (lib/nnc/ccv_nnc_cmd.c:651-693):
 * cmd.backend == CCV_NNC_NO_BACKEND picks the first row whose masks cover the tensors' bits. */
int nnc_mi355x_cmd_exec(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context);
int nnc_mi355x_cmd_ok(const uint32_t cmd, const uint32_t backend); /* ccv_nnc_cmd_ok, ccv_nnc_cmd.c:117 */

/* Multi-process data parallel (one process per GPU): join an RCCL communicator created from a
 * 128-byte unique id that rank 0 obtained from nnc_mi355x_comm_unique_id() and shipped to the other
 * ranks out of band (torch.distributed store in bench.py).  After this, the COMM_* commands operate
 * across processes with one tensor per process.  The single-process N-device form (the reference's
 * ncclCommInitAll, lib/nnc/gpu/ccv_nnc_compat.cu:1404-1445) needs no call. */
int nnc_mi355x_comm_unique_id(void* id_out_128_bytes);
int nnc_mi355x_comm_init_rank(const void* id_128_bytes, int rank, int world_size);
void nnc_mi355x_comm_destroy(void);
/* Deployment (b) (one process per GPU): overlap the gradient all-reduces with the backward pass.  The reference's single-process data parallelism gets the
 * overlap from its scheduler -- an all-reduce node behind every gradient, on other streams (lib/nnc/ccv_nnc_symbolic_graph_parallel.c:545-575) --; a process
 * that drives the model API (ccv_cnnp_model_backward, then ccv_cnnp_model_parameter_gradients_map(COMM_ALLREDUCE_FORWARD)) issues them all behind the whole
 * backward pass.  With the mode on (1; or NNC_MI355X_COMM_OVERLAP=1 in the environment; -1 = the environment decides) those all-reduces go out on a
 * communication stream, sorted by the order their gradients were written and cut into buckets of NNC_MI355X_COMM_BUCKET_MB (32), each bucket behind its own
 * gradients' writers only; every other stream joins them at its next launch / synchronise / signal.  Results are those of immediate issue (the same collectives,
 * grouped differently).  Gradients must be written by CONVOLUTION_ / GEMM_ / BATCH_NORM_BACKWARD (anything else takes the issuing stream's order). */
void nnc_mi355x_comm_overlap(int on);
void nnc_mi355x_comm_overlap_stats(long* collectives, long* buckets);
/* Ranks of the process communicator as RCCL counts them (ncclCommCount); 0 before nnc_mi355x_comm_init_rank.  bench.py prints it as `rccl_ranks`. */
int nnc_mi355x_comm_count(void);
/* Counters of the COMM commands' coalescing (cmd_comm.cpp): per-device collectives issued so far, and the RCCL groups they
 * travelled in (consecutive COMM commands share one group). */
void nnc_mi355x_comm_stats(long* collectives, long* groups);

/* ------------------------------------------------ 5. classic image pre-process loops, the data pipeline's batch kernels and its pinned staging ring:
 * include/nnc_mi355x_pipeline.h (a header of its own: host-side glue includes it NEXT TO the reference's own headers, integration/nnc_mi355x_dataframe.c) */
#include "nnc_mi355x_pipeline.h"

/* HIP-event timing on the stream a context launches on (bench.py roofline leg). */
void* nnc_mi355x_event_new(void);
void  nnc_mi355x_event_record(void* event, const ccv_nnc_stream_context_t* const stream_context);
float nnc_mi355x_event_elapsed_ms(void* start, void* stop); /* synchronizes on stop */
void  nnc_mi355x_event_free(void* event);
/* Per-kernel timing: while enabled, every contraction (conv / GEMM) kernel launch is bracketed by a HIP event pair on
 * its own stream and filed with its algorithmic FLOPs and problem dims (M, N, K, groups/batch, split-K slices).
 * enable(1) clears the table; get() synchronizes on the record's stop event. */
void nnc_mi355x_profile_enable(int on);
int  nnc_mi355x_profile_count(void);
int  nnc_mi355x_profile_get(int i, char* name, int name_len, double* flops, double* bytes, float* ms, int dims[5]);
/* Test hook: force the contraction block tile to (64*wm) x (64*wn) for every later conv / GEMM launch of the process
 * ((2,2) (2,1) (1,2) (1,1) exist); (0,0) restores the built-in choice.  Used by the parity tests to cover every
 * tile shape on every loader at sizes the oracle finishes. */
void nnc_mi355x_debug_force_tile(int wm, int wn);
void nnc_mi355x_debug_force_splits(int splits);
/* Bytes of reserved space LSTM_FORWARD writes (output 3) and LSTM_BACKWARD reads (input 12) -- the function both LSTM rows carry in registry->aux,
 * which the host's shape inference calls through ccv_nnc_cmd_aux (lib/nnc/cmd/rnn/ccv_nnc_lstm.c:35,64-71); replaces
 * _ccv_nnc_lstm_reserve_space_size / cudnnGetRNNTempSpaceSizes (lib/nnc/cmd/rnn/gpu/ccv_nnc_lstm_gpu_cudnn.cu:17-48).  0 when cmd.info.rnn.is_test. */
size_t nnc_mi355x_lstm_reserve_space_size(const ccv_nnc_cmd_t cmd, int datatype, int feature_size, int batch_count, int max_seq_count);
/* Palettized tensors (datatype = CCV_QX | qbits << 8 | palette datatype >> 12, info.reserved = elements per block; lib/nnc/ccv_nnc_easy.h:210-238).
 * nnc_mi355x_depalettize replaces ccv_nnc_compat_depalettize (lib/nnc/gpu/ccv_nnc_compat.h:59, lib/nnc/gpu/ccv_nnc_palettize.cu:321-469): `input` is the byte
 * stream the host's ccv_nnc_palettize wrote (lib/nnc/ccv_nnc_palettize.c:9-208), in device memory; `output` receives output_length elements of `datatype`
 * (CCV_16F / CCV_32F / CCV_64F), bit for bit what lib/nnc/ccv_nnc_palettize.c:211-956 produces on the CPU.  Returns CCV_NNC_EXEC_SUCCESS or _INVALID (qbits
 * outside 4 .. 8, an input_length shorter than nnc_mi355x_palettized_bytes).  The GEMM, convolution and transposed-convolution rows accept CCV_QX inputs
 * and run on dense images of them, DATA_TRANSFER moves the byte stream (the rows that list CCV_QX in the reference). */
int nnc_mi355x_depalettize(const void* input, int datatype, size_t input_length, int qbits, int number_in_blocks, void* output, size_t output_length, ccv_nnc_stream_context_t* stream_context);
/* ccv_nnc_tensor_data_size_without_padding of a palettized tensor (lib/nnc/ccv_nnc_easy.h:220-238). */
size_t nnc_mi355x_palettized_bytes(int datatype, size_t count, int qbits, int number_in_blocks);
/* Opt-in fusion for callers that know a CONVOLUTION_FORWARD's only consumer is the RELU_FORWARD behind it (the reference's graphs run
 * that ReLU in place, test/int/nnc/graph.vgg.d.tests.c:80): cmd.algorithm = NNC_MI355X_CONV_ALGO_FUSE_RELU | a, a = 0 .. 2 or 0xff for the
 * backend's choice, makes the command write max(0, conv + bias); the RELU_FORWARD may then be dropped.  Applied in the epilogue of the
 * fused Winograd kernel, the Winograd output transform and the 3-channel kernel; as one more pass behind the other paths.
 * ccv_nnc_cmd_autotune never produces such a value.  The symmetric change a maintainer would make in the host: a
 * (CONVOLUTION_FORWARD, RELU_FORWARD) entry in ccv_nnc_ops_fusions[] (lib/nnc/ccv_nnc_symbolic_graph_simplify.c:595-). */
#define NNC_MI355X_CONV_ALGO_FUSE_RELU 0x100
/* The same on the way back, for callers that know the gradient a command writes goes through a RELU_BACKWARD (in place) next and that
 * the command's input a IS that ReLU's output (the reference's VGG-D graph: every convolution and pooling reads a rectified map):
 *   MAX_POOL_BACKWARD (g, a, b) -> h with cmd.algorithm = NNC_MI355X_POOL_ALGO_FUSE_RELU_BACKWARD: h = (a == b's window max && a > 0) ? g : 0,
 *   i.e. MAX_POOL_BACKWARD followed by RELU_BACKWARD (h, -, a) -> h; the RELU_BACKWARD may then be dropped.  The kernels read a anyway
 *   and mask as they write: no extra traffic, one pass over the map less. */
#define NNC_MI355X_POOL_ALGO_FUSE_RELU_BACKWARD 0x100
/* BATCH_NORM_FORWARD with cmd.algorithm = NNC_MI355X_BNORM_ALGO_FUSE_RELU: y = max(0, batch norm) -- the in-place RELU_FORWARD of a
 * conv - bn - relu block applied in the pass that writes y (statistics are of x: unchanged). */
#define NNC_MI355X_BNORM_ALGO_FUSE_RELU 0x100
/* EWSUM_FORWARD with cmd.algorithm = NNC_MI355X_EWSUM_ALGO_FUSE_RELU: c = max(0, a + b + ...) -- the in-place RELU_FORWARD behind the residual sum of a
 * ResNet block (bin/nnc/imagenet.c: ccv_cnnp_sum then ccv_cnnp_relu) applied in the pass that writes the sum. */
#define NNC_MI355X_EWSUM_ALGO_FUSE_RELU 0x100
/* EWSUM_FORWARD with cmd.algorithm = NNC_MI355X_EWSUM_ALGO_FUSE_RELU_BACKWARD: the LAST input is a mask map b (a ReLU's forward output), the others are summed:
 * c = b > 0 ? a0 + a1 : 0 -- the gradient sum at the head of a residual block followed by the in-place RELU_BACKWARD of the block before it (the backward pass
 * of bin/nnc/imagenet.c's bottlenecks issues exactly that pair), in one pass.  Two summands only.  The look-ahead (peephole.cpp) sets it for the unmodified host. */
#define NNC_MI355X_EWSUM_ALGO_FUSE_RELU_BACKWARD 0x200
/* Callers that do NOT set these bits get the same folding from a one-command look-ahead (ccv_amd/csrc/peephole.cpp): a
 * CONVOLUTION_FORWARD / CONVOLUTION_BACKWARD / MAX_POOL_BACKWARD whose like has run before is recorded instead of launched; the
 * in-place RELU_FORWARD / RELU_BACKWARD the reference's graphs issue next on the same stream completes it, anything else that could
 * observe the stream's order launches it as it was.  Same results either way.  On by default; NNC_MI355X_PEEPHOLE=0 in the
 * environment or nnc_mi355x_set_peephole(0) turns it off (set_peephole launches what is recorded first). */
void nnc_mi355x_set_peephole(int on);
/* Test hook: commands recorded so far, how many of them a ReLU completed (folded), how many were launched as they were (plain). */
void nnc_mi355x_debug_peephole_counts(long* recorded, long* folded, long* plain);
/* The look-ahead's TRAIL (round 5): the reference's static schedule issues the signal emit behind a CONVOLUTION_BACKWARD, the waits of the layer's two
 * SGD_FORWARD streams, those commands and their emits BEFORE the RELU_BACKWARD that completes the convolution (lib/nnc/ccv_nnc_graph_run.c:581-675).  Such
 * operations are kept behind the recorded command, in arrival order, and replayed right behind it when it launches; nothing else can observe the difference
 * (peephole.cpp).  NNC_MI355X_PEEPHOLE_TRAIL=0 restores the flush at the first of them.  Test hook: operations that have waited in a trail so far. */
long nnc_mi355x_debug_peephole_trailed(void);
/* Device memory (nnc_mi355x_malloc / _free = cumalloc / cufree, lib/nnc/gpu/ccv_nnc_compat.cu:101-141; the layer lib/nnc/ccv_nnc_xpu_alloc.c sits on): freed
 * blocks are kept per device and (rounded) size and handed out again (device_rt.cpp).  The free is STREAM-ORDERED (round 6): it records an event behind every
 * stream of the device that still has work in flight and returns -- no device drain --; the block is handed out again once those events have completed (an
 * allocation that finds only unfinished blocks of its size waits for the oldest one's).  Under memory pressure every kept block goes back to the driver, the
 * host's curegmp callbacks run, the allocation is retried.  Freeing a block twice aborts.  NNC_MI355X_POOL_ALLOC=0 selects plain hipMalloc / hipFree.
 * Hook: allocations served from kept blocks, pressure retries, bytes held (kept + handed out) and bytes handed out. */
void nnc_mi355x_debug_pool_counts(long* allocs, long* retries, long* reserved_bytes, long* used_bytes);
/* The kept bytes are bounded per device (a quarter of the device's memory; NNC_MI355X_POOL_KEEP_MB overrides): beyond the cap the oldest kept blocks go back to
 * the driver.  Hook: blocks returned that way so far. */
long nnc_mi355x_debug_pool_trimmed(void);
/* Every kept block of `device` (< 0: of every device) back to the driver now: for a host that is about to let another allocator of the process (a communicator
 * library, a second framework) use the device's memory.  The library calls it itself before it creates RCCL communicators and when one of its own direct
 * allocations fails. */
void nnc_mi355x_pool_trim(int device);
/* ---- HIP-graph capture of a compiled schedule (the reference walks the schedule node by node on one host thread for every step and
 * device: lib/nnc/ccv_nnc_graph_run.c:581-675 _ccv_nnc_graph_exec_run_loop, :686-843 _ccv_nnc_graph_topsorted_run_coro).  The host brackets ONE step -- any
 * sequence of enqueue-only calls on `stream`: ccv_nnc_graph_run, ccv_cnnp_model_fit / _evaluate / _backward / _apply_gradients, ccv_nnc_cmd_exec -- and replays it:
 *     nnc_mi355x_capture_begin(stream);  ccv_cnnp_model_fit(model, ..., stream);  void* step = nnc_mi355x_capture_end(stream);
 *     for (...) nnc_mi355x_graph_launch(step, stream);          nnc_mi355x_graph_free(step);
 * Nothing executes between begin and end; the schedule's other streams join through the signals the host emits and waits for.  Rules: capture after a warm-up
 * step (compilation, autotune and first allocations are over); the tensors the step names keep their addresses while the graph lives (new batches are copied INTO
 * the bound inputs); no wait for / blocking copy out of a recording stream (the runtime's error stops the process); one capture at a time.  Cluster batch norm,
 * DROPOUT and the LSTM's dropout are replay-safe (a fresh mask per replay); device memory freed meanwhile is set aside until the graphs that may name it are freed.
 * capture_begin: 0, or -1 (not a GPU stream context, a capture already running, NNC_MI355X_POOL_ALLOC=0).  capture_end: the graph, or NULL with the runtime's
 * message on stderr (a stream of the step did not join back, an operation invalidated the capture).  graph_launch: 0 / -1. */
int nnc_mi355x_capture_begin(ccv_nnc_stream_context_t* stream_context);
void* nnc_mi355x_capture_end(ccv_nnc_stream_context_t* stream_context);
int nnc_mi355x_graph_launch(void* graph, ccv_nnc_stream_context_t* stream_context);
int nnc_mi355x_graph_node_count(void* graph);
void nnc_mi355x_graph_free(void* graph);
/* The form of the next captures.  0 (default): the streams of the recording device that the step reaches are folded into the recording stream (issue order
 * is a valid order of the step; the graph is one chain).  1 (also NNC_MI355X_CAPTURE_STREAMS=1): the streams join as HIP streams and the graph keeps the
 * schedule's branches -- only for steps whose side streams never wait for one another's signals: with the reference's schedules ROCm 7.2's
 * hipStreamEndCapture recurses without end (device_rt.cpp "HIP-graph capture"). */
void nnc_mi355x_capture_keep_streams(int on);
/* Hook: bytes of freed device memory currently set aside for captured graphs. */
long nnc_mi355x_debug_pool_parked_bytes(void);
/* Hook: events recorded by frees (a free that finds every stream idle records none) and allocations that had to wait for a kept block's last users. */
void nnc_mi355x_debug_pool_fences(long* events, long* waits);
/* batch-norm commands that ran on the cluster kernels (cmd_norm.cpp: a cluster of workgroups per channel keeps the channel in registers between the
 * statistics and the apply pass; NNC_MI355X_BN_CLUSTER=0 / nnc_mi355x_tune_set("BN_CLUSTER", 0) selects the plane kernels). */
long nnc_mi355x_debug_bn_cluster_launches(void);
/* commands that have reached an exec function of this library since it was loaded (tools/host_resnet_bench.c divides the host's enqueue time by it) */
long nnc_mi355x_debug_exec_count(void);
/* Test hook for the CCV_16F datapath (half_stage.cpp): how many half-precision tensors have been given an fp32 image so far
 * (staged) and how many were handed to a kernel as halves (native) since the library was loaded. */
void nnc_mi355x_debug_half_counts(long* staged, long* native);
/* Performance tunables (policy only -- results do not depend on them beyond floating-point re-association): by name, e.g.
 * "WINO_SLICE_KB"; the environment variable NNC_MI355X_<NAME> sets the same value at first use.  Returns 0 / -1 (unknown). */
int  nnc_mi355x_tune_set(const char* name, long value);
long nnc_mi355x_tune_get(const char* name);
/* Name of the device kernel the last command on this thread launched for its dominant work
 * (conv/gemm contraction), for matching against rocprofv3 kernel-trace rows. */
const char* nnc_mi355x_last_kernel_name(void);
const char* nnc_mi355x_version(void);

#ifdef __cplusplus
}
#endif

/* The registration entry points themselves (one per row) are declared in
 * nnc_mi355x_registry.h, generated from ccv_amd/csrc/registry.def. */
#include "nnc_mi355x_registry.h"

#endif
