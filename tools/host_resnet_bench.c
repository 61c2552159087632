/* ResNet-50 v1d training step THROUGH THE REFERENCE HOST's model API (BASELINE config 4): a client of lib/nnc/ccv_nnc.h +
 * ccv_cnnp_* that assembles the network the reference's trainer uses (bin/nnc/imagenet.c:17-98: three 3x3 stem convolutions,
 * max pool, bottleneck layers {64 x3, 128 x4, 256 x6, 512 x3} with expansion 4, average-pool + 1x1 projection shortcuts,
 * global average pool, dense 1000, softmax), NCHW tensors like the trainer (imagenet.c:354), compiles it with the trainer's
 * minimizer and loss (CMD_SGD_FORWARD(1, lr, 1 / batch, wd, 0.9, 0), categorical cross-entropy on smoothed one-hot labels,
 * imagenet.c:314-317, 357, 395) and times ccv_cnnp_model_fit on synthetic data.  Everything above ccv_nnc_cmd_exec -- cnnp,
 * symbolic graph, autodiff, simplify, compile, the multi-stream scheduler -- is the reference's unmodified code; this file is
 * the benchmark driver only.  Built by oracle/build_ref_host.sh against libccv_host_gpu.so (and the CPU-emulator build for the
 * small-size test of the CPU tier).
 *   host_resnet_bench.gpu <batch> <input hw> <steps> <warmup> <32|16> [mini|dawn]      -> one JSON line
 * 16 = CCV_16F tensors, what the trainer itself runs (imagenet.c:344); 32 = CCV_32F (BASELINE config 4).
 * dawn = BASELINE config 5's network instead: the CIFAR-10 "DawnNet" of bin/nnc/cifar-10.c:76-127 (3x3 convolutions 64-128-256-512 with
 * batch norm + ReLU, 2x2 max pools, two residual pairs, global max pool, dense 10), 32 x 32 inputs, and that trainer's own step
 * (cifar-10.c:259-273): ccv_cnnp_model_evaluate(requires_grad) -> SOFTMAX_CROSSENTROPY forward / backward on the outputs ->
 * ccv_cnnp_model_backward -> ccv_cnnp_model_apply_gradients, compiled with CMD_SGD_FORWARD(1, lr, 1 / batch, 0.01, 0.9, 0) and no loss.
 *
 * Whole-network parity (tests/test_via_host.py): built a second time with -DHOST_BENCH_CPU against oracle/_ref/libccv_ref.so -- the SAME
 * host on the reference's own CPU backends, every tensor in CPU memory (host_resnet_bench.cpu).  With HOST_BENCH_CHECK=1 both builds set every
 * parameter from the counter hash before the first step (the two backends' random generators differ, ccv_cnnp_model_set_parameter makes the
 * start identical) and print, after step 1: the per-image loss, the outputs' sum / sum of squares, and sum / sum of squares of every updated
 * parameter.  The test compares the two lines. */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE /* RTLD_NEXT (CPU build) */
#endif
#include <ccv.h>
#include <nnc/ccv_nnc.h>
#include <nnc/ccv_nnc_easy.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <sys/time.h>

/* Tensor layout.  The trainer keeps NCHW (imagenet.c:354) and so does the GPU build; the reference's CPU backends have no NCHW pooling and no
 * NCHW convolution gradient (pool/ccv_nnc_max_pool_cpu_ref.c:145, convolution/ccv_nnc_conv_cpu_ref.c:358 register NHWC only), so the CPU build --
 * and any build under HOST_BENCH_FORMAT=nhwc -- runs the same network on NHWC tensors holding the SAME numbers: images and 4-d filters are filled
 * through the NCHW linear index of each element (g_nhwc), sums over a tensor do not depend on its layout, and both networks reduce the map to 1 x 1
 * before the dense layer, so every compared quantity is the same function of the same parameters. */
static int g_nhwc = 0;
static ccv_nnc_tensor_param_t tensor4(ccv_nnc_tensor_param_t p, const int n, const int c, const int h, const int w)
{
	if (g_nhwc) { p.format = CCV_TENSOR_FORMAT_NHWC; p.dim[0] = n; p.dim[1] = h; p.dim[2] = w; p.dim[3] = c; }
	return p;
}
/* position in the tensor's own layout of the element whose NCHW linear index is j (dims n, c, h, w) */
static size_t layout_index(const size_t j, const int c, const int h, const int w)
{
	if (!g_nhwc) return j;
	const size_t x = j % w, y = (j / w) % h, ch = (j / ((size_t)w * h)) % c, n = j / ((size_t)w * h * c);
	return ((n * h + y) * w + x) * c + ch;
}
#ifdef HOST_BENCH_CPU
/* the reference's CPU backends: tensors in CPU memory, no stream, no launch records */
#define DEV_TENSOR_NCHW(...) CPU_TENSOR_NCHW(32F, __VA_ARGS__)
/* The oracle of this repository is the reference's CPU_REF backend (north star: "outputs match the reference CPU backend").  Left alone, the host
 * would pick CPU_OPT for the convolutions: its 1 x 1 path needs a BLAS library the image does not have (convolution/cpu_opt/_ccv_nnc_conv_cpu_gemm.c:35
 * -> lib/ccv_algebra.c:340 asserts) and its direct path disagrees with CPU_REF, numpy and this backend on strided, padded layers
 * (tests/test_oracle_pin.py::test_cpu_opt_direct_convolution_differs_on_strided_padded_layers).  The executable therefore answers the two CPU_OPT
 * registrations itself with empty rows -- the dynamic linker resolves the host's call to these -- and ccv_nnc_cmd_find_backend falls through to CPU_REF. */
void _register_command_CCV_NNC_CONVOLUTION_FORWARD_backend_CCV_NNC_BACKEND_CPU_OPT(void* const registry) {}
void _register_command_CCV_NNC_CONVOLUTION_BACKWARD_backend_CCV_NNC_BACKEND_CPU_OPT(void* const registry) {}
/* The reference's CPU pooling loops walk ONE image: no batch loop in pool/ccv_nnc_max_pool_cpu_ref.c:37-63 / ccv_nnc_avg_pool_cpu_ref.c (the GPU
 * backend being replaced pools the whole batch, as cuDNN does; SURVEY.md section 7 and ccv_amd/vgg.py's pool_per_image are the same finding).
 * The four CPU_REF pooling rows are therefore wrapped the same way: the reference's own registration runs, then its exec function is called once
 * per image on 3-d tensors carved out of the 4-d ones -- the reference's loops do all the arithmetic. */
#include <dlfcn.h>
#include <nnc/ccv_nnc_internal.h>
#define POOL_PER_IMAGE(CMD, slot) \
static ccv_nnc_cmd_exec_f pool_exec_##slot; \
static int pool_per_image_##slot(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context) \
{ \
	int i, n, batch = 1; \
	for (i = 0; i < input_size; i++) if (inputs[i] && ccv_nnc_tensor_nd(inputs[i]->info.dim) == 4) batch = inputs[i]->info.dim[0]; \
	if (batch == 1) return pool_exec_##slot(cmd, hint, flags, inputs, input_size, outputs, output_size, stream_context); \
	ccv_nnc_tensor_t ti[8], to[8]; \
	ccv_nnc_tensor_t* pi[8]; ccv_nnc_tensor_t* po[8]; \
	for (n = 0; n < batch; n++) { \
		for (i = 0; i < input_size + output_size; i++) { \
			ccv_nnc_tensor_t* const src = i < input_size ? inputs[i] : outputs[i - input_size]; \
			ccv_nnc_tensor_t* const dst = i < input_size ? &ti[i] : &to[i - input_size]; \
			if (i < input_size) pi[i] = src ? dst : 0; else po[i - input_size] = src ? dst : 0; \
			if (!src) continue; \
			if (CCV_IS_TENSOR_VIEW(src) || ccv_nnc_tensor_nd(src->info.dim) != 4 || src->info.dim[0] != batch) { fprintf(stderr, "host_resnet_bench: pooling tensor is not a dense 4-d batch\n"); abort(); } \
			*dst = *src; \
			dst->info.dim[0] = src->info.dim[1]; dst->info.dim[1] = src->info.dim[2]; dst->info.dim[2] = src->info.dim[3]; dst->info.dim[3] = 0; \
			dst->data.f32 = src->data.f32 + (size_t)n * src->info.dim[1] * src->info.dim[2] * src->info.dim[3]; \
		} \
		const int r = pool_exec_##slot(cmd, hint, flags, pi, input_size, po, output_size, stream_context); \
		if (r != CCV_NNC_EXEC_SUCCESS) return r; \
	} \
	return CCV_NNC_EXEC_SUCCESS; \
} \
void _register_command_##CMD##_backend_CCV_NNC_BACKEND_CPU_REF(ccv_nnc_cmd_backend_registry_t* const registry) \
{ \
	void (*real)(ccv_nnc_cmd_backend_registry_t* const) = (void (*)(ccv_nnc_cmd_backend_registry_t* const))dlsym(RTLD_NEXT, "_register_command_" #CMD "_backend_CCV_NNC_BACKEND_CPU_REF"); \
	if (!real) { fprintf(stderr, "host_resnet_bench: the reference's " #CMD " CPU_REF registration is missing\n"); abort(); } \
	real(registry); \
	pool_exec_##slot = registry->exec; \
	registry->exec = pool_per_image_##slot; \
}
POOL_PER_IMAGE(CCV_NNC_MAX_POOL_FORWARD, 0)
POOL_PER_IMAGE(CCV_NNC_MAX_POOL_BACKWARD, 1)
POOL_PER_IMAGE(CCV_NNC_AVERAGE_POOL_FORWARD, 2)
POOL_PER_IMAGE(CCV_NNC_AVERAGE_POOL_BACKWARD, 3)
static void nnc_mi355x_profile_enable(int on) {}
static int nnc_mi355x_profile_count(void) { return 0; }
static int nnc_mi355x_profile_get(int i, char* name, int name_len, double* flops, double* bytes, float* ms, int dims[5]) { return -1; }
static long nnc_mi355x_debug_exec_count(void) { return 0; }
static int nnc_mi355x_comm_init_rank(const void* id, int rank, int world) { return -1; }
static int nnc_mi355x_comm_count(void) { return 0; }
static void nnc_mi355x_comm_destroy(void) {}
static void nnc_mi355x_comm_overlap_stats(long* collectives, long* buckets) { *collectives = *buckets = 0; }
static int nnc_mi355x_capture_begin(ccv_nnc_stream_context_t* s) { return -1; }
static void* nnc_mi355x_capture_end(ccv_nnc_stream_context_t* s) { return 0; }
static int nnc_mi355x_graph_launch(void* g, ccv_nnc_stream_context_t* s) { return -1; }
static int nnc_mi355x_graph_node_count(void* g) { return 0; }
static void nnc_mi355x_graph_free(void* g) {}
#else
/* HOST_BENCH_DEVICE: the device this process trains on (the process-per-GPU form with every GPU visible: rank r takes device r) */
static int g_device = 0;
static ccv_nnc_tensor_param_t on_device(ccv_nnc_tensor_param_t p) { CCV_TENSOR_SET_DEVICE_ID(p.type, g_device); return p; }
#define DEV_TENSOR_NCHW(...) on_device(GPU_TENSOR_NCHW(000, 32F, __VA_ARGS__))
/* the backend's in-library launch records (include/nnc_mi355x.h) */
void nnc_mi355x_profile_enable(int on);
int nnc_mi355x_profile_count(void);
int nnc_mi355x_profile_get(int i, char* name, int name_len, double* flops, double* bytes, float* ms, int dims[5]);
long nnc_mi355x_debug_exec_count(void);
int nnc_mi355x_comm_init_rank(const void* id_128_bytes, int rank, int world_size);
int nnc_mi355x_comm_count(void);
void nnc_mi355x_comm_destroy(void);
void nnc_mi355x_comm_overlap_stats(long* collectives, long* buckets); /* NNC_MI355X_COMM_OVERLAP=1: the gradient all-reduces that went out in buckets beside the backward pass */
/* HIP-graph capture of the step (include/nnc_mi355x.h; HOST_BENCH_CAPTURE=1): the two calls a host adds around ONE step, and the replay */
int nnc_mi355x_capture_begin(ccv_nnc_stream_context_t* stream_context);
void* nnc_mi355x_capture_end(ccv_nnc_stream_context_t* stream_context);
int nnc_mi355x_graph_launch(void* graph, ccv_nnc_stream_context_t* stream_context);
int nnc_mi355x_graph_node_count(void* graph);
void nnc_mi355x_graph_free(void* graph);
#endif

static float hash_unit(const uint64_t i, const uint64_t seed)
{
	uint64_t h = (i + 1) * 0x9E3779B97F4A7C15ull ^ (seed + 1) * 0xD1B54A32D192ED03ull;
	h ^= h >> 30; h *= 0xBF58476D1CE4E5B9ull;
	h ^= h >> 27; h *= 0x94D049BB133111EBull;
	h ^= h >> 31;
	return (float)(h >> 40) * (1.0f / 16777216.0f);
}

/* HOST_BENCH_CHECK's parameter j of tensor i (j = the element's NCHW linear index): filters He-uniform, the dense matrix a tenth of that (moderate
 * logits: a saturated softmax would compare denormal probabilities), vectors (batch-norm scale / bias, biases) in 0.75 .. 1.25 */
static float check_init(const int i, const size_t j, const int nd, const size_t fan)
{
	const float bound = sqrtf(6.f / (float)fan) * (nd == 2 ? 0.1f : 1.f);
	return nd >= 2 ? (hash_unit(j, 5000 + i) * 2 - 1) * bound : 0.75f + 0.5f * hash_unit(j, 5000 + i);
}

static double now_ms(void)
{
	struct timeval tv;
	gettimeofday(&tv, 0);
	return tv.tv_sec * 1e3 + tv.tv_usec * 1e-3;
}

/* no_bias as the trainer has it: the stem and the projection shortcuts without, the bottleneck convolutions with (imagenet.c:28-46, 75-82) */
static ccv_cnnp_model_t* conv_bn(const int filters, const int k, const int stride, const int relu, const int no_bias)
{
	ccv_cnnp_model_t* m[3];
	int n = 0;
	m[n++] = ccv_cnnp_convolution(1, filters, DIM_ALLOC(k, k), DIM_ALLOC(), no_bias, HINT((stride, stride), (k / 2, k / 2)), 0, 1, 0);
	m[n++] = ccv_cnnp_batch_norm(0.9, 1e-4, 1, 0);
	if (relu) m[n++] = ccv_cnnp_relu(0);
	return ccv_cnnp_sequential_new(m, n, 1, 0);
}

/* one bottleneck: 1x1 -> 3x3 (stride) -> 1x1 x expansion, + shortcut (identity, or average pool + 1x1 projection), ReLU */
static ccv_cnnp_model_t* bottleneck(const int filters, const int expansion, const int stride, const int projection)
{
	const ccv_cnnp_model_io_t in = ccv_cnnp_input();
	ccv_cnnp_model_io_t shortcut = in;
	if (projection) {
		if (stride > 1) shortcut = ccv_cnnp_model_apply(ccv_cnnp_average_pool(DIM_ALLOC(stride, stride), HINT((stride, stride), (0, 0)), 0), MODEL_IO_LIST(in));
		shortcut = ccv_cnnp_model_apply(ccv_cnnp_convolution(1, filters * expansion, DIM_ALLOC(1, 1), DIM_ALLOC(), 1, HINT((1, 1), (0, 0)), 0, 1, 0), MODEL_IO_LIST(shortcut));
	}
	ccv_cnnp_model_io_t out = ccv_cnnp_model_apply(conv_bn(filters, 1, 1, 1, 0), MODEL_IO_LIST(in));
	out = ccv_cnnp_model_apply(conv_bn(filters, 3, stride, 1, 0), MODEL_IO_LIST(out));
	out = ccv_cnnp_model_apply(conv_bn(filters * expansion, 1, 1, 0, 0), MODEL_IO_LIST(out));
	out = ccv_cnnp_model_apply(ccv_cnnp_sum(0), MODEL_IO_LIST(out, shortcut));
	out = ccv_cnnp_model_apply(ccv_cnnp_relu(0), MODEL_IO_LIST(out));
	return ccv_cnnp_model_new(MODEL_IO_LIST(in), MODEL_IO_LIST(out), 1, 0);
}

static ccv_cnnp_model_t* resnet(const int* const blocks, const int* const widths, const int stages, const int stem, const int classes)
{
	const ccv_cnnp_model_io_t in = ccv_cnnp_input();
	ccv_cnnp_model_io_t out = ccv_cnnp_model_apply(conv_bn(stem / 2, 3, 2, 1, 1), MODEL_IO_LIST(in));
	out = ccv_cnnp_model_apply(conv_bn(stem / 2, 3, 1, 1, 1), MODEL_IO_LIST(out));
	out = ccv_cnnp_model_apply(conv_bn(stem, 3, 1, 1, 1), MODEL_IO_LIST(out));
	out = ccv_cnnp_model_apply(ccv_cnnp_max_pool(DIM_ALLOC(3, 3), HINT((2, 2), (1, 1)), 0), MODEL_IO_LIST(out));
	int s, b;
	for (s = 0; s < stages; s++)
		for (b = 0; b < blocks[s]; b++)
			out = ccv_cnnp_model_apply(bottleneck(widths[s], 4, (b == 0 && s > 0) ? 2 : 1, b == 0), MODEL_IO_LIST(out));
	out = ccv_cnnp_model_apply(ccv_cnnp_average_pool(DIM_ALLOC(0, 0), ccv_nnc_no_hint, 0), MODEL_IO_LIST(out));
	out = ccv_cnnp_model_apply(ccv_cnnp_flatten(0), MODEL_IO_LIST(out));
	out = ccv_cnnp_model_apply(ccv_cnnp_dense(classes, 0, 0, 1, 0), MODEL_IO_LIST(out));
	out = ccv_cnnp_model_apply(ccv_cnnp_softmax(0), MODEL_IO_LIST(out));
	return ccv_cnnp_model_new(MODEL_IO_LIST(in), MODEL_IO_LIST(out), 1, 0);
}

static ccv_cnnp_model_t* dawn_conv(const int filters)
{
	return ccv_cnnp_sequential_new(MODEL_LIST(
		ccv_cnnp_convolution(1, filters, DIM_ALLOC(3, 3), DIM_ALLOC(), 0, HINT((1, 1), (1, 1)), 0, 1, 0),
		ccv_cnnp_batch_norm(0.9, 1e-4, 1, 0),
		ccv_cnnp_relu(0)), 1, 0);
}
static ccv_cnnp_model_t* dawn_layer(const int filters, const int residual)
{
	const ccv_cnnp_model_io_t in = ccv_cnnp_input();
	ccv_cnnp_model_io_t out = ccv_cnnp_model_apply(dawn_conv(filters), MODEL_IO_LIST(in));
	out = ccv_cnnp_model_apply(ccv_cnnp_max_pool(DIM_ALLOC(2, 2), HINT((2, 2), (0, 0)), 0), MODEL_IO_LIST(out));
	if (residual) {
		const ccv_cnnp_model_io_t shortcut = out;
		out = ccv_cnnp_model_apply(dawn_conv(filters), MODEL_IO_LIST(out));
		out = ccv_cnnp_model_apply(dawn_conv(filters), MODEL_IO_LIST(out));
		out = ccv_cnnp_model_apply(ccv_cnnp_sum(0), MODEL_IO_LIST(out, shortcut));
	}
	return ccv_cnnp_model_new(MODEL_IO_LIST(in), MODEL_IO_LIST(out), 1, 0);
}
static ccv_cnnp_model_t* dawn(void)
{
	return ccv_cnnp_sequential_new(MODEL_LIST(
		dawn_conv(64), dawn_layer(128, 1), dawn_layer(256, 0), dawn_layer(512, 1),
		ccv_cnnp_max_pool(DIM_ALLOC(0, 0), ccv_nnc_no_hint, 0),
		ccv_cnnp_flatten(0),
		ccv_cnnp_dense(10, 0, 0, 1, 0)), 1, 0);
}

int main(int argc, char** argv)
{
	const int batch = argc > 1 ? atoi(argv[1]) : 256, hw = argc > 2 ? atoi(argv[2]) : 224;
	const int steps = argc > 3 ? atoi(argv[3]) : 4, warmup = argc > 4 ? atoi(argv[4]) : 1;
	const int half = argc > 5 && atoi(argv[5]) == 16;
	const int mini = argc > 6 && strcmp(argv[6], "mini") == 0;
	const int is_dawn = argc > 6 && strcmp(argv[6], "dawn") == 0;
	/* argv[7]: devices.  > 1 = the reference's single-process data parallelism (ccv_cnnp_model_set_data_parallel,
	 * lib/nnc/ccv_cnnp_model.c; the graph is replicated per device by ccv_nnc_symbolic_graph_data_parallel and every parameter
	 * gradient all-reduced with COMM_ALLREDUCE): `batch` images PER DEVICE, as bin/nnc/imagenet.c:314-317 drives it. */
	const int devices = argc > 7 ? atoi(argv[7]) : 1;
	/* argv[8]: rotation of the shards over the devices (device d gets shard (d + rot) % devices).  A test runs rot = 0 and rot = 1:
	 * replicas whose gradients are really summed hold the same parameters either way, so a shard's outputs must not depend on
	 * which device it ran on. */
	const int rot = argc > 8 ? atoi(argv[8]) : 0;
	if (devices < 1 || devices > 8) { fprintf(stderr, "devices must be 1..8\n"); return 2; }
	/* ONE PROCESS PER GPU (round 4): HOST_BENCH_WORLD = P processes, this one HOST_BENCH_RANK, RCCL bootstrapped from the 128-byte id in HOST_BENCH_COMM_ID
	 * (hex; bench.py's launcher makes it with nnc_mi355x_comm_unique_id and hands it to every rank).  Each process drives ONE device with its own host thread
	 * -- the reference's single-process form has one thread enqueue for all N devices, which is what bounds configs 4-f16 / 5 at 8 GPUs --
	 * and the replicas meet in the reference's own multi-stage training API: evaluate, the loss commands, ccv_cnnp_model_backward, then
	 * ccv_cnnp_model_parameter_gradients_map(COMM_ALLREDUCE_FORWARD) -- one in-place all-reduce per parameter gradient through this backend's COMM row
	 * (cmd_comm.cpp deployment (b): the communicator spans the processes; the back-to-back commands leave as ONE RCCL group) -- and
	 * ccv_cnnp_model_apply_gradients with the minimizer's scale 1 / (batch x P).  HOST_BENCH_WORLD=1 with HOST_BENCH_COMM_ID set runs the same code on a
	 * communicator of one (what a one-GPU box and the emulator can check). */
	const int world = getenv("HOST_BENCH_WORLD") ? atoi(getenv("HOST_BENCH_WORLD")) : 1, rank = getenv("HOST_BENCH_RANK") ? atoi(getenv("HOST_BENCH_RANK")) : 0;
	const char* const comm_id = getenv("HOST_BENCH_COMM_ID");
	const int ranks = comm_id && *comm_id ? 1 : 0; /* the process-per-GPU form */
#ifndef HOST_BENCH_CPU
	g_device = getenv("HOST_BENCH_DEVICE") ? atoi(getenv("HOST_BENCH_DEVICE")) : 0;
	if (g_device != 0 && devices != 1) { fprintf(stderr, "HOST_BENCH_DEVICE with the single-process N-device form\n"); return 2; }
#endif
	if (ranks) {
		unsigned char id[128];
		int k;
		if (devices != 1 || world < 1 || rank < 0 || rank >= world || strlen(comm_id) != 256) { fprintf(stderr, "host_resnet_bench: process-per-GPU mode wants devices = 1, 0 <= rank < world and a 256-digit id\n"); return 2; }
		for (k = 0; k < 128; k++) { unsigned v = 0; sscanf(comm_id + 2 * k, "%2x", &v); id[k] = (unsigned char)v; }
		const int r = nnc_mi355x_comm_init_rank(id, rank, world);
		if (r != 0) { fprintf(stderr, "host_resnet_bench: nnc_mi355x_comm_init_rank failed (%d)\n", r); return 3; }
	}
#ifdef HOST_BENCH_CPU
	g_nhwc = 1;
#else
	g_nhwc = getenv("HOST_BENCH_FORMAT") && strcmp(getenv("HOST_BENCH_FORMAT"), "nhwc") == 0;
#endif
	const int dt = half ? CCV_16F : CCV_32F;
	static const int blocks50[] = { 3, 4, 6, 3 }, widths50[] = { 64, 128, 256, 512 };
	static const int blocks_m[] = { 1, 1 }, widths_m[] = { 8, 16 };
	const int classes = (mini || is_dawn) ? 10 : 1000;
	ccv_nnc_init();
	ccv_cnnp_model_t* const model = is_dawn ? dawn() : mini ? resnet(blocks_m, widths_m, 2, 8, classes) : resnet(blocks50, widths50, 4, 64, classes);
	ccv_nnc_tensor_param_t input = tensor4(DEV_TENSOR_NCHW(batch, 3, hw, hw), batch, 3, hw, hw);
	input.datatype = dt;
	const float lr = 0.01f, wd = 0.0001f;
	if (is_dawn) ccv_cnnp_model_compile(model, &input, 1, CMD_SGD_FORWARD(1, lr, 1. / (batch * devices * world), 0.01, 0.9, 0), CMD_NOOP());
	else ccv_cnnp_model_compile(model, &input, 1, CMD_SGD_FORWARD(1, lr, 1. / (batch * devices * world), wd, 0.9, 0), ranks ? CMD_NOOP() : CMD_CATEGORICAL_CROSSENTROPY_FORWARD());
	if (devices > 1) ccv_cnnp_model_set_data_parallel(model, devices);
	/* synthetic batch: images ~ U(-1, 1) (normalised pixels), labels as the trainer's smoothed one-hot rows (eta = 0.1) */
	ccv_nnc_tensor_t* const hx = ccv_nnc_tensor_new(0, tensor4(CPU_TENSOR_NCHW(32F, batch, 3, hw, hw), batch, 3, hw, hw), 0);
	ccv_nnc_tensor_t* const hfit = ccv_nnc_tensor_new(0, CPU_TENSOR_NCHW(32F, batch, classes), 0);
	size_t j;
	const size_t nx = (size_t)batch * 3 * hw * hw;
	const int shard0 = ranks ? rank : rot % devices; /* (process-per-GPU: every rank its own shard) */
	for (j = 0; j < nx; j++) hx->data.f32[layout_index(j, 3, hw, hw)] = hash_unit(j, 2000 + 10 * shard0) * 2 - 1;
	const float eta = 0.1f;
	int i;
	for (i = 0; i < batch; i++) {
		const int c = (int)(hash_unit(i, 2001 + 10 * shard0) * classes);
		int k;
		for (k = 0; k < classes; k++) hfit->data.f32[(size_t)i * classes + k] = (k == c ? 1 - eta : 0) + eta / classes;
	}
	ccv_nnc_tensor_param_t xp = tensor4(DEV_TENSOR_NCHW(batch, 3, hw, hw), batch, 3, hw, hw), fp = DEV_TENSOR_NCHW(batch, classes);
	xp.datatype = dt; fp.datatype = dt;
	if (g_nhwc) fp.format = CCV_TENSOR_FORMAT_NHWC;
	ccv_nnc_tensor_t* const x = ccv_nnc_tensor_new(0, xp, 0);
	ccv_nnc_tensor_t* const fit = ccv_nnc_tensor_new(0, fp, 0);
	ccv_nnc_tensor_t* const out = ccv_nnc_tensor_new(0, fp, 0);
	if (half) {
		ccv_nnc_tensor_t* const hx16 = ccv_nnc_tensor_new(0, tensor4(CPU_TENSOR_NCHW(16F, batch, 3, hw, hw), batch, 3, hw, hw), 0);
		ccv_nnc_tensor_t* const hfit16 = ccv_nnc_tensor_new(0, CPU_TENSOR_NCHW(16F, batch, classes), 0);
		ccv_nnc_cmd_exec(CMD_DATATYPE_CONVERSION_FORWARD(), ccv_nnc_no_hint, 0, TENSOR_LIST(hx, hfit), TENSOR_LIST(hx16, hfit16), 0);
		ccv_nnc_cmd_exec(CMD_DATA_TRANSFER_FORWARD(), ccv_nnc_no_hint, 0, TENSOR_LIST(hx16, hfit16), TENSOR_LIST(x, fit), 0);
		ccv_nnc_tensor_free(hx16);
		ccv_nnc_tensor_free(hfit16);
	} else
		ccv_nnc_cmd_exec(CMD_DATA_TRANSFER_FORWARD(), ccv_nnc_no_hint, 0, TENSOR_LIST(hx, hfit), TENSOR_LIST(x, fit), 0);
	/* devices 1 .. n-1: their own shards */
	ccv_nnc_tensor_t* xs[8]; ccv_nnc_tensor_t* fits[8]; ccv_nnc_tensor_t* outs[8];
	xs[0] = x; fits[0] = fit; outs[0] = out;
	{
		int d;
		for (d = 1; d < devices; d++) {
			ccv_nnc_tensor_param_t xd = xp, fd = fp;
			CCV_TENSOR_SET_DEVICE_ID(xd.type, d); CCV_TENSOR_SET_DEVICE_ID(fd.type, d);
			xs[d] = ccv_nnc_tensor_new(0, xd, 0); fits[d] = ccv_nnc_tensor_new(0, fd, 0); outs[d] = ccv_nnc_tensor_new(0, fd, 0);
			const int shard = (d + rot) % devices;
			for (j = 0; j < nx; j++) hx->data.f32[layout_index(j, 3, hw, hw)] = hash_unit(j, 2000 + 10 * shard) * 2 - 1;
			for (i = 0; i < batch; i++) {
				const int c = (int)(hash_unit(i, 2001 + 10 * shard) * classes);
				int k;
				for (k = 0; k < classes; k++) hfit->data.f32[(size_t)i * classes + k] = (k == c ? 1 - eta : 0) + eta / classes;
			}
			if (half) {
				ccv_nnc_tensor_t* const hx16 = ccv_nnc_tensor_new(0, tensor4(CPU_TENSOR_NCHW(16F, batch, 3, hw, hw), batch, 3, hw, hw), 0);
				ccv_nnc_tensor_t* const hfit16 = ccv_nnc_tensor_new(0, CPU_TENSOR_NCHW(16F, batch, classes), 0);
				ccv_nnc_cmd_exec(CMD_DATATYPE_CONVERSION_FORWARD(), ccv_nnc_no_hint, 0, TENSOR_LIST(hx, hfit), TENSOR_LIST(hx16, hfit16), 0);
				ccv_nnc_cmd_exec(CMD_DATA_TRANSFER_FORWARD(), ccv_nnc_no_hint, 0, TENSOR_LIST(hx16, hfit16), TENSOR_LIST(xs[d], fits[d]), 0);
				ccv_nnc_tensor_free(hx16);
				ccv_nnc_tensor_free(hfit16);
			} else
				ccv_nnc_cmd_exec(CMD_DATA_TRANSFER_FORWARD(), ccv_nnc_no_hint, 0, TENSOR_LIST(hx, hfit), TENSOR_LIST(xs[d], fits[d]), 0);
		}
	}
#ifdef HOST_BENCH_CPU
	ccv_nnc_stream_context_t* const stream = 0;
#else
	int stream_type = CCV_STREAM_CONTEXT_GPU;
	CCV_STREAM_SET_DEVICE_ID(stream_type, g_device);
	ccv_nnc_stream_context_t* const stream = ccv_nnc_stream_context_new(stream_type);
#endif
	/* reproducible parameter initialisation: the host seeds its generators from a thread-local ADDRESS otherwise (ccv_nnc_stream.c:262-281) */
	ccv_nnc_stream_context_set_seed(0, 20240923);
	if (stream) ccv_nnc_stream_context_set_seed(stream, 20240924);
	/* HOST_BENCH_CHECK=1: identical parameters on every backend.  Filters / matrices: uniform in +-sqrt(6 / fan-in) (He); vectors (batch-norm
	 * scales and biases, convolution / dense biases): 0.75 .. 1.25 -- the backends' own initialisers differ in their generators. */
	const int check = getenv("HOST_BENCH_CHECK") && atoi(getenv("HOST_BENCH_CHECK"));
	int param_count = 0;
	if (check) {
		/* the parameter tensors exist only after a first run (ccv_cnnp_model_parameter_tensor_params asserts it): one test-mode forward pass
		 * allocates and randomly initialises them (batch-norm running statistics are not touched in test mode), then every one is overwritten;
		 * the training step that follows re-compiles the graph in its own mode with zeroed momentum */
		ccv_cnnp_model_evaluate(model, (ccv_cnnp_evaluate_param_t){ .is_test = 1 }, xs, devices, outs, devices, 0, stream);
		ccv_nnc_stream_context_wait(stream);
		param_count = ccv_cnnp_model_parameter_count(model);
		for (i = 0; i < param_count; i++) {
			const ccv_cnnp_model_io_t pio = ccv_cnnp_model_parameters(model, -1, i);
			ccv_nnc_tensor_param_t pp = ccv_cnnp_model_parameter_tensor_params(model, pio);
			const int pdt = pp.datatype;
			pp.type = CCV_TENSOR_CPU_MEMORY; pp.datatype = CCV_32F;
			ccv_nnc_tensor_t* const hp = ccv_nnc_tensor_new(0, pp, 0);
			const size_t cnt = ccv_nnc_tensor_count(pp);
			const int nd = ccv_nnc_tensor_nd(pp.dim);
			size_t fan = 1;
			int a;
			for (a = 1; a < nd; a++) fan *= pp.dim[a];
			/* a 4-d filter is [K][C][kh][kw] on NCHW tensors and [K][kh][kw][C] on NHWC ones (the fan-in is the product of the last three either way) */
			const int permute = nd == 4 && pp.format == CCV_TENSOR_FORMAT_NHWC;
			for (j = 0; j < cnt; j++) {
				const float v = check_init(i, j, nd, fan);
				hp->data.f32[permute ? layout_index(j, pp.dim[3], pp.dim[1], pp.dim[2]) : j] = v;
			}
			if (pdt == CCV_16F) {
				pp.datatype = CCV_16F;
				ccv_nnc_tensor_t* const hp16 = ccv_nnc_tensor_new(0, pp, 0);
				ccv_nnc_cmd_exec(CMD_DATATYPE_CONVERSION_FORWARD(), ccv_nnc_no_hint, 0, TENSOR_LIST(hp), TENSOR_LIST(hp16), 0);
				ccv_cnnp_model_set_parameter(model, pio, hp16);
				ccv_nnc_tensor_free(hp16);
			} else
				ccv_cnnp_model_set_parameter(model, pio, hp);
			ccv_nnc_tensor_free(hp);
		}
	}
	/* dawn: the CIFAR trainer's step (cifar-10.c:259-273); labels are class indices in fp32, the softmax / gradient tensors have the outputs' type */
	ccv_nnc_tensor_t* labels_d[8]; ccv_nnc_tensor_t* softmax_d[8]; ccv_nnc_tensor_t* grad_d[8];
	{
		ccv_nnc_tensor_t* const hl = ccv_nnc_tensor_new(0, CPU_TENSOR_NCHW(32F, batch), 0);
		int d;
		for (d = 0; d < devices; d++) {
			ccv_nnc_tensor_param_t lp = DEV_TENSOR_NCHW(batch), sp = fp;
			CCV_TENSOR_SET_DEVICE_ID(lp.type, d); CCV_TENSOR_SET_DEVICE_ID(sp.type, d);
			labels_d[d] = ccv_nnc_tensor_new(0, lp, 0); softmax_d[d] = ccv_nnc_tensor_new(0, sp, 0); grad_d[d] = ccv_nnc_tensor_new(0, sp, 0);
			const int shard = ranks ? rank : (d + rot) % devices;
			for (i = 0; i < batch; i++) hl->data.f32[i] = (float)(int)(hash_unit(i, 2001 + 10 * shard) * classes);
			ccv_nnc_cmd_exec(CMD_DATA_TRANSFER_FORWARD(), ccv_nnc_no_hint, 0, TENSOR_LIST(hl), TENSOR_LIST(labels_d[d]), 0);
		}
		ccv_nnc_tensor_free(hl);
	}
	ccv_nnc_tensor_t* const labels = labels_d[0];
	ccv_nnc_tensor_t* const softmax = softmax_d[0];
	ccv_nnc_tensor_t* const grad = grad_d[0];
	/* N devices: the CIFAR trainer's own spelling (cifar-10.c:259-273) -- every call without a stream, the loss commands once per device on that
	 * device's tensors, ccv_cnnp_model_backward over the N gradients; the host all-reduces the parameter gradients (COMM_ALLREDUCE rows) */
	ccv_nnc_stream_context_t* const step_stream = (is_dawn && devices > 1) ? 0 : stream;
	/* end of a timed region: the stream, and -- the N-device DawnNet step runs without one -- every device's legacy stream (a blocking device-to-host
	 * copy of a few bytes per device orders behind everything queued there) */
	ccv_nnc_tensor_t* const sync_host = ccv_nnc_tensor_new(0, CPU_TENSOR_NCHW(32F, batch), 0);
	ccv_nnc_tensor_t* const barrier_t = ccv_nnc_tensor_new(0, DEV_TENSOR_NCHW(16), 0);
	ccv_nnc_cmd_exec(CMD_SET_FORWARD(0), ccv_nnc_no_hint, 0, TENSOR_LIST(), TENSOR_LIST(barrier_t), 0);
#define SYNC_ALL() do { \
		if (ranks) ccv_nnc_cmd_exec(CMD_COMM_ALLREDUCE_FORWARD(), ccv_nnc_no_hint, 0, TENSOR_LIST(barrier_t), TENSOR_LIST(barrier_t), stream); /* every rank's queue has reached this point */ \
		ccv_nnc_stream_context_wait(stream); \
		if (!step_stream) { int d_; for (d_ = 0; d_ < devices; d_++) ccv_nnc_cmd_exec(CMD_DATA_TRANSFER_FORWARD(), ccv_nnc_no_hint, 0, TENSOR_LIST(labels_d[d_]), TENSOR_LIST(sync_host), 0); } \
	} while (0)
#define ALLREDUCE_GRADIENTS() do { \
		if (ranks) ccv_cnnp_model_parameter_gradients_map(model, ccv_cnnp_model_parameters(model, ALL_PARAMETERS, ALL_PARAMETERS), CMD_COMM_ALLREDUCE_FORWARD(), ccv_nnc_no_hint, 0, 0, 0, 0, 0, step_stream); \
	} while (0)
#define TRAIN_STEP() do { \
		if (is_dawn) { \
			int d_; \
			ccv_cnnp_model_evaluate(model, (ccv_cnnp_evaluate_param_t){ .requires_grad = 1, .disable_outgrad = CCV_CNNP_DISABLE_OUTGRAD_ALL }, xs, devices, outs, devices, 0, step_stream); \
			for (d_ = 0; d_ < devices; d_++) { \
				ccv_nnc_cmd_exec(CMD_SOFTMAX_CROSSENTROPY_FORWARD(), ccv_nnc_no_hint, 0, TENSOR_LIST(outs[d_], labels_d[d_]), TENSOR_LIST(0, softmax_d[d_]), step_stream); \
				ccv_nnc_cmd_exec(CMD_SOFTMAX_CROSSENTROPY_BACKWARD(), ccv_nnc_no_hint, 0, TENSOR_LIST(0, 0, outs[d_], labels_d[d_], 0, softmax_d[d_]), TENSOR_LIST(grad_d[d_], 0), step_stream); \
			} \
			ccv_cnnp_model_backward(model, grad_d, devices, TENSOR_LIST(), 0, step_stream); \
			ALLREDUCE_GRADIENTS(); \
			ccv_cnnp_model_apply_gradients(model, step_stream); \
		} else if (ranks) { /* the ResNet trainer's step in the multi-stage form: the loss the fit call compiles in, issued by hand on the softmax outputs.  NB in f16 this \
		                     * UNFUSED loss (-1 / p on half-precision softmax outputs, then the softmax's own backward) is not stable: the full ResNet-50 holds non-finite \
		                     * parameters after two steps (profiles/r06_v12_force_comm_f16.txt) where the fit call -- whose graph the host simplifies to SOFTMAX_CROSSENTROPY -- \
		                     * and the DawnNet branch above (SOFTMAX_CROSSENTROPY by hand) stay finite; fp32 is fine. */ \
			ccv_cnnp_model_evaluate(model, (ccv_cnnp_evaluate_param_t){ .requires_grad = 1, .disable_outgrad = CCV_CNNP_DISABLE_OUTGRAD_ALL }, xs, 1, outs, 1, 0, step_stream); \
			ccv_nnc_cmd_exec(CMD_CATEGORICAL_CROSSENTROPY_BACKWARD(), ccv_nnc_no_hint, 0, TENSOR_LIST(0, outs[0], fits[0]), TENSOR_LIST(grad_d[0]), step_stream); \
			ccv_cnnp_model_backward(model, grad_d, 1, TENSOR_LIST(), 0, step_stream); \
			ALLREDUCE_GRADIENTS(); \
			ccv_cnnp_model_apply_gradients(model, step_stream); \
		} else \
			ccv_cnnp_model_fit(model, xs, devices, fits, devices, outs, devices, 0, stream); \
	} while (0)
	/* step 1: compiles the graph (autodiff, simplify, arena, schedule) and initialises the parameters */
	const double t_first0 = now_ms();
	TRAIN_STEP();
	SYNC_ALL();
	const double t_first = now_ms() - t_first0;
	/* the first step's softmax outputs: finite, rows summing to one (read back in fp32) */
	ccv_nnc_tensor_t* const hout = ccv_nnc_tensor_new(0, CPU_TENSOR_NCHW(32F, batch, classes), 0);
	if (half) {
		ccv_nnc_tensor_t* const hout16 = ccv_nnc_tensor_new(0, CPU_TENSOR_NCHW(16F, batch, classes), 0);
		ccv_nnc_cmd_exec(CMD_DATA_TRANSFER_FORWARD(), ccv_nnc_no_hint, 0, TENSOR_LIST(is_dawn ? softmax : out), TENSOR_LIST(hout16), 0);
		ccv_nnc_cmd_exec(CMD_DATATYPE_CONVERSION_FORWARD(), ccv_nnc_no_hint, 0, TENSOR_LIST(hout16), TENSOR_LIST(hout), 0);
		ccv_nnc_tensor_free(hout16);
	} else
		ccv_nnc_cmd_exec(CMD_DATA_TRANSFER_FORWARD(), ccv_nnc_no_hint, 0, TENSOR_LIST(is_dawn ? softmax : out), TENSOR_LIST(hout), 0);
	double row0 = 0, worst = 0;
	int finite = 1;
	for (i = 0; i < batch; i++) {
		double s = 0;
		int k;
		for (k = 0; k < classes; k++) { const float v = hout->data.f32[(size_t)i * classes + k]; if (!(v == v) || v < 0 || v > 1.001f) finite = 0; s += v; }
		if (i == 0) row0 = s;
		if (fabs(s - 1) > worst) worst = fabs(s - 1);
	}
	if (check) { /* after step 1: per-image loss -sum_k fit[k] log(out[k]) (dawn: -log softmax[label]), outputs, every updated parameter */
		printf("{\"check\": {\"loss\": [");
		for (i = 0; i < batch && i < 8; i++) {
			double l = 0;
			int k;
			if (is_dawn) { const int c = (int)(hash_unit(i, 2001) * classes); l = -log((double)hout->data.f32[(size_t)i * classes + c]); }
			else for (k = 0; k < classes; k++) l -= (double)hfit->data.f32[(size_t)i * classes + k] * log((double)hout->data.f32[(size_t)i * classes + k] + 1e-30);
			printf("%s%.9g", i ? ", " : "", l);
		}
		double osum = 0, osumsq = 0;
		for (j = 0; j < (size_t)batch * classes; j++) { osum += hout->data.f32[j]; osumsq += (double)hout->data.f32[j] * hout->data.f32[j]; }
		printf("], \"out_sum\": %.12g, \"out_sumsq\": %.12g, \"out_first\": [", osum, osumsq);
		for (i = 0; i < classes && i < 10; i++) printf("%s%.9g", i ? ", " : "", hout->data.f32[i]);
		printf("], \"param_count\": %d, \"param_sum\": [", param_count);
		double* const psq = (double*)malloc(sizeof(double) * (param_count + 1));
		double* const pmax = (double*)malloc(sizeof(double) * (param_count + 1));
		double* const dsum = (double*)malloc(sizeof(double) * (param_count + 1));
		double* const dsq = (double*)malloc(sizeof(double) * (param_count + 1));
		for (i = 0; i < param_count; i++) {
			const ccv_cnnp_model_io_t pio = ccv_cnnp_model_parameters(model, -1, i);
			ccv_nnc_tensor_param_t pp = ccv_cnnp_model_parameter_tensor_params(model, pio);
			const int pdt = pp.datatype;
			pp.type = CCV_TENSOR_CPU_MEMORY; pp.datatype = CCV_32F;
			ccv_nnc_tensor_t* const hp = ccv_nnc_tensor_new(0, pp, 0);
			if (pdt == CCV_16F) {
				pp.datatype = CCV_16F;
				ccv_nnc_tensor_t* const hp16 = ccv_nnc_tensor_new(0, pp, 0);
				ccv_cnnp_model_parameter_copy(model, pio, hp16);
				ccv_nnc_cmd_exec(CMD_DATATYPE_CONVERSION_FORWARD(), ccv_nnc_no_hint, 0, TENSOR_LIST(hp16), TENSOR_LIST(hp), 0);
				ccv_nnc_tensor_free(hp16);
			} else
				ccv_cnnp_model_parameter_copy(model, pio, hp);
			const size_t cnt = ccv_nnc_tensor_count(pp);
			/* the step itself: updated - initial (the sums above are dominated by the initial values; these are the gradient's) */
			const int nd = ccv_nnc_tensor_nd(pp.dim);
			size_t fan = 1;
			int ax;
			for (ax = 1; ax < nd; ax++) fan *= pp.dim[ax];
			const int permute = nd == 4 && pp.format == CCV_TENSOR_FORMAT_NHWC;
			double a = 0, b = 0, m = 0, da = 0, db = 0;
			for (j = 0; j < cnt; j++) {
				const double v = hp->data.f32[permute ? layout_index(j, pp.dim[3], pp.dim[1], pp.dim[2]) : j];
				float v0 = check_init(i, j, nd, fan);
				if (pdt == CCV_16F) { uint16_t h16; ccv_float_to_half_precision(&v0, &h16, 1); ccv_half_precision_to_float(&h16, &v0, 1); }
				a += v; b += v * v; if (fabs(v) > m) m = fabs(v);
				da += v - v0; db += (v - v0) * (v - v0);
			}
			psq[i] = b; pmax[i] = m; dsum[i] = da; dsq[i] = db;
			printf("%s%.12g", i ? ", " : "", a);
			ccv_nnc_tensor_free(hp);
		}
		printf("], \"param_sumsq\": [");
		for (i = 0; i < param_count; i++) printf("%s%.12g", i ? ", " : "", psq[i]);
		printf("], \"param_absmax\": [");
		for (i = 0; i < param_count; i++) printf("%s%.9g", i ? ", " : "", pmax[i]);
		printf("], \"step_sum\": [");
		for (i = 0; i < param_count; i++) printf("%s%.9g", i ? ", " : "", dsum[i]);
		printf("], \"step_sumsq\": [");
		for (i = 0; i < param_count; i++) printf("%s%.9g", i ? ", " : "", dsq[i]);
		printf("]}, \"devices\": %d, \"dtype\": \"%s\", \"format\": \"%s\"}\n", devices, half ? "f16" : "f32", g_nhwc ? "NHWC" : "NCHW");
		free(psq); free(pmax); free(dsum); free(dsq);
		return 0; /* the check line is the whole output: nothing is timed in this mode */
	}
	for (i = 1; i < warmup; i++) TRAIN_STEP();
	SYNC_ALL();
	/* HOST_BENCH_CAPTURE=1: the step -- whatever the branch above issues: ccv_cnnp_model_fit, or the CIFAR trainer's evaluate / loss / backward / apply_gradients
	 * sequence -- is recorded ONCE between nnc_mi355x_capture_begin / _end (nothing executes meanwhile) and every timed step is one nnc_mi355x_graph_launch.
	 * The same number of steps EXECUTES either way, so the probes below must not depend on the switch (tests/test_via_host.py). */
	void* step_graph = 0;
	double t_capture = 0, ms_issued = 0, cap_host_ms[5] = { 0, 0, 0, 0, 0 };
	const int capmode = getenv("HOST_BENCH_CAPTURE") ? atoi(getenv("HOST_BENCH_CAPTURE")) : 0; /* 1: every timed step is a replay; 2: BOTH forms are timed, command by command first (bench.py) */
	if (capmode == 2) {
		const double p0_ = now_ms();
		for (i = 0; i < steps; i++) TRAIN_STEP();
		SYNC_ALL();
		ms_issued = (now_ms() - p0_) / (steps > 0 ? steps : 1);
	}
	if (capmode) {
		if (!step_stream || ranks) { fprintf(stderr, "host_resnet_bench: HOST_BENCH_CAPTURE needs the one-stream step (no process ranks, not the N-device DawnNet form)\n"); return 4; }
		const double c0_ = now_ms();
		if (nnc_mi355x_capture_begin(stream) != 0) { fprintf(stderr, "host_resnet_bench: nnc_mi355x_capture_begin failed\n"); return 4; }
		TRAIN_STEP();
		step_graph = nnc_mi355x_capture_end(stream);
		if (!step_graph) { fprintf(stderr, "host_resnet_bench: nnc_mi355x_capture_end failed\n"); return 4; }
		t_capture = now_ms() - c0_;
	}
#define RUN_STEP() do { if (step_graph) { if (nnc_mi355x_graph_launch(step_graph, stream) != 0) { fprintf(stderr, "host_resnet_bench: nnc_mi355x_graph_launch failed\n"); exit(4); } } else TRAIN_STEP(); } while (0)
	const double t0 = now_ms();
	for (i = 0; i < steps; i++) RUN_STEP();
	SYNC_ALL();
	const double ms = (now_ms() - t0) / (steps > 0 ? steps : 1);
	/* per device: sum and sum of squares of the softmax outputs of the last timed step (replicas fed the same shard must agree
	 * exactly, and -- the all-reduced gradient of identical shards being the single-device gradient -- with a one-device run) */
	double dev_sum[8], dev_sumsq[8];
	for (i = 0; i < devices; i++) {
		ccv_nnc_tensor_t* const src = (is_dawn && i == 0) ? softmax : outs[i];
		if (half) {
			ccv_nnc_tensor_t* const h16 = ccv_nnc_tensor_new(0, CPU_TENSOR_NCHW(16F, batch, classes), 0);
			ccv_nnc_cmd_exec(CMD_DATA_TRANSFER_FORWARD(), ccv_nnc_no_hint, 0, TENSOR_LIST(src), TENSOR_LIST(h16), 0);
			ccv_nnc_cmd_exec(CMD_DATATYPE_CONVERSION_FORWARD(), ccv_nnc_no_hint, 0, TENSOR_LIST(h16), TENSOR_LIST(hout), 0);
			ccv_nnc_tensor_free(h16);
		} else
			ccv_nnc_cmd_exec(CMD_DATA_TRANSFER_FORWARD(), ccv_nnc_no_hint, 0, TENSOR_LIST(src), TENSOR_LIST(hout), 0);
		double a = 0, b = 0;
		for (j = 0; j < (size_t)batch * classes; j++) { a += hout->data.f32[j]; b += (double)hout->data.f32[j] * hout->data.f32[j]; }
		dev_sum[i] = a; dev_sumsq[i] = b;
	}
	/* host-enqueue leg (round 4): how long ONE host thread needs to put a step into the streams.  Each step starts on drained streams, so nothing the GPU
	 * does can hold the host back; the step call returning = everything enqueued (the host's scheduler runs the graph asynchronously on the stream).
	 * With the reference's single-process data parallelism one thread enqueues for all N devices: N x this number has to stay below the GPU time of a step. */
	double enq_ms[5], enq_step_ms[5];
	long enq_cmds = 0;
	for (i = 0; i < 5; i++) { enq_ms[i] = enq_step_ms[i] = 0; }
#ifndef HOST_BENCH_CPU /* (the CPU build runs its commands inside the call: nothing to separate, and each step takes seconds) */
	for (i = 0; i < 5; i++) {
		SYNC_ALL();
		const long c0 = nnc_mi355x_debug_exec_count();
		const double e0 = now_ms();
		if (capmode == 2) TRAIN_STEP(); else RUN_STEP(); /* (both forms: this leg is the step issued command by command, the replay's host time follows) */
		const double e1 = now_ms();
		SYNC_ALL();
		enq_ms[i] = e1 - e0; enq_step_ms[i] = now_ms() - e0;
		enq_cmds = nnc_mi355x_debug_exec_count() - c0;
	}
	for (i = 0; i < 5 && step_graph && capmode == 1; i++) cap_host_ms[i] = enq_ms[i]; /* (the leg above WAS the replay; the same number of steps executes as without the switch) */
	for (i = 0; i < 5 && step_graph && capmode == 2; i++) { /* host time of ONE nnc_mi355x_graph_launch on drained streams */
		SYNC_ALL();
		const double e0 = now_ms();
		RUN_STEP();
		cap_host_ms[i] = now_ms() - e0;
		SYNC_ALL();
	}
	{ int a_, b_; for (a_ = 0; a_ < 5; a_++) for (b_ = a_ + 1; b_ < 5; b_++) if (cap_host_ms[b_] < cap_host_ms[a_]) { const double t_ = cap_host_ms[a_]; cap_host_ms[a_] = cap_host_ms[b_]; cap_host_ms[b_] = t_; } }
#endif
	{ int a_, b_; for (a_ = 0; a_ < 5; a_++) for (b_ = a_ + 1; b_ < 5; b_++) if (enq_ms[b_] < enq_ms[a_]) { double t_ = enq_ms[a_]; enq_ms[a_] = enq_ms[b_]; enq_ms[b_] = t_; t_ = enq_step_ms[a_]; enq_step_ms[a_] = enq_step_ms[b_]; enq_step_ms[b_] = t_; } }
	/* roofline leg: one more step with the backend's per-launch HIP-event records on (contractions and batch norm) */
	nnc_mi355x_profile_enable(1);
	TRAIN_STEP();
	SYNC_ALL();
	struct { char name[192]; double ms, flops, bytes; int n; } agg[64];
	int nagg = 0;
	/* the half-precision contractions of the step by what bounds each LAUNCH: algorithmic FLOP per algorithmic byte under the machine balance
	 * (2.5 PFLOP/s / 8 TB/s = 312) = HBM-bound (the 1 x 1 convolutions with 64 .. 512 channels), else matrix-pipe-bound */
	struct { double ms, flops, bytes; int n; } f16b[2] = { { 0, 0, 0, 0 }, { 0, 0, 0, 0 } }; /* [0] mfma-bound, [1] hbm-bound */
	const int nrec = nnc_mi355x_profile_count();
	FILE* const recf = getenv("HOST_BENCH_RECORDS") ? fopen(getenv("HOST_BENCH_RECORDS"), "w") : 0; /* every launch of that step: name, dims, ms, TFLOP/s, GB/s */
	for (i = 0; i < nrec; i++) {
		char name[256];
		double fl, by;
		float rms;
		int dims[5], k;
		nnc_mi355x_profile_get(i, name, 256, &fl, &by, &rms, dims);
		if (recf) fprintf(recf, "%4d %-150s dims %d %d %d %d %d  %8.4f ms  %7.1f TFLOP/s  %7.1f GB/s\n", i, name, dims[0], dims[1], dims[2], dims[3], dims[4], rms, rms > 0 ? fl / (rms * 1e-3) / 1e12 : 0.0, rms > 0 ? by / (rms * 1e-3) / 1e9 : 0.0);
		if (strstr(name, "mfma_gemm_f16") && fl > 0 && by > 0 && rms > 0) {
			const int hb = fl / by < 2.5e15 / 8e12 ? 1 : 0;
			f16b[hb].ms += rms; f16b[hb].flops += fl; f16b[hb].bytes += by; f16b[hb].n++;
		}
		char* bar = strchr(name, '|'); /* aggregate per KERNEL symbol (behind '|') */
		const char* key = bar ? bar + 1 : name;
		for (k = 0; k < nagg; k++) if (strncmp(agg[k].name, key, 191) == 0) break;
		if (k == nagg) { if (nagg == 64) continue; snprintf(agg[k].name, 192, "%s", key); agg[k].ms = agg[k].flops = agg[k].bytes = 0; agg[k].n = 0; nagg++; }
		agg[k].ms += rms; agg[k].flops += fl; agg[k].bytes += by; agg[k].n++;
	}
	nnc_mi355x_profile_enable(0);
	if (recf) fclose(recf);
	/* replica probe (the process-per-GPU form's data-parallel check, bench.py compares it across the ranks): after the timed steps every rank must hold the
	 * same parameters -- sum of squares of the first and the last parameter tensor, read back in the parameter's own precision */
	double probe_sq[2] = { 0, 0 };
	{
		const int pc = ccv_cnnp_model_parameter_count(model);
		int which;
		for (which = 0; which < 2 && pc > 0; which++) {
			const ccv_cnnp_model_io_t pio = ccv_cnnp_model_parameters(model, -1, which ? pc - 1 : 0);
			ccv_nnc_tensor_param_t pp = ccv_cnnp_model_parameter_tensor_params(model, pio);
			const int pdt = pp.datatype;
			pp.type = CCV_TENSOR_CPU_MEMORY;
			ccv_nnc_tensor_t* const hp = ccv_nnc_tensor_new(0, pp, 0);
			ccv_cnnp_model_parameter_copy(model, pio, hp);
			const size_t cnt = ccv_nnc_tensor_count(pp);
			size_t jj;
			double b = 0;
			if (pdt == CCV_16F) {
				float* const f = (float*)malloc(sizeof(float) * cnt);
				ccv_half_precision_to_float((uint16_t*)hp->data.f16, f, cnt);
				for (jj = 0; jj < cnt; jj++) b += (double)f[jj] * f[jj];
				free(f);
			} else
				for (jj = 0; jj < cnt; jj++) b += (double)hp->data.f32[jj] * hp->data.f32[jj];
			probe_sq[which] = b;
			ccv_nnc_tensor_free(hp);
		}
	}
	{ /* (a diverged half-precision run has no number to print: JSON has no NaN) */
		char pb[2][40];
		int w_;
		for (w_ = 0; w_ < 2; w_++) { if (probe_sq[w_] == probe_sq[w_] && probe_sq[w_] - probe_sq[w_] == 0) snprintf(pb[w_], 40, "%.17g", probe_sq[w_]); else snprintf(pb[w_], 40, "null"); }
		printf("{\"replica_probe_sumsq\": [%s, %s], ", pb[0], pb[1]);
	}
	{
		long ov_c = 0, ov_b = 0;
		nnc_mi355x_comm_overlap_stats(&ov_c, &ov_b);
		printf("\"comm_overlap\": {\"collectives\": %ld, \"buckets\": %ld}, ", ov_c, ov_b);
	}
	printf("\"capture\": {\"on\": %s, \"graph_nodes\": %d, \"capture_ms\": %.2f, \"issued_per_command_ms_per_step\": %.4f, \"launch_host_ms_median\": %.4f}, ", step_graph ? "true" : "false", nnc_mi355x_graph_node_count(step_graph), t_capture, ms_issued, cap_host_ms[2]);
	printf("\"f16_contractions_by_bound\": {\"mfma\": {\"launches\": %d, \"ms\": %.4f, \"flops\": %.6g, \"bytes\": %.6g}, \"hbm\": {\"launches\": %d, \"ms\": %.4f, \"flops\": %.6g, \"bytes\": %.6g}}, ",
		f16b[0].n, f16b[0].ms, f16b[0].flops, f16b[0].bytes, f16b[1].n, f16b[1].ms, f16b[1].flops, f16b[1].bytes);
	printf("\"kernels\": [");
	for (i = 0; i < nagg; i++) printf("%s{\"name\": \"%s\", \"launches\": %d, \"ms\": %.4f, \"flops\": %.6g, \"bytes\": %.6g}", i ? ", " : "", agg[i].name, agg[i].n, agg[i].ms, agg[i].flops, agg[i].bytes);
	printf("], ");
	printf("\"device_out_sumsq\": [");
	for (i = 0; i < devices; i++) printf("%s%.17g", i ? ", " : "", dev_sumsq[i]);
	printf("], \"device_out_sum\": [");
	for (i = 0; i < devices; i++) printf("%s%.17g", i ? ", " : "", dev_sum[i]);
	printf("], ");
	printf("\"driver\": \"reference host (ccv_cnnp_model_fit: cnnp, autodiff, compile, scheduler)\", \"model\": \"%s\", \"dtype\": \"%s\", \"format\": \"NCHW\", \"batch\": %d, \"input_hw\": %d, "
		"\"devices\": %d, \"process_per_gpu\": {\"world\": %d, \"rank\": %d, \"rccl_ranks\": %d}, \"ms_per_step\": %.4f, \"images_per_s\": %.2f, \"host_enqueue\": {\"ms_per_step_median\": %.4f, \"ms_per_step_min\": %.4f, \"drained_step_ms\": %.4f, \"commands_per_step\": %ld, \"us_per_command\": %.3f}, \"first_step_ms\": %.1f, \"softmax_row0_sum\": %.6f, \"softmax_worst_row_sum_err\": %.3g, \"outputs_finite\": %s, \"memory_gib\": %.3f}\n",
		is_dawn ? "CIFAR-10 DawnNet (bin/nnc/cifar-10.c)" : mini ? "resnet-mini (2 bottlenecks)" : "ResNet-50 v1d", half ? "f16" : "f32", batch, hw, devices, ranks ? world : 0, rank, ranks ? nnc_mi355x_comm_count() : 0, ms, (double)batch * devices / (ms * 1e-3), enq_ms[2], enq_ms[0], enq_step_ms[2], enq_cmds, enq_cmds > 0 ? enq_ms[2] * 1e3 / enq_cmds : 0.0, t_first, row0, worst, finite ? "true" : "false",
		(double)ccv_cnnp_model_memory_size(model) / (1024.0 * 1024.0 * 1024.0));
	for (i = 0; i < devices; i++) { ccv_nnc_tensor_free(labels_d[i]); ccv_nnc_tensor_free(softmax_d[i]); ccv_nnc_tensor_free(grad_d[i]); }
	ccv_nnc_tensor_free(hout);
	ccv_nnc_tensor_free(sync_host);
	ccv_nnc_tensor_free(hx);
	ccv_nnc_tensor_free(hfit);
	for (i = 0; i < devices; i++) { ccv_nnc_tensor_free(xs[i]); ccv_nnc_tensor_free(fits[i]); ccv_nnc_tensor_free(outs[i]); }
	if (step_graph) nnc_mi355x_graph_free(step_graph); /* (before the tensors it names go) */
	ccv_cnnp_model_free(model);
	if (stream) ccv_nnc_stream_context_free(stream);
	return 0;
}
