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
 * ccv_cnnp_model_backward -> ccv_cnnp_model_apply_gradients, compiled with CMD_SGD_FORWARD(1, lr, 1 / batch, 0.01, 0.9, 0) and no loss. */
#include <ccv.h>
#include <nnc/ccv_nnc.h>
#include <nnc/ccv_nnc_easy.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <sys/time.h>

/* the backend's in-library launch records (include/nnc_mi355x.h) */
void nnc_mi355x_profile_enable(int on);
int nnc_mi355x_profile_count(void);
int nnc_mi355x_profile_get(int i, char* name, int name_len, double* flops, double* bytes, float* ms, int dims[5]);

static float hash_unit(const uint64_t i, const uint64_t seed)
{
	uint64_t h = (i + 1) * 0x9E3779B97F4A7C15ull ^ (seed + 1) * 0xD1B54A32D192ED03ull;
	h ^= h >> 30; h *= 0xBF58476D1CE4E5B9ull;
	h ^= h >> 27; h *= 0x94D049BB133111EBull;
	h ^= h >> 31;
	return (float)(h >> 40) * (1.0f / 16777216.0f);
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
	if (devices < 1 || devices > 8 || (devices > 1 && is_dawn)) { fprintf(stderr, "devices must be 1..8 (fit path only)\n"); return 2; }
	const int dt = half ? CCV_16F : CCV_32F;
	static const int blocks50[] = { 3, 4, 6, 3 }, widths50[] = { 64, 128, 256, 512 };
	static const int blocks_m[] = { 1, 1 }, widths_m[] = { 8, 16 };
	const int classes = (mini || is_dawn) ? 10 : 1000;
	ccv_nnc_init();
	ccv_cnnp_model_t* const model = is_dawn ? dawn() : mini ? resnet(blocks_m, widths_m, 2, 8, classes) : resnet(blocks50, widths50, 4, 64, classes);
	ccv_nnc_tensor_param_t input = GPU_TENSOR_NCHW(000, 32F, batch, 3, hw, hw);
	input.datatype = dt;
	const float lr = 0.01f, wd = 0.0001f;
	if (is_dawn) ccv_cnnp_model_compile(model, &input, 1, CMD_SGD_FORWARD(1, lr, 1. / batch, 0.01, 0.9, 0), CMD_NOOP());
	else ccv_cnnp_model_compile(model, &input, 1, CMD_SGD_FORWARD(1, lr, 1. / (batch * devices), wd, 0.9, 0), CMD_CATEGORICAL_CROSSENTROPY_FORWARD());
	if (devices > 1) ccv_cnnp_model_set_data_parallel(model, devices);
	/* synthetic batch: images ~ U(-1, 1) (normalised pixels), labels as the trainer's smoothed one-hot rows (eta = 0.1) */
	ccv_nnc_tensor_t* const hx = ccv_nnc_tensor_new(0, CPU_TENSOR_NCHW(32F, batch, 3, hw, hw), 0);
	ccv_nnc_tensor_t* const hfit = ccv_nnc_tensor_new(0, CPU_TENSOR_NCHW(32F, batch, classes), 0);
	size_t j;
	const size_t nx = (size_t)batch * 3 * hw * hw;
	const int shard0 = rot % devices;
	for (j = 0; j < nx; j++) hx->data.f32[j] = hash_unit(j, 2000 + 10 * shard0) * 2 - 1;
	const float eta = 0.1f;
	int i;
	for (i = 0; i < batch; i++) {
		const int c = (int)(hash_unit(i, 2001 + 10 * shard0) * classes);
		int k;
		for (k = 0; k < classes; k++) hfit->data.f32[(size_t)i * classes + k] = (k == c ? 1 - eta : 0) + eta / classes;
	}
	ccv_nnc_tensor_param_t xp = GPU_TENSOR_NCHW(000, 32F, batch, 3, hw, hw), fp = GPU_TENSOR_NCHW(000, 32F, batch, classes);
	xp.datatype = dt; fp.datatype = dt;
	ccv_nnc_tensor_t* const x = ccv_nnc_tensor_new(0, xp, 0);
	ccv_nnc_tensor_t* const fit = ccv_nnc_tensor_new(0, fp, 0);
	ccv_nnc_tensor_t* const out = ccv_nnc_tensor_new(0, fp, 0);
	if (half) {
		ccv_nnc_tensor_t* const hx16 = ccv_nnc_tensor_new(0, CPU_TENSOR_NCHW(16F, batch, 3, hw, hw), 0);
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
			for (j = 0; j < nx; j++) hx->data.f32[j] = hash_unit(j, 2000 + 10 * shard) * 2 - 1;
			for (i = 0; i < batch; i++) {
				const int c = (int)(hash_unit(i, 2001 + 10 * shard) * classes);
				int k;
				for (k = 0; k < classes; k++) hfit->data.f32[(size_t)i * classes + k] = (k == c ? 1 - eta : 0) + eta / classes;
			}
			if (half) {
				ccv_nnc_tensor_t* const hx16 = ccv_nnc_tensor_new(0, CPU_TENSOR_NCHW(16F, batch, 3, hw, hw), 0);
				ccv_nnc_tensor_t* const hfit16 = ccv_nnc_tensor_new(0, CPU_TENSOR_NCHW(16F, batch, classes), 0);
				ccv_nnc_cmd_exec(CMD_DATATYPE_CONVERSION_FORWARD(), ccv_nnc_no_hint, 0, TENSOR_LIST(hx, hfit), TENSOR_LIST(hx16, hfit16), 0);
				ccv_nnc_cmd_exec(CMD_DATA_TRANSFER_FORWARD(), ccv_nnc_no_hint, 0, TENSOR_LIST(hx16, hfit16), TENSOR_LIST(xs[d], fits[d]), 0);
				ccv_nnc_tensor_free(hx16);
				ccv_nnc_tensor_free(hfit16);
			} else
				ccv_nnc_cmd_exec(CMD_DATA_TRANSFER_FORWARD(), ccv_nnc_no_hint, 0, TENSOR_LIST(hx, hfit), TENSOR_LIST(xs[d], fits[d]), 0);
		}
	}
	ccv_nnc_stream_context_t* const stream = ccv_nnc_stream_context_new(CCV_STREAM_CONTEXT_GPU);
	/* reproducible parameter initialisation: the host seeds its generators from a thread-local ADDRESS otherwise (ccv_nnc_stream.c:262-281) */
	ccv_nnc_stream_context_set_seed(0, 20240923);
	ccv_nnc_stream_context_set_seed(stream, 20240924);
	/* dawn: the CIFAR trainer's step (cifar-10.c:259-273); labels are class indices in fp32, the softmax / gradient tensors have the outputs' type */
	ccv_nnc_tensor_t* const labels = ccv_nnc_tensor_new(0, GPU_TENSOR_NCHW(000, 32F, batch), 0);
	ccv_nnc_tensor_t* const softmax = ccv_nnc_tensor_new(0, fp, 0);
	ccv_nnc_tensor_t* const grad = ccv_nnc_tensor_new(0, fp, 0);
	{
		ccv_nnc_tensor_t* const hl = ccv_nnc_tensor_new(0, CPU_TENSOR_NCHW(32F, batch), 0);
		for (i = 0; i < batch; i++) hl->data.f32[i] = (float)(int)(hash_unit(i, 2001) * classes);
		ccv_nnc_cmd_exec(CMD_DATA_TRANSFER_FORWARD(), ccv_nnc_no_hint, 0, TENSOR_LIST(hl), TENSOR_LIST(labels), 0);
		ccv_nnc_tensor_free(hl);
	}
#define TRAIN_STEP() do { \
		if (is_dawn) { \
			ccv_cnnp_model_evaluate(model, (ccv_cnnp_evaluate_param_t){ .requires_grad = 1, .disable_outgrad = CCV_CNNP_DISABLE_OUTGRAD_ALL }, TENSOR_LIST(x), TENSOR_LIST(out), 0, stream); \
			ccv_nnc_cmd_exec(CMD_SOFTMAX_CROSSENTROPY_FORWARD(), ccv_nnc_no_hint, 0, TENSOR_LIST(out, labels), TENSOR_LIST(0, softmax), stream); \
			ccv_nnc_cmd_exec(CMD_SOFTMAX_CROSSENTROPY_BACKWARD(), ccv_nnc_no_hint, 0, TENSOR_LIST(0, 0, out, labels, 0, softmax), TENSOR_LIST(grad, 0), stream); \
			ccv_cnnp_model_backward(model, TENSOR_LIST(grad), TENSOR_LIST(), 0, stream); \
			ccv_cnnp_model_apply_gradients(model, stream); \
		} else \
			ccv_cnnp_model_fit(model, xs, devices, fits, devices, outs, devices, 0, stream); \
	} while (0)
	/* step 1: compiles the graph (autodiff, simplify, arena, schedule) and initialises the parameters */
	const double t_first0 = now_ms();
	TRAIN_STEP();
	ccv_nnc_stream_context_wait(stream);
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
	for (i = 1; i < warmup; i++) TRAIN_STEP();
	ccv_nnc_stream_context_wait(stream);
	const double t0 = now_ms();
	for (i = 0; i < steps; i++) TRAIN_STEP();
	ccv_nnc_stream_context_wait(stream);
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
	/* roofline leg: one more step with the backend's per-launch HIP-event records on (contractions and batch norm) */
	nnc_mi355x_profile_enable(1);
	TRAIN_STEP();
	ccv_nnc_stream_context_wait(stream);
	struct { char name[192]; double ms, flops, bytes; int n; } agg[64];
	int nagg = 0;
	const int nrec = nnc_mi355x_profile_count();
	for (i = 0; i < nrec; i++) {
		char name[256];
		double fl, by;
		float rms;
		int dims[5], k;
		nnc_mi355x_profile_get(i, name, 256, &fl, &by, &rms, dims);
		char* bar = strchr(name, '|'); /* aggregate per KERNEL symbol (behind '|') */
		const char* key = bar ? bar + 1 : name;
		for (k = 0; k < nagg; k++) if (strncmp(agg[k].name, key, 191) == 0) break;
		if (k == nagg) { if (nagg == 64) continue; snprintf(agg[k].name, 192, "%s", key); agg[k].ms = agg[k].flops = agg[k].bytes = 0; agg[k].n = 0; nagg++; }
		agg[k].ms += rms; agg[k].flops += fl; agg[k].bytes += by; agg[k].n++;
	}
	nnc_mi355x_profile_enable(0);
	printf("{\"kernels\": [");
	for (i = 0; i < nagg; i++) printf("%s{\"name\": \"%s\", \"launches\": %d, \"ms\": %.4f, \"flops\": %.6g, \"bytes\": %.6g}", i ? ", " : "", agg[i].name, agg[i].n, agg[i].ms, agg[i].flops, agg[i].bytes);
	printf("], ");
	printf("\"device_out_sumsq\": [");
	for (i = 0; i < devices; i++) printf("%s%.17g", i ? ", " : "", dev_sumsq[i]);
	printf("], \"device_out_sum\": [");
	for (i = 0; i < devices; i++) printf("%s%.17g", i ? ", " : "", dev_sum[i]);
	printf("], ");
	printf("\"driver\": \"reference host (ccv_cnnp_model_fit: cnnp, autodiff, compile, scheduler)\", \"model\": \"%s\", \"dtype\": \"%s\", \"format\": \"NCHW\", \"batch\": %d, \"input_hw\": %d, "
		"\"devices\": %d, \"ms_per_step\": %.4f, \"images_per_s\": %.2f, \"first_step_ms\": %.1f, \"softmax_row0_sum\": %.6f, \"softmax_worst_row_sum_err\": %.3g, \"outputs_finite\": %s, \"memory_gib\": %.3f}\n",
		is_dawn ? "CIFAR-10 DawnNet (bin/nnc/cifar-10.c)" : mini ? "resnet-mini (2 bottlenecks)" : "ResNet-50 v1d", half ? "f16" : "f32", batch, hw, devices, ms, (double)batch * devices / (ms * 1e-3), t_first, row0, worst, finite ? "true" : "false",
		(double)ccv_cnnp_model_memory_size(model) / (1024.0 * 1024.0 * 1024.0));
	ccv_nnc_tensor_free(labels);
	ccv_nnc_tensor_free(softmax);
	ccv_nnc_tensor_free(grad);
	ccv_nnc_tensor_free(hout);
	ccv_nnc_tensor_free(hx);
	ccv_nnc_tensor_free(hfit);
	for (i = 0; i < devices; i++) { ccv_nnc_tensor_free(xs[i]); ccv_nnc_tensor_free(fits[i]); ccv_nnc_tensor_free(outs[i]); }
	ccv_cnnp_model_free(model);
	ccv_nnc_stream_context_free(stream);
	return 0;
}
