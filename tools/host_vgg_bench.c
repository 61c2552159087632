/* VGG-D training step THROUGH THE REFERENCE HOST: a client of the reference's public nnc API (lib/nnc/ccv_nnc.h) that
 * builds the network as a symbolic graph -- the forward node sequence of test/int/nnc/symbolic.graph.vgg.d.tests.c:14-90 with
 * the layer table of vgg_d_params (bin/vgg_models.inc:361-838), random weights instead of the sqlite model, no PNG --, lets
 * ccv_nnc_symbolic_graph_minimize derive backward + one SGD command per parameter (as test/int/nnc/parallel.tests.c:49-57
 * does), compiles it (tensor arena, exec arena), autotunes every node (ccv_nnc_graph_autotune -> the backend's autotune
 * entry points), installs the static schedule (ccv_nnc_graph_set_default_static_schedule: the host's own multi-stream
 * scheduler) and runs it on the MI355X backend.  Everything above ccv_nnc_cmd_exec is the reference's unmodified code;
 * this file is the benchmark driver only.  Built by oracle/build_ref_host.sh against libccv_host_gpu.so (and against the CPU
 * emulator build for the small-size test of the CPU tier).
 *   host_vgg_bench.gpu <batch> <input hw> <steps> <warmup> [mini|full] [fwd]     -> one JSON line
 * "fwd": the forward graph alone (BASELINE config 2; the node sequence of test/int/nnc/symbolic.graph.vgg.d.tests.c:14-90 /
 * graph.vgg.d.tests.c:14-90 on a synthetic image instead of the PNG): no minimize, the loss tensor kept, autotune, static schedule.
 * Weights / images / labels come from the counter hash below, which ccv_amd/vgg.py reproduces (init="hash"): bench.py runs
 * both drivers on the same numbers and compares losses and updated parameters. */
#include <ccv.h>
#include <nnc/ccv_nnc.h>
#include <nnc/ccv_nnc_easy.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <sys/time.h>

static float hash_unit(const uint64_t i, const uint64_t seed)
{ /* splitmix64 finaliser over (index, stream) -> [0, 1) with 24 bits */
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

/* layer table: > 0 conv with that many output channels, 0 pool, < 0 fully connected with -n outputs */
static const int vgg_d[] = { 64, 64, 0, 128, 128, 0, 256, 256, 256, 0, 512, 512, 512, 0, 512, 512, 512, 0, -4096, -4096, -1000 };
static const int mini[] = { 8, 8, 0, 16, 16, 0, -32, -10 };

#define MAXP 64

int main(int argc, char** argv)
{
	const int batch = argc > 1 ? atoi(argv[1]) : 256, hw0 = argc > 2 ? atoi(argv[2]) : 225;
	const int steps = argc > 3 ? atoi(argv[3]) : 4, warmup = argc > 4 ? atoi(argv[4]) : 1;
	const int use_mini = argc > 5 && strcmp(argv[5], "mini") == 0;
	const int fwd_only = (argc > 5 && strcmp(argv[5], "fwd") == 0) || (argc > 6 && strcmp(argv[6], "fwd") == 0);
	const int* const layers = use_mini ? mini : vgg_d;
	const int nlayers = use_mini ? (int)(sizeof(mini) / sizeof(int)) : (int)(sizeof(vgg_d) / sizeof(int));
	ccv_nnc_init();
	ccv_nnc_symbolic_graph_t* const sg = ccv_nnc_symbolic_graph_new();
	ccv_nnc_tensor_symbol_t params[MAXP];
	int param_fan_in[MAXP], nparams = 0;
	int h = hw0, w = hw0, c = 3, first = 1, flat = 0, i;
	const ccv_nnc_tensor_symbol_t x = ccv_nnc_tensor_symbol_new(sg, GPU_TENSOR_NHWC(000, 32F, batch, h, w, c), "x");
	const ccv_nnc_tensor_symbol_t label = ccv_nnc_tensor_symbol_new(sg, GPU_TENSOR_NHWC(000, 32F, batch), "label");
	ccv_nnc_tensor_symbol_t cur = x;
	for (i = 0; i < nlayers; i++) {
		const int l = layers[i];
		if (l > 0) { /* conv 3x3 stride 1 (+ ReLU); the first one without padding (vgg_d_params' 225 -> 223) */
			const int border = first ? 0 : 1, oh = h + 2 * border - 2, ow = w + 2 * border - 2;
			const ccv_nnc_tensor_symbol_t wt = ccv_nnc_tensor_symbol_new(sg, GPU_TENSOR_NHWC(000, 32F, l, 3, 3, c), "w");
			const ccv_nnc_tensor_symbol_t bias = ccv_nnc_tensor_symbol_new(sg, GPU_TENSOR_NHWC(000, 32F, l), "bias");
			const ccv_nnc_tensor_symbol_t y = ccv_nnc_tensor_symbol_new(sg, GPU_TENSOR_NHWC(000, 32F, batch, oh, ow, l), "conv");
			const ccv_nnc_graph_exec_symbol_t e = ccv_nnc_graph_exec_symbol_new(sg, CMD_CONVOLUTION_FORWARD(1, l, 3, 3, c), TENSOR_SYMBOL_LIST(cur, wt, bias), TENSOR_SYMBOL_LIST(y), "conv");
			ccv_nnc_graph_exec_symbol_set_hint(sg, e, HINT((1, 1), (border, border)));
			const ccv_nnc_tensor_symbol_t r = ccv_nnc_tensor_symbol_new(sg, GPU_TENSOR_NHWC(000, 32F, batch, oh, ow, l), "relu");
			ccv_nnc_graph_exec_symbol_new(sg, CMD_RELU_FORWARD(), TENSOR_SYMBOL_LIST(y), TENSOR_SYMBOL_LIST(r), "relu");
			param_fan_in[nparams] = 9 * c; params[nparams++] = wt;
			param_fan_in[nparams] = 0; params[nparams++] = bias;
			cur = r; h = oh; w = ow; c = l; first = 0;
		} else if (l == 0) { /* max pool 3x3 stride 2, no padding */
			const int oh = (h - 3) / 2 + 1, ow = (w - 3) / 2 + 1;
			const ccv_nnc_tensor_symbol_t y = ccv_nnc_tensor_symbol_new(sg, GPU_TENSOR_NHWC(000, 32F, batch, oh, ow, c), "pool");
			const ccv_nnc_graph_exec_symbol_t e = ccv_nnc_graph_exec_symbol_new(sg, CMD_MAX_POOL_FORWARD(3, 3), TENSOR_SYMBOL_LIST(cur), TENSOR_SYMBOL_LIST(y), "pool");
			ccv_nnc_graph_exec_symbol_set_hint(sg, e, HINT((2, 2), (0, 0)));
			cur = y; h = oh; w = ow;
		} else { /* fully connected (+ ReLU except after the last one) */
			const int k = -l, fan_in = flat ? c : h * w * c;
			if (!flat) cur = ccv_nnc_tensor_symbol_alias_new(sg, cur, ccv_nnc_no_ofs, DIM_ALLOC(fan_in, 1), GPU_TENSOR_NHWC(000, 32F, batch, fan_in), "flat");
			const ccv_nnc_tensor_symbol_t wt = ccv_nnc_tensor_symbol_new(sg, GPU_TENSOR_NHWC(000, 32F, k, fan_in), "w");
			const ccv_nnc_tensor_symbol_t bias = ccv_nnc_tensor_symbol_new(sg, GPU_TENSOR_NHWC(000, 32F, k), "bias");
			const ccv_nnc_tensor_symbol_t y = ccv_nnc_tensor_symbol_new(sg, GPU_TENSOR_NHWC(000, 32F, batch, k), "fc");
			ccv_nnc_graph_exec_symbol_new(sg, CMD_GEMM_FORWARD(NO_TRANSPOSE, TRANSPOSE(0, 1)), TENSOR_SYMBOL_LIST(cur, wt, bias), TENSOR_SYMBOL_LIST(y), "fc");
			param_fan_in[nparams] = fan_in; params[nparams++] = wt;
			param_fan_in[nparams] = 0; params[nparams++] = bias;
			cur = y;
			if (i < nlayers - 1) {
				const ccv_nnc_tensor_symbol_t r = ccv_nnc_tensor_symbol_new(sg, GPU_TENSOR_NHWC(000, 32F, batch, k), "relu");
				ccv_nnc_graph_exec_symbol_new(sg, CMD_RELU_FORWARD(), TENSOR_SYMBOL_LIST(y), TENSOR_SYMBOL_LIST(r), "relu");
				cur = r;
			}
			flat = 1; c = k;
		}
	}
	const int classes = c;
	const ccv_nnc_tensor_symbol_t loss = ccv_nnc_tensor_symbol_new(sg, GPU_TENSOR_NHWC(000, 32F, batch), "loss");
	const ccv_nnc_tensor_symbol_t softmax = ccv_nnc_tensor_symbol_new(sg, GPU_TENSOR_NHWC(000, 32F, batch, classes), "softmax");
	ccv_nnc_graph_exec_symbol_new(sg, CMD_SOFTMAX_CROSSENTROPY_FORWARD(), TENSOR_SYMBOL_LIST(cur, label), TENSOR_SYMBOL_LIST(loss, softmax), "softmax crossentropy");
	ccv_nnc_graph_exec_symbol_autogen(sg, 0, 0, CCV_NNC_AUTOGEN_ALL_EXECS | CCV_NNC_AUTOGEN_SOURCES_AND_DESTINATIONS);
	/* backward + SGD: what ccv_cnnp_model_fit compiles (lib/nnc/ccv_cnnp_model.c), spelled out as parallel.tests.c:49-57 does */
	const ccv_nnc_cmd_t sgd = CMD_SGD_FORWARD(0, 0.001, 1. / batch, 0.0005, 0.9, 0.9);
	ccv_nnc_tensor_symbol_t gradients[MAXP], updated[MAXP];
	const int aux_size = ccv_nnc_minimizer_saved_aux_size(sgd);
	ccv_nnc_tensor_symbol_map_t* const saved_aux = (ccv_nnc_tensor_symbol_map_t*)malloc(sizeof(ccv_nnc_tensor_symbol_map_t) * aux_size * nparams);
	ccv_nnc_graph_exec_symbol_t update_execs[MAXP];
	if (!fwd_only) {
		ccv_nnc_symbolic_graph_minimize(sg, sgd, TENSOR_SYMBOL_LIST(loss), params, nparams, 0, 0, SYMBOLIC_GRAPH_SOURCES(sg), SYMBOLIC_GRAPH_DESTINATIONS(sg), gradients, updated, saved_aux, update_execs);
		const ccv_nnc_tensor_symbol_t dloss = ccv_nnc_tensor_symbol_for_backward(sg, loss);
		ccv_nnc_graph_exec_symbol_new(sg, CMD_SET_FORWARD(1), TENSOR_SYMBOL_LIST(), TENSOR_SYMBOL_LIST(dloss), "set 1");
		ccv_nnc_graph_exec_symbol_autogen(sg, 0, 0, CCV_NNC_AUTOGEN_ALL_EXECS | CCV_NNC_AUTOGEN_SOURCES_AND_DESTINATIONS);
	}
	ccv_nnc_graph_t* graph;
	ccv_nnc_tensor_arena_t* arena;
	ccv_nnc_graph_exec_arena_t* exec_arena;
	/* outputs kept alive: the updated parameters, the loss and every momentum source / destination */
	ccv_nnc_tensor_symbol_t keep[MAXP * 3 + 3];
	int nkeep = 0;
	if (!fwd_only) {
		for (i = 0; i < nparams; i++) keep[nkeep++] = updated[i];
		for (i = 0; i < aux_size * nparams; i++) keep[nkeep++] = saved_aux[i].destination;
	}
	keep[nkeep++] = loss;
	keep[nkeep++] = softmax;
	ccv_nnc_symbolic_graph_compile(sg, ccv_nnc_default_compile_params, 0, 0, keep, nkeep, SYMBOLIC_GRAPH_SOURCES(sg), SYMBOLIC_GRAPH_DESTINATIONS(sg), &graph, &arena, &exec_arena);
	/* data: parameters, momenta, images, labels */
	uint64_t stream_id = 0;
	for (i = 0; i < nparams; i++, stream_id++) {
		ccv_nnc_tensor_t* const dev = ccv_nnc_tensor_from_symbol(arena, params[i]);
		ccv_nnc_tensor_param_t info = dev->info;
		info.type = CCV_TENSOR_CPU_MEMORY;
		ccv_nnc_tensor_t* const host = ccv_nnc_tensor_new(0, info, 0);
		const size_t n = ccv_nnc_tensor_count(info);
		size_t j;
		if (param_fan_in[i]) {
			const float s = sqrtf(6.0f / param_fan_in[i]);
			for (j = 0; j < n; j++) host->data.f32[j] = (hash_unit(j, stream_id) - 0.5f) * 2 * s;
		} else
			for (j = 0; j < n; j++) host->data.f32[j] = hash_unit(j, stream_id) * 0.01f;
		ccv_nnc_cmd_exec(CMD_DATA_TRANSFER_FORWARD(), ccv_nnc_no_hint, 0, TENSOR_LIST(host), TENSOR_LIST(dev), 0);
		ccv_nnc_tensor_free(host);
	}
	for (i = 0; !fwd_only && i < aux_size * nparams; i++)
		ccv_nnc_cmd_exec(CMD_SET_FORWARD(0), ccv_nnc_no_hint, 0, TENSOR_LIST(), TENSOR_LIST(ccv_nnc_tensor_from_symbol(arena, saved_aux[i].source)), 0);
	{
		ccv_nnc_tensor_t* const host = ccv_nnc_tensor_new(0, CPU_TENSOR_NHWC(32F, batch, hw0, hw0, 3), 0);
		const size_t n = (size_t)batch * hw0 * hw0 * 3;
		size_t j;
		for (j = 0; j < n; j++) host->data.f32[j] = hash_unit(j, 1000);
		ccv_nnc_cmd_exec(CMD_DATA_TRANSFER_FORWARD(), ccv_nnc_no_hint, 0, TENSOR_LIST(host), TENSOR_LIST(ccv_nnc_tensor_from_symbol(arena, x)), 0);
		ccv_nnc_tensor_free(host);
		ccv_nnc_tensor_t* const hl = ccv_nnc_tensor_new(0, CPU_TENSOR_NHWC(32F, batch), 0);
		for (i = 0; i < batch; i++) hl->data.f32[i] = (float)(int)(hash_unit(i, 1001) * classes);
		ccv_nnc_cmd_exec(CMD_DATA_TRANSFER_FORWARD(), ccv_nnc_no_hint, 0, TENSOR_LIST(hl), TENSOR_LIST(ccv_nnc_tensor_from_symbol(arena, label)), 0);
		ccv_nnc_tensor_free(hl);
	}
	const double t_tune0 = now_ms();
	ccv_nnc_graph_autotune(graph, (size_t)64 << 30, 0, TRAVERSE_FULL);
	const double t_tune = now_ms() - t_tune0;
	ccv_nnc_graph_set_default_static_schedule(graph, CCV_STREAM_CONTEXT_GPU, 0);
	ccv_nnc_stream_context_t* const stream = ccv_nnc_graph_default_stream(graph);
	/* step 1 (on the initial parameters): what the parity check reads */
	ccv_nnc_graph_run_with_schedule(graph, 0, 0, 0, stream);
	ccv_nnc_stream_context_wait(stream);
	ccv_nnc_tensor_t* const hloss = ccv_nnc_tensor_new(0, CPU_TENSOR_NHWC(32F, batch), 0);
	ccv_nnc_cmd_exec(CMD_DATA_TRANSFER_FORWARD(), ccv_nnc_no_hint, 0, TENSOR_LIST(ccv_nnc_tensor_from_symbol(arena, loss)), TENSOR_LIST(hloss), 0);
	double psum[MAXP], psq[MAXP];
	for (i = 0; !fwd_only && i < nparams; i++) {
		ccv_nnc_tensor_t* const dev = ccv_nnc_tensor_from_symbol(arena, updated[i]);
		ccv_nnc_tensor_param_t info = dev->info;
		info.type = CCV_TENSOR_CPU_MEMORY;
		ccv_nnc_tensor_t* const host = ccv_nnc_tensor_new(0, info, 0);
		ccv_nnc_cmd_exec(CMD_DATA_TRANSFER_FORWARD(), ccv_nnc_no_hint, 0, TENSOR_LIST(dev), TENSOR_LIST(host), 0);
		const size_t n = ccv_nnc_tensor_count(info);
		size_t j;
		double s = 0, q = 0;
		for (j = 0; j < n; j++) { s += host->data.f32[j]; q += (double)host->data.f32[j] * host->data.f32[j]; }
		psum[i] = s; psq[i] = q;
		ccv_nnc_tensor_free(host);
	}
	for (i = 1; i < warmup; i++) ccv_nnc_graph_run_with_schedule(graph, 0, 0, 0, stream);
	ccv_nnc_stream_context_wait(stream);
	const double t0 = now_ms();
	for (i = 0; i < steps; i++) ccv_nnc_graph_run_with_schedule(graph, 0, 0, 0, stream);
	ccv_nnc_stream_context_wait(stream);
	const double ms = (now_ms() - t0) / (steps > 0 ? steps : 1);
	/* forward only: the softmax rows of the first images, for the parity check against the command driver / the CPU reference */
	double softmax_row_sum_err = 0;
	float top_prob = 0;
	if (fwd_only) {
		ccv_nnc_tensor_t* const hs = ccv_nnc_tensor_new(0, CPU_TENSOR_NHWC(32F, batch, classes), 0);
		ccv_nnc_cmd_exec(CMD_DATA_TRANSFER_FORWARD(), ccv_nnc_no_hint, 0, TENSOR_LIST(ccv_nnc_tensor_from_symbol(arena, softmax)), TENSOR_LIST(hs), 0);
		int b, k;
		for (b = 0; b < batch; b++) {
			double s = 0;
			for (k = 0; k < classes; k++) { s += hs->data.f32[b * classes + k]; if (b == 0 && hs->data.f32[k] > top_prob) top_prob = hs->data.f32[k]; }
			if (fabs(s - 1) > softmax_row_sum_err) softmax_row_sum_err = fabs(s - 1);
		}
		ccv_nnc_tensor_free(hs);
	}
	printf("{\"driver\": \"reference host (symbolic graph, %scompile, autotune, static schedule)\", \"forward_only\": %d, \"softmax_worst_row_sum_err\": %.3g, \"image0_top_prob\": %.9g, \"batch\": %d,", fwd_only ? "" : "minimize, ", fwd_only, softmax_row_sum_err, top_prob, batch);
	printf(" \"ms_per_step\": %.4f, \"images_per_s\": %.2f, \"autotune_ms\": %.1f, \"loss\": [", ms, batch / (ms * 1e-3), t_tune);
	for (i = 0; i < batch && i < 8; i++) printf("%s%.9g", i ? ", " : "", hloss->data.f32[i]);
	printf("], \"updated_param_sum\": [");
	for (i = 0; !fwd_only && i < nparams; i++) printf("%s%.12g", i ? ", " : "", psum[i]);
	printf("], \"updated_param_sumsq\": [");
	for (i = 0; !fwd_only && i < nparams; i++) printf("%s%.12g", i ? ", " : "", psq[i]);
	printf("]}\n");
	ccv_nnc_tensor_free(hloss);
	ccv_nnc_graph_free(graph);
	ccv_nnc_tensor_arena_free(arena);
	ccv_nnc_graph_exec_arena_free(exec_arena);
	ccv_nnc_symbolic_graph_free(sg);
	free(saved_aux);
	return 0;
}
