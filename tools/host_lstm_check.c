/* The reference's OWN recurrent model -- ccv_cnnp_lstm (lib/nnc/ccv_cnnp_model_addons.c:3394-3480; what the IMDB classifier of test/int/nnc/imdb.tests.c:972-980
 * applies) -- compiled, evaluated and differentiated by the reference's unmodified host on this backend's LSTM rows: the host sizes the weights
 * (_ccv_cnnp_lstm_weight_dim) and the reserved space (its shape inference calls registry->aux, lib/nnc/cmd/rnn/ccv_nnc_lstm.c:64-71), initialises the weights with
 * RANDOM_UNIFORM, schedules LSTM_FORWARD and -- through its autodiff -- LSTM_BACKWARD.  The harness dumps x, the weights the host drew, y, dy and dx to a file;
 * tests/test_via_host.py replays oracle/lstm_numpy.py on them.  Test infrastructure (built by oracle/build_ref_host.sh next to the other harnesses), not product.
 *   host_lstm_check.{gpu,emu} T B I H layers bidirectional batch_first masked out.bin
 * HOST_LSTM_BENCH=<K> in the environment: after the checked pass, K more evaluate + backward passes are timed (wall clock between two blocking read-backs)
 * and a second JSON line reports them -- the recurrent half of a training step of the IMDB classifier at its own shape (bench.py --config imdb-lstm-bs64). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <nnc/ccv_nnc.h>
#include <nnc/ccv_nnc_easy.h>
#include <sys/time.h>

static unsigned lcg_state = 12345u;
static float lcg(void) { lcg_state = lcg_state * 1664525u + 1013904223u; return (float)(lcg_state >> 8) / 16777216.f - 0.5f; }

int main(int argc, char** argv)
{
	if (argc < 10) { fprintf(stderr, "usage: %s T B I H layers bidirectional batch_first masked out.bin\n", argv[0]); return 2; }
	const int T = atoi(argv[1]), B = atoi(argv[2]), I = atoi(argv[3]), H = atoi(argv[4]), L = atoi(argv[5]), bidir = atoi(argv[6]), batch_first = atoi(argv[7]), masked = atoi(argv[8]);
	const int D = bidir ? 2 : 1;
	ccv_nnc_init();
	ccv_cnnp_model_t* const model = ccv_cnnp_lstm(masked, H, 0, L, 1, batch_first, bidir, 0, 1, "lstm");
	const int d0 = batch_first ? B : T, d1 = batch_first ? T : B;
	ccv_nnc_tensor_param_t params[2] = { GPU_TENSOR_NHWC(000, 32F, d0, d1, I), CPU_TENSOR_NHWC(32S, B) };
	ccv_cnnp_model_compile(model, params, masked ? 2 : 1, CMD_NOOP(), CMD_NOOP());
	ccv_nnc_tensor_t* const x = ccv_nnc_tensor_new(0, CPU_TENSOR_NHWC(32F, d0, d1, I), 0);
	ccv_nnc_tensor_t* const dy = ccv_nnc_tensor_new(0, CPU_TENSOR_NHWC(32F, d0, d1, D * H), 0);
	ccv_nnc_tensor_t* const y = ccv_nnc_tensor_new(0, CPU_TENSOR_NHWC(32F, d0, d1, D * H), 0);
	ccv_nnc_tensor_t* const dx = ccv_nnc_tensor_new(0, CPU_TENSOR_NHWC(32F, d0, d1, I), 0);
	ccv_nnc_tensor_t* const lens = ccv_nnc_tensor_new(0, CPU_TENSOR_NHWC(32S, B), 0);
	int i;
	for (i = 0; i < T * B * I; i++) x->data.f32[i] = lcg();
	for (i = 0; i < T * B * D * H; i++) dy->data.f32[i] = lcg();
	for (i = 0; i < B; i++) lens->data.i32[i] = masked ? 1 + (i * 7 + 3) % T : T;
	ccv_nnc_tensor_t* const gx = ccv_nnc_tensor_new(0, GPU_TENSOR_NHWC(000, 32F, d0, d1, I), 0);
	ccv_nnc_tensor_t* const gdx = ccv_nnc_tensor_new(0, GPU_TENSOR_NHWC(000, 32F, d0, d1, I), 0);
	ccv_nnc_tensor_t* const gy = ccv_nnc_tensor_new(0, GPU_TENSOR_NHWC(000, 32F, d0, d1, D * H), 0);
	ccv_nnc_tensor_t* const gdy = ccv_nnc_tensor_new(0, GPU_TENSOR_NHWC(000, 32F, d0, d1, D * H), 0);
	ccv_nnc_cmd_exec(CMD_DATA_TRANSFER_FORWARD(), ccv_nnc_no_hint, 0, TENSOR_LIST(x, dy), TENSOR_LIST(gx, gdy), 0);
	ccv_nnc_tensor_t* ins[2] = { gx, lens };
	ccv_cnnp_model_evaluate(model, (ccv_cnnp_evaluate_param_t){ .requires_grad = 1 }, ins, masked ? 2 : 1, TENSOR_LIST(gy), 0, 0);
	ccv_nnc_tensor_t* outgrads[2] = { gdx, 0 };
	ccv_cnnp_model_backward(model, TENSOR_LIST(gdy), outgrads, masked ? 2 : 1, 0, 0);
	/* the weights the host drew: the one parameter of the model */
	int wrows = 0, l;
	for (l = 0; l < L; l++) wrows += D * (4 * (l == 0 ? I : D * H) + 4 * H + 8);
	ccv_nnc_tensor_t* const gw = ccv_nnc_tensor_new(0, GPU_TENSOR_NHWC(000, 32F, wrows, H), 0);
	ccv_nnc_tensor_t* const w = ccv_nnc_tensor_new(0, CPU_TENSOR_NHWC(32F, wrows, H), 0);
	ccv_cnnp_model_parameter_copy(model, ccv_cnnp_model_parameters(model, ALL_PARAMETERS, 0), gw);
	ccv_nnc_cmd_exec(CMD_DATA_TRANSFER_FORWARD(), ccv_nnc_no_hint, 0, TENSOR_LIST(gy, gdx, gw), TENSOR_LIST(y, dx, w), 0);
	FILE* const f = fopen(argv[9], "wb");
	if (!f) return 3;
	const int head[10] = { T, B, I, H, L, bidir, batch_first, masked, wrows, 0 };
	fwrite(head, sizeof(int), 10, f);
	fwrite(lens->data.i32, sizeof(int), B, f);
	fwrite(x->data.f32, sizeof(float), (size_t)T * B * I, f);
	fwrite(w->data.f32, sizeof(float), (size_t)wrows * H, f);
	fwrite(y->data.f32, sizeof(float), (size_t)T * B * D * H, f);
	fwrite(dy->data.f32, sizeof(float), (size_t)T * B * D * H, f);
	fwrite(dx->data.f32, sizeof(float), (size_t)T * B * I, f);
	fclose(f);
	printf("{\"lstm_via_host\": true, \"weight_rows\": %d}\n", wrows);
	if (getenv("HOST_LSTM_BENCH") && atoi(getenv("HOST_LSTM_BENCH")) > 0) {
		const int K = atoi(getenv("HOST_LSTM_BENCH"));
		ccv_nnc_tensor_t* const tiny = ccv_nnc_tensor_new(0, CPU_TENSOR_NHWC(32F, 1), 0);
		ccv_nnc_tensor_t* const gtiny = ccv_nnc_tensor_new(0, GPU_TENSOR_NHWC(000, 32F, 1), 0);
		int k;
		for (k = 0; k < 2; k++) { /* warm-up */
			ccv_cnnp_model_evaluate(model, (ccv_cnnp_evaluate_param_t){ .requires_grad = 1 }, ins, masked ? 2 : 1, TENSOR_LIST(gy), 0, 0);
			ccv_cnnp_model_backward(model, TENSOR_LIST(gdy), outgrads, masked ? 2 : 1, 0, 0);
		}
		ccv_nnc_cmd_exec(CMD_DATA_TRANSFER_FORWARD(), ccv_nnc_no_hint, 0, TENSOR_LIST(gtiny), TENSOR_LIST(tiny), 0); /* blocking: everything queued has run */
		struct timeval t0, t1;
		gettimeofday(&t0, 0);
		for (k = 0; k < K; k++) {
			ccv_cnnp_model_evaluate(model, (ccv_cnnp_evaluate_param_t){ .requires_grad = 1 }, ins, masked ? 2 : 1, TENSOR_LIST(gy), 0, 0);
			ccv_cnnp_model_backward(model, TENSOR_LIST(gdy), outgrads, masked ? 2 : 1, 0, 0);
		}
		ccv_nnc_cmd_exec(CMD_DATA_TRANSFER_FORWARD(), ccv_nnc_no_hint, 0, TENSOR_LIST(gtiny), TENSOR_LIST(tiny), 0);
		gettimeofday(&t1, 0);
		const double ms = ((t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_usec - t0.tv_usec) * 1e-3) / K;
		/* algorithmic flops of one evaluate + backward: 2 T B 4H (in + H) per pseudo-layer forward, three times that with the two gradients */
		double fl = 0;
		for (l = 0; l < L; l++) fl += 2.0 * T * B * 4 * H * ((l == 0 ? I : D * H) + H) * D;
		fl *= 3;
		printf("{\"lstm_bench\": true, \"steps\": %d, \"ms_per_step\": %.4f, \"sequences_per_s\": %.2f, \"timesteps_per_s\": %.1f, \"gflop_per_step\": %.4f, \"tflops\": %.4f}\n", K, ms, B / (ms * 1e-3), (double)B * T / (ms * 1e-3), fl / 1e9, fl / (ms * 1e-3) / 1e12);
	}
	ccv_cnnp_model_free(model);
	return 0;
}
