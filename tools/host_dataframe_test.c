/* TEST CLIENT (built by oracle/build_ref_host.sh against the unmodified reference host + this backend; run by tests/test_dataframe_binding.py):
 * the reference's dataframe iterated twice over the same synthetic decoded images with the same jitter seed --
 *   (1) the reference's own pipeline on the CPU: ccv_cnnp_dataframe_image_random_jitter, row by row (+ ccv_cnnp_dataframe_one_hot);
 *   (2) the GPU stage of integration/nnc_mi355x_dataframe.c: whole batches, decisions from the same SFMT stream, pixels on the device --
 * and the images / one-hot rows compared element by element.  Prints one JSON line.
 * host_dataframe_test <images> <batch> <mode> <32|16> bench: the GPU stage alone over <images> rows (64 distinct decoded images, repeated), no read-back: images/s.
 * usage: host_dataframe_test <images> <batch> <mode>   mode: "imagenet" (resize 256..480 -> 224 crop, aspect, colour jitter, flip) | "cifar" (32 x 32, offsets, flip) | "pad" (late crop with zero overhang) */
#include "ccv.h"
#include "nnc/ccv_nnc.h"
#include "nnc/ccv_nnc_easy.h"
#include "../integration/nnc_mi355x_dataframe.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <sys/time.h>

static double now_ms(void) { struct timeval tv; gettimeofday(&tv, 0); return tv.tv_sec * 1e3 + tv.tv_usec * 1e-3; }

static unsigned hash32(unsigned long long i, unsigned long long seed)
{
	unsigned long long h = (i + 1) * 0x9E3779B97F4A7C15ull ^ (seed + 1) * 0xD1B54A32D192ED03ull;
	h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
	return (unsigned)h;
}

int main(int argc, char** argv)
{
	const int count = argc > 1 ? atoi(argv[1]) : 12, batch = argc > 2 ? atoi(argv[2]) : 4;
	const char* const mode = argc > 3 ? argv[3] : "imagenet";
	const int half = argc > 4 && atoi(argv[4]) == 16;
	const int bench = argc > 5 && strcmp(argv[5], "bench") == 0;
	ccv_nnc_init();
	const int cifar = strcmp(mode, "cifar") == 0, pad = strcmp(mode, "pad") == 0;
	const int range = cifar ? 10 : 1000;
	ccv_array_t* const set = ccv_array_new(sizeof(ccv_categorized_t), count, 0);
	int i, j;
	for (i = 0; i < count; i++) { /* smooth-ish synthetic photographs of assorted sizes (what ccv_read would decode) */
		if (bench && i >= 64) { ccv_categorized_t cat = *(ccv_categorized_t*)ccv_array_get(set, i % 64); ccv_array_push(set, &cat); continue; }
		const int rows = cifar ? 32 : 180 + (int)(hash32(i, 1) % 260), cols = cifar ? 32 : 200 + (int)(hash32(i, 2) % 300);
		ccv_dense_matrix_t* const m = ccv_dense_matrix_new(rows, cols, CCV_8U | CCV_C3, 0, 0);
		int y, x, c;
		for (y = 0; y < rows; y++)
			for (x = 0; x < cols; x++)
				for (c = 0; c < 3; c++) {
					const double v = 128 + 70 * sin(0.05 * x + 0.3 * c + i) * cos(0.04 * y + 0.2 * i) + (double)(hash32((unsigned long long)(y * cols + x) * 3 + c, 77 + i) % 40) - 20;
					m->data.u8[y * m->step + x * 3 + c] = (unsigned char)(v < 0 ? 0 : v > 255 ? 255 : v);
				}
		ccv_categorized_t cat = ccv_categorized((int)(hash32(i, 3) % range), m, 0);
		ccv_array_push(set, &cat);
	}
	ccv_cnnp_random_jitter_t jitter;
	memset(&jitter, 0, sizeof(jitter));
	jitter.seed = 20240924;
	jitter.symmetric = 1;
	if (cifar) { jitter.resize.min = jitter.resize.max = 32; jitter.size.rows = jitter.size.cols = 32; jitter.offset.x = jitter.offset.y = 4; jitter.normalize.mean[0] = 125.3f; jitter.normalize.mean[1] = 122.9f; jitter.normalize.mean[2] = 113.9f; }
	else if (pad) { jitter.resize.min = 120; jitter.resize.max = 200; jitter.size.rows = jitter.size.cols = 224; jitter.normalize.mean[0] = 100; jitter.normalize.std[0] = jitter.normalize.std[1] = jitter.normalize.std[2] = 50; }
	else { /* bin/nnc/imagenet.c:361-388 */
		jitter.resize.min = 256; jitter.resize.max = 480; jitter.size.rows = jitter.size.cols = 224; jitter.aspect_ratio = 0.5f;
		jitter.brightness = 0.4f; jitter.contrast = 0.4f; jitter.saturation = 0.4f; jitter.lighting = 0.1f;
		jitter.normalize.mean[0] = 123.68f; jitter.normalize.mean[1] = 116.779f; jitter.normalize.mean[2] = 103.939f;
		jitter.normalize.std[0] = 58.393f; jitter.normalize.std[1] = 57.12f; jitter.normalize.std[2] = 57.375f;
	}
	const int rows = jitter.size.rows, cols = jitter.size.cols;
	const float eta = 0.1f, onval = 1 - eta + eta / range, offval = eta / range;
	if (bench) { /* throughput of the GPU stage from ONE host thread, raw images resident in host memory */
		ccv_cnnp_dataframe_t* const df = ccv_cnnp_dataframe_from_array_new(set);
		const int images = ccv_cnnp_dataframe_extract_value(df, 0, offsetof(ccv_categorized_t, matrix), 0);
		ccv_cnnp_dataframe_t* const bdf = nnc_mi355x_dataframe_jitter_batch_new(df, images, 0, offsetof(ccv_categorized_t, c), batch, jitter, range, onval, offval, half ? CCV_16F : CCV_32F, CCV_TENSOR_FORMAT_NCHW, 0, 3);
		ccv_nnc_stream_context_t* const stream = ccv_nnc_stream_context_new(CCV_STREAM_CONTEXT_GPU);
		int epoch, done = 0;
		double t0 = 0;
		for (epoch = 0; epoch < 3; epoch++) { /* epoch 0 warms up (ring allocation, first launches) */
			ccv_cnnp_dataframe_iter_t* const iter = ccv_cnnp_dataframe_iter_new(bdf, COLUMN_ID_LIST(0));
			void* data[1];
			if (epoch == 1) { ccv_nnc_stream_context_wait(stream); t0 = now_ms(); done = 0; }
			while (ccv_cnnp_dataframe_iter_next(iter, data, 1, stream) == 0) done += ((const nnc_mi355x_batch_t*)data[0])->count;
			ccv_cnnp_dataframe_iter_free(iter);
		}
		ccv_nnc_stream_context_wait(stream);
		const double ms = now_ms() - t0;
		printf("{\"mode\": \"%s\", \"bench\": true, \"images\": %d, \"batch\": %d, \"dtype\": \"%s\", \"ms\": %.1f, \"images_per_s\": %.1f}\n", mode, done, batch, half ? "f16" : "f32", ms, done / (ms * 1e-3));
		return 0;
	}
	/* (1) the reference's stages, CPU, one row at a time */
	float* const want = (float*)malloc(sizeof(float) * (size_t)count * rows * cols * 3);
	float* const want_hot = (float*)malloc(sizeof(float) * (size_t)count * range);
	double t_cpu;
	{
		ccv_cnnp_dataframe_t* const df = ccv_cnnp_dataframe_from_array_new(set);
		const int images = ccv_cnnp_dataframe_extract_value(df, 0, offsetof(ccv_categorized_t, matrix), 0);
		const int jit = ccv_cnnp_dataframe_image_random_jitter(df, images, CCV_32F, jitter, 0);
		const int hot = ccv_cnnp_dataframe_one_hot(df, 0, offsetof(ccv_categorized_t, c), range, onval, offval, CCV_32F, CCV_TENSOR_FORMAT_NCHW, 0);
		ccv_cnnp_dataframe_iter_t* const iter = ccv_cnnp_dataframe_iter_new(df, COLUMN_ID_LIST(jit, hot));
		void* data[2];
		const double t0 = now_ms();
		for (i = 0; i < count; i++) {
			if (ccv_cnnp_dataframe_iter_next(iter, data, 2, 0) != 0) { fprintf(stderr, "reference iterator ended early\n"); return 2; }
			const ccv_dense_matrix_t* const p = (const ccv_dense_matrix_t*)data[0];
			if (p->rows != rows || p->cols != cols || CCV_GET_DATA_TYPE(p->type) != CCV_32F) { fprintf(stderr, "unexpected patch %d x %d\n", p->rows, p->cols); return 2; }
			memcpy(want + (size_t)i * rows * cols * 3, p->data.f32, sizeof(float) * rows * cols * 3);
			memcpy(want_hot + (size_t)i * range, ((ccv_nnc_tensor_t*)data[1])->data.f32, sizeof(float) * range);
		}
		t_cpu = now_ms() - t0;
		ccv_cnnp_dataframe_iter_free(iter);
		ccv_cnnp_dataframe_free(df);
	}
	/* (2) the GPU stage, whole batches */
	double max_abs = 0, max_hot = 0, sum_abs = 0, t_gpu;
	size_t worst = 0;
	int batches = 0;
	{
		ccv_cnnp_dataframe_t* const df = ccv_cnnp_dataframe_from_array_new(set);
		const int images = ccv_cnnp_dataframe_extract_value(df, 0, offsetof(ccv_categorized_t, matrix), 0);
		ccv_cnnp_dataframe_t* const bdf = nnc_mi355x_dataframe_jitter_batch_new(df, images, 0, offsetof(ccv_categorized_t, c), batch, jitter, range, onval, offval, half ? CCV_16F : CCV_32F, CCV_TENSOR_FORMAT_NCHW, 0, 3);
		ccv_cnnp_dataframe_iter_t* const iter = ccv_cnnp_dataframe_iter_new(bdf, COLUMN_ID_LIST(0));
		ccv_nnc_tensor_t* const himg = ccv_nnc_tensor_new(0, CPU_TENSOR_NCHW(32F, batch, 3, rows, cols), 0);
		ccv_nnc_tensor_t* const hhot = ccv_nnc_tensor_new(0, CPU_TENSOR_NCHW(32F, batch, range), 0);
		ccv_nnc_tensor_t* const himg16 = half ? ccv_nnc_tensor_new(0, CPU_TENSOR_NCHW(16F, batch, 3, rows, cols), 0) : 0;
		ccv_nnc_tensor_t* const hhot16 = half ? ccv_nnc_tensor_new(0, CPU_TENSOR_NCHW(16F, batch, range), 0) : 0;
		void* data[1];
		int done = 0;
		const double t0 = now_ms();
		while (done < count && ccv_cnnp_dataframe_iter_next(iter, data, 1, 0) == 0) {
			const nnc_mi355x_batch_t* const b = (const nnc_mi355x_batch_t*)data[0];
			if (half) {
				ccv_nnc_cmd_exec(CMD_DATA_TRANSFER_FORWARD(), ccv_nnc_no_hint, 0, TENSOR_LIST(b->images, b->one_hot), TENSOR_LIST(himg16, hhot16), 0);
				ccv_nnc_cmd_exec(CMD_DATATYPE_CONVERSION_FORWARD(), ccv_nnc_no_hint, 0, TENSOR_LIST(himg16, hhot16), TENSOR_LIST(himg, hhot), 0);
			} else
				ccv_nnc_cmd_exec(CMD_DATA_TRANSFER_FORWARD(), ccv_nnc_no_hint, 0, TENSOR_LIST(b->images, b->one_hot), TENSOR_LIST(himg, hhot), 0);
			for (i = 0; i < b->count; i++) {
				const float* const w = want + (size_t)(done + i) * rows * cols * 3; /* the reference's patches are [y][x][c] */
				int y, x, c;
				for (c = 0; c < 3; c++) for (y = 0; y < rows; y++) for (x = 0; x < cols; x++) {
					const double d = fabs((double)himg->data.f32[(((size_t)i * 3 + c) * rows + y) * cols + x] - (double)w[(y * cols + x) * 3 + c]);
					sum_abs += d;
					if (d > max_abs) { max_abs = d; worst = (size_t)(done + i); }
				}
				for (j = 0; j < range; j++) { const double d = fabs((double)hhot->data.f32[(size_t)i * range + j] - (double)want_hot[(size_t)(done + i) * range + j]); if (d > max_hot) max_hot = d; }
			}
			done += b->count;
			batches++;
		}
		t_gpu = now_ms() - t0;
		if (done != count) { fprintf(stderr, "GPU stage yielded %d of %d images\n", done, count); return 3; }
		ccv_cnnp_dataframe_iter_free(iter);
		ccv_cnnp_dataframe_free(bdf);
		ccv_cnnp_dataframe_free(df);
	}
	double want_max = 0;
	for (i = 0; i < count * rows * cols * 3; i++) if (fabs(want[i]) > want_max) want_max = fabs(want[i]);
	printf("{\"mode\": \"%s\", \"images\": %d, \"batch\": %d, \"batches\": %d, \"dtype\": \"%s\", \"max_abs_diff\": %.6g, \"mean_abs_diff\": %.6g, \"worst_image\": %zu, \"reference_abs_max\": %.6g, \"one_hot_max_abs_diff\": %.6g, \"cpu_ms\": %.1f, \"gpu_stage_ms_incl_readback\": %.1f}\n",
		mode, count, batch, batches, half ? "f16" : "f32", max_abs, sum_abs / ((double)count * rows * cols * 3), worst, want_max, max_hot, t_cpu, t_gpu);
	return 0;
}
