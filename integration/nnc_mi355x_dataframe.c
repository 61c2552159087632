/* See nnc_mi355x_dataframe.h.  The per-image decisions below restate lib/nnc/ccv_cnnp_dataframe_addons.c:265-330 (_ccv_cnnp_random_jitter) and :216-253
 * (_ccv_cnnp_image_manip) draw for draw -- they must, the generator's stream is the contract -- and stop where the reference starts touching pixels:
 * from there on it is nnc_mi355x_jitter_batch (ccv_amd/csrc/img_preproc.cpp). */
#include "nnc_mi355x_dataframe.h"
#include "nnc/ccv_nnc_easy.h"
#include "3rdparty/sfmt/SFMT.h"
#include "../include/nnc_mi355x_pipeline.h"
#include <math.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

/* A failed stage must not hand the trainer an unwritten batch: these checks stay in NDEBUG builds and stop the process with the reason,
 * as the reference's own *_ENFORCE macros do for device errors. */
#define NNC_DF_ENFORCE(cond, what) do { if (!(cond)) { fprintf(stderr, "[nnc_mi355x dataframe] %s (%s:%d)\n", what, __FILE__, __LINE__); abort(); } } while (0)

typedef struct {
	sfmt_t sfmt;
	ccv_cnnp_random_jitter_t jitter; /* normalize.std holds 1 / std, as the reference stores it (:386-387) */
	int batch_size, label_structof_valid, range, datatype, format, device_id, slots, next_slot;
	off_t label_structof;
	float onval, offval;
	void* ring;
	size_t slot_bytes;
} jitter_batch_context_t;

static float random_logexp(sfmt_t* const sfmt, const float jitter)
{ /* :205-214 */
	const double log_jitter_limit = log(1 + jitter);
	const double log_random_jitter = sfmt_genrand_real1(sfmt) * 2 * log_jitter_limit - log_jitter_limit;
	return (float)exp(log_random_jitter);
}

/* one image's decisions (:276-330), then its colour operations in the shuffled order (:216-253) */
static void decide(const ccv_cnnp_random_jitter_t random_jitter, sfmt_t* const sfmt, const int rows, const int cols, nnc_mi355x_jitter_image_t* const im)
{
	const int resize = ccv_clamp((int)(sfmt_genrand_real1(sfmt) * (random_jitter.resize.max - random_jitter.resize.min) + 0.5) + random_jitter.resize.min, random_jitter.resize.min, random_jitter.resize.max);
	int resize_rows = ccv_max(resize, (int)(rows * (float)resize / cols + 0.5));
	int resize_cols = ccv_max(resize, (int)(cols * (float)resize / rows + 0.5));
	if (random_jitter.aspect_ratio > 0) {
		const float aspect_ratio = sqrtf(random_logexp(sfmt, random_jitter.aspect_ratio));
		resize_rows = (int)(resize_rows * aspect_ratio + 0.5);
		resize_cols = (int)(resize_cols / aspect_ratio + 0.5);
	}
	if (random_jitter.resize.roundup > 0) {
		const int roundup = random_jitter.resize.roundup;
		const int roundup_2 = roundup / 2;
		resize_rows = (resize_rows + roundup_2) / roundup * roundup;
		resize_cols = (resize_cols + roundup_2) / roundup * roundup;
	}
	const int need_crop = (random_jitter.size.cols > 0 && random_jitter.size.rows > 0 &&
		((resize_cols != random_jitter.size.cols || resize_rows != random_jitter.size.rows) || (random_jitter.offset.x != 0 || random_jitter.offset.y != 0)));
	int crop_x = 0, crop_y = 0;
	im->slice_x = im->slice_y = 0; im->slice_rows = rows; im->slice_cols = cols;
	if (need_crop) {
		crop_x = random_jitter.center_crop ? (resize_cols - random_jitter.size.cols + 1) / 2 : (int)(sfmt_genrand_real1(sfmt) * (resize_cols - random_jitter.size.cols + 1));
		crop_x = ccv_clamp(crop_x, ccv_min(0, resize_cols - random_jitter.size.cols), ccv_max(0, resize_cols - random_jitter.size.cols));
		crop_y = random_jitter.center_crop ? (resize_rows - random_jitter.size.rows + 1) / 2 : (int)(sfmt_genrand_real1(sfmt) * (resize_rows - random_jitter.size.rows + 1));
		crop_y = ccv_clamp(crop_y, ccv_min(0, resize_rows - random_jitter.size.rows), ccv_max(0, resize_rows - random_jitter.size.rows));
		if (random_jitter.offset.x != 0) crop_x += sfmt_genrand_real1(sfmt) * random_jitter.offset.x * 2 - random_jitter.offset.x;
		if (random_jitter.offset.y != 0) crop_y += sfmt_genrand_real1(sfmt) * random_jitter.offset.y * 2 - random_jitter.offset.y;
		if (resize_cols >= random_jitter.size.cols && resize_rows >= random_jitter.size.rows) { /* crop first, then scale */
			const float scale_x = (float)cols / resize_cols;
			const float scale_y = (float)rows / resize_rows;
			const int slice_cols = (int)(random_jitter.size.cols * scale_x + 0.5);
			const int slice_rows = (int)(random_jitter.size.rows * scale_y + 0.5);
			im->slice_x = ccv_clamp((int)(crop_x * scale_x + 0.5), 0, cols - slice_cols);
			im->slice_y = ccv_clamp((int)(crop_y * scale_y + 0.5), 0, rows - slice_rows);
			im->slice_rows = slice_rows; im->slice_cols = slice_cols;
			resize_cols = random_jitter.size.cols;
			resize_rows = random_jitter.size.rows;
			crop_x = crop_y = 0;
		}
	}
	im->resize_rows = resize_rows; im->resize_cols = resize_cols;
	im->crop_x = crop_x; im->crop_y = crop_y;
	im->flip = (random_jitter.symmetric && (sfmt_genrand_uint32(sfmt) & 1) == 0) ? 1 : 0;
	int idx[4] = { 0, 1, 2, 3 };
	sfmt_genrand_shuffle(sfmt, idx, 4, sizeof(int));
	int i, n = 0;
	for (i = 0; i < 4; i++)
		switch (idx[i]) {
			case 0:
				if (random_jitter.brightness == 0) break;
				im->color[n].kind = NNC_MI355X_COLOR_BRIGHTNESS; im->color[n].v[0] = random_logexp(sfmt, random_jitter.brightness); n++;
				break;
			case 1:
				if (random_jitter.saturation == 0) break;
				im->color[n].kind = NNC_MI355X_COLOR_SATURATION; im->color[n].v[0] = random_logexp(sfmt, random_jitter.saturation); n++;
				break;
			case 2:
				if (random_jitter.contrast == 0) break;
				im->color[n].kind = NNC_MI355X_COLOR_CONTRAST; im->color[n].v[0] = random_logexp(sfmt, random_jitter.contrast); n++;
				break;
			case 3: {
				if (random_jitter.lighting == 0) break;
				/* (the reference passes the three draws as ARGUMENTS of one call, :249: their evaluation order is the compiler's; the reference
				 * build evaluates them left to right -- the integration test pins it) */
				const float alpha_r = sfmt_genrand_real1(sfmt) * random_jitter.lighting;
				const float alpha_g = sfmt_genrand_real1(sfmt) * random_jitter.lighting;
				const float alpha_b = sfmt_genrand_real1(sfmt) * random_jitter.lighting;
				im->color[n].kind = NNC_MI355X_COLOR_LIGHTING; /* :187-198 */
				im->color[n].v[0] = alpha_r * (55.46 * -0.5675) + alpha_g * (4.794 * 0.7192) + alpha_b * (1.148 * 0.4009);
				im->color[n].v[1] = alpha_r * (55.46 * -0.5808) + alpha_g * (4.794 * -0.0045) + alpha_b * (1.148 * -0.8140);
				im->color[n].v[2] = alpha_r * (55.46 * -0.5836) + alpha_g * (4.794 * -0.6948) + alpha_b * (1.148 * 0.4203);
				n++;
				break;
			}
		}
	im->color_ops = n;
}

static void batch_deinit(void* const data, void* const context)
{
	nnc_mi355x_batch_t* const b = (nnc_mi355x_batch_t*)data;
	if (!b) return;
	if (b->images) ccv_nnc_tensor_free(b->images);
	if (b->one_hot) ccv_nnc_tensor_free(b->one_hot);
	free(b);
}

static void context_deinit(void* const context)
{
	jitter_batch_context_t* const ctx = (jitter_batch_context_t*)context;
	if (ctx->ring) nnc_mi355x_staging_ring_free(ctx->ring);
	free(ctx);
}

/* input_data[i]: the tuple (image, label row object) of row i */
static void jitter_batch_sample(void* const* const input_data, const int batch_size, void** const output_data, void* const context, ccv_nnc_stream_context_t* const stream_context)
{
	jitter_batch_context_t* const ctx = (jitter_batch_context_t*)context;
	nnc_mi355x_batch_t* b = (nnc_mi355x_batch_t*)*output_data;
	const int rows = ctx->jitter.size.rows, cols = ctx->jitter.size.cols;
	if (!b) { /* (a recycled batch object keeps its device tensors: the iterator hands back one the consumer is done with) */
		b = (nnc_mi355x_batch_t*)calloc(1, sizeof(nnc_mi355x_batch_t));
		ccv_nnc_tensor_param_t ip = ctx->format == CCV_TENSOR_FORMAT_NCHW ? GPU_TENSOR_NCHW(000, 32F, ctx->batch_size, 3, rows, cols) : GPU_TENSOR_NHWC(000, 32F, ctx->batch_size, rows, cols, 3);
		ip.datatype = ctx->datatype;
		CCV_TENSOR_SET_DEVICE_ID(ip.type, ctx->device_id);
		b->images = ccv_nnc_tensor_new(0, ip, 0);
		if (ctx->label_structof_valid) {
			ccv_nnc_tensor_param_t op = GPU_TENSOR_NCHW(000, 32F, ctx->batch_size, ctx->range);
			op.datatype = ctx->datatype; op.format = ctx->format;
			CCV_TENSOR_SET_DEVICE_ID(op.type, ctx->device_id);
			b->one_hot = ccv_nnc_tensor_new(0, op, 0);
		}
		*output_data = b;
	}
	b->count = batch_size;
	nnc_mi355x_jitter_image_t* const items = (nnc_mi355x_jitter_image_t*)calloc(batch_size, sizeof(nnc_mi355x_jitter_image_t));
	int* const labels = (int*)calloc(batch_size, sizeof(int));
	/* per-image generators seeded from the column's generator in row order, exactly as the reference's map does for the rows it is handed (:268-270) */
	sfmt_t* const sfmt = (sfmt_t*)malloc(sizeof(sfmt_t) * batch_size);
	int i;
	for (i = 0; i < batch_size; i++) sfmt_init_gen_rand(&sfmt[i], sfmt_genrand_uint32(&ctx->sfmt));
	size_t bytes = 0;
	for (i = 0; i < batch_size; i++) {
		void* const* const tuple = (void* const*)input_data[i];
		const ccv_dense_matrix_t* const image = (const ccv_dense_matrix_t*)tuple[0];
		NNC_DF_ENFORCE(CCV_GET_DATA_TYPE(image->type) == CCV_8U && CCV_GET_CHANNEL(image->type) == CCV_C3, "the image column must hold 8-bit, 3-channel matrices");
		items[i].offset = bytes; items[i].rows = image->rows; items[i].cols = image->cols; items[i].step = image->step;
		bytes += ((size_t)image->step * image->rows + 15) & ~(size_t)15;
		decide(ctx->jitter, &sfmt[i], image->rows, image->cols, &items[i]);
		if (ctx->label_structof_valid) labels[i] = *(const int*)((const char*)tuple[1] + ctx->label_structof);
	}
	free(sfmt);
	/* the raw images: one pinned slot, one asynchronous copy, handed to this stream on the device (include/nnc_mi355x.h: staging ring) */
	if (!ctx->ring || bytes > ctx->slot_bytes) {
		if (ctx->ring) nnc_mi355x_staging_ring_free(ctx->ring); /* (waits for its copies) */
		ctx->slot_bytes = bytes + bytes / 4 + 4096;
		ctx->ring = nnc_mi355x_staging_ring_new(ctx->device_id, ctx->slots, ctx->slot_bytes);
		ctx->next_slot = 0;
		NNC_DF_ENFORCE(ctx->ring, "the pinned staging ring could not be allocated");
	}
	const int slot = ctx->next_slot;
	ctx->next_slot = (slot + 1) % ctx->slots;
	unsigned char* const host = (unsigned char*)nnc_mi355x_staging_ring_host(ctx->ring, slot);
	for (i = 0; i < batch_size; i++) {
		const ccv_dense_matrix_t* const image = (const ccv_dense_matrix_t*)((void* const*)input_data[i])[0];
		memcpy(host + items[i].offset, image->data.u8, (size_t)image->step * image->rows);
	}
	int ok = nnc_mi355x_staging_ring_submit(ctx->ring, slot, bytes);
	ok = ok && nnc_mi355x_staging_ring_acquire(ctx->ring, slot, stream_context);
	NNC_DF_ENFORCE(ok, "the staging ring refused the slot (submit / acquire out of order)");
	nnc_mi355x_jitter_params_t params;
	memset(&params, 0, sizeof(params));
	params.out_rows = rows; params.out_cols = cols; params.channels = 3;
	for (i = 0; i < 3; i++) { params.mean[i] = ctx->jitter.normalize.mean[i]; params.inv_std[i] = ctx->jitter.normalize.std[i]; }
	params.format = ctx->format; params.datatype = ctx->datatype;
	const int r = nnc_mi355x_jitter_batch(nnc_mi355x_staging_ring_device(ctx->ring, slot), items, batch_size, params, b->images->data.u8, stream_context);
	NNC_DF_ENFORCE(r == CCV_NNC_EXEC_SUCCESS, "nnc_mi355x_jitter_batch failed: the batch tensor was not written");
	ok = nnc_mi355x_staging_ring_release(ctx->ring, slot, stream_context);
	NNC_DF_ENFORCE(ok, "the staging ring refused the release");
	if (b->one_hot) {
		const int r2 = nnc_mi355x_one_hot_batch(labels, batch_size, ctx->range, ctx->onval, ctx->offval, ctx->datatype, b->one_hot->data.u8, stream_context);
		NNC_DF_ENFORCE(r2 == CCV_NNC_EXEC_SUCCESS, "nnc_mi355x_one_hot_batch failed: the label tensor was not written");
	}
	free(items);
	free(labels);
}

ccv_cnnp_dataframe_t* nnc_mi355x_dataframe_jitter_batch_new(ccv_cnnp_dataframe_t* const dataframe, const int image_column, const int label_column, const off_t label_structof,
	const int batch_size, const ccv_cnnp_random_jitter_t random_jitter, const int one_hot_range, const float onval, const float offval,
	const int datatype, const int format, const int device_id, const int slots)
{
	NNC_DF_ENFORCE(random_jitter.resize.min > 0 && random_jitter.resize.max >= random_jitter.resize.min, "random_jitter.resize: 0 < min <= max");
	NNC_DF_ENFORCE(random_jitter.size.rows > 0 && random_jitter.size.cols > 0, "random_jitter.size: a batch tensor has ONE image size"); /* a batch tensor has ONE image size */
	jitter_batch_context_t* const ctx = (jitter_batch_context_t*)calloc(1, sizeof(jitter_batch_context_t));
	if (random_jitter.seed) sfmt_init_gen_rand(&ctx->sfmt, (uint32_t)random_jitter.seed);
	else sfmt_init_gen_rand(&ctx->sfmt, ccv_nnc_stream_context_genrand_uint32(0));
	ctx->jitter = random_jitter;
	int i;
	for (i = 0; i < 3; i++) ctx->jitter.normalize.std[i] = ctx->jitter.normalize.std[i] ? 1. / ctx->jitter.normalize.std[i] : 1; /* :386-387 */
	ctx->batch_size = batch_size; ctx->label_structof_valid = label_column >= 0; ctx->label_structof = label_structof;
	ctx->range = one_hot_range; ctx->onval = onval; ctx->offval = offval;
	ctx->datatype = datatype; ctx->format = format; ctx->device_id = device_id; ctx->slots = slots > 1 ? slots : 2;
	const int tuple_idx = ccv_cnnp_dataframe_make_tuple(dataframe, COLUMN_ID_LIST(image_column, label_column >= 0 ? label_column : image_column), 0);
	return ccv_cnnp_dataframe_sample_new(dataframe, jitter_batch_sample, batch_deinit, tuple_idx, batch_size, ctx, context_deinit);
}
