/*
 * nnc_mi355x_pipeline.h -- the image side of libnnc_mi355x.so's C-ABI: the classic ccv_resample / ccv_filter loops in batch form, the data pipeline's random-jitter
 * and one-hot batch kernels, and the pinned staging ring that brings raw images to the device.  Part of include/nnc_mi355x.h (which includes it); kept in a
 * file of its own because it needs NOTHING of the nnc struct mirrors -- only the stream context as an opaque type -- so host-side glue can include it next to
 * the reference's own ccv.h / nnc/ccv_nnc.h (integration/nnc_mi355x_dataframe.c), where the mirrors of nnc_mi355x.h would collide with the originals.
 */
#ifndef NNC_MI355X_PIPELINE_H
#define NNC_MI355X_PIPELINE_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct ccv_nnc_stream_context_s ccv_nnc_stream_context_t; /* (the same opaque typedef the reference's headers carry) */

/* ------------------------------------------------ 5. classic image pre-process loops (batch form) -------------------- */
/* A batch of `count` same-sized images resident in HBM: image i starts at base + i * image_stride; rows are `step` bytes
 * apart; pixels are `channels` interleaved elements of `datatype` (CCV_8U or CCV_32F) -- ccv_dense_matrix_t's raster
 * (lib/nnc/ccv_nnc_tfb.h:118-153) without the header. */
typedef struct {
	int rows, cols, channels;
	int datatype;
	long step;         /* bytes between rows   */
	long image_stride; /* bytes between images */
} nnc_mi355x_image_batch_t;
/* ccv_resample (lib/ccv.h:1294, lib/ccv_resample.c:433-478) over a batch: `type` is CCV_INTER_AREA (0x01, down-scaling) or
 * CCV_INTER_CUBIC (0x04); rows_scale / cols_scale as in the reference (the output size is b's).  8u -> 8u area is
 * bit-exact with the reference; the float paths replay its accumulation order.  Returns CCV_NNC_EXEC_*. */
int nnc_mi355x_resample_batch(const void* a, const nnc_mi355x_image_batch_t a_desc, void* b, const nnc_mi355x_image_batch_t b_desc, const int count, const double rows_scale, const double cols_scale, const int type, ccv_nnc_stream_context_t* const stream_context);
/* ccv_filter (lib/ccv.h, lib/ccv_numeric.c:1036-1061) over a batch: correlation of every image with a small HOST-side 32F
 * kernel (kernel_rows x kernel_cols x kernel_channels, channels = 1 or the image's), same-size output.  8u -> 8u follows the
 * reference's direct fixed-point path bit for bit (replicated border, ccv_numeric.c:960-1034); the float path is the linear
 * correlation the reference's FFT path computes: zero border, centre tap (size - 1) / 2 (ccv_numeric.c:771-). */
int nnc_mi355x_filter_batch(const void* a, const nnc_mi355x_image_batch_t a_desc, const void* kernel_host, const int kernel_rows, const int kernel_cols, const int kernel_channels, void* d, const nnc_mi355x_image_batch_t d_desc, const int count, ccv_nnc_stream_context_t* const stream_context);

/* The pixel half of the data pipeline's random jitter (lib/nnc/ccv_cnnp_dataframe_addons.c:265-366, _ccv_cnnp_random_jitter) for a whole
 * batch: the HOST keeps the decisions -- per image, from the reference's own generator and integer arithmetic (:276-330): the source
 * slice, the size it is resampled to, the mirror flag, the crop window -- and hands them over; the DEVICE resamples (area when
 * shrinking, bicubic otherwise, :335-344), mirrors (:347), normalises (:351-354; before the late crop, whose overhang stays 0) and
 * writes the batch tensor the trainer consumes (NHWC / NCHW, CCV_32F / CCV_16F) in one kernel.  Source images: 8u, interleaved
 * channels (what ccv_read produces), anywhere in ONE device buffer (e.g. a pinned staging ring copied with one H2D per batch).
 * Colour jitter (_ccv_cnnp_image_manip, :213-253): up to four per-image operations in the order the reference's shuffle produced,
 * with the factors its generator drew -- brightness (ccv_scale), saturation (ccv_saturation), contrast (ccv_contrast: about the
 * per-channel MEAN of the image as it stands at that point, which the device computes over the whole resampled image) and lighting
 * (three PCA offsets, clamped to [0, 255], :187-198) -- applied per pixel between the resample and the normalisation, in double
 * like the reference's ccv_* functions.  3-channel images only. */
typedef struct {
	size_t offset;                 /* byte offset of the image in the source buffer */
	int rows, cols, step;          /* extent, row pitch in bytes */
	int slice_x, slice_y, slice_rows, slice_cols; /* the region that is resampled (the whole image, or the crop-first slice :316-326) */
	int resize_rows, resize_cols;  /* the size the slice is resampled to */
	int crop_x, crop_y;            /* origin of the output window in the resampled image (0, 0 when cropped first); may overhang: zeros */
	int flip;                      /* mirror in x */
	int color_ops;                 /* 0 .. 4 colour operations, applied in this order */
	struct { int kind; float v[3]; } color[4]; /* kind: NNC_MI355X_COLOR_*; v[0] = the factor (brightness / saturation / contrast), v = the three offsets (lighting) */
} nnc_mi355x_jitter_image_t;
enum { NNC_MI355X_COLOR_BRIGHTNESS = 1, NNC_MI355X_COLOR_SATURATION = 2, NNC_MI355X_COLOR_CONTRAST = 3, NNC_MI355X_COLOR_LIGHTING = 4 };
typedef struct {
	int out_rows, out_cols, channels; /* random_jitter.size, 3 */
	float mean[3], inv_std[3];        /* (v - mean) * inv_std; the reference stores 1 / std (:388-389) */
	int format, datatype;             /* CCV_TENSOR_FORMAT_NHWC | NCHW; CCV_32F | CCV_16F */
} nnc_mi355x_jitter_params_t;
int nnc_mi355x_jitter_batch(const void* src, const nnc_mi355x_jitter_image_t* images_host, const int count, const nnc_mi355x_jitter_params_t params, void* out, ccv_nnc_stream_context_t* const stream_context);
/* _ccv_cnnp_one_hot (:378-): out[i][k] = k == labels[i] ? onval : offval, `range` values per row, CCV_32F or CCV_16F. */
int nnc_mi355x_one_hot_batch(const int* labels_host, const int count, const int range, const float onval, const float offval, const int datatype, void* out, ccv_nnc_stream_context_t* const stream_context);

/* Pinned staging ring: the host side of the GPU data pipeline (replaces the per-batch pageable copies behind
 * ccv_cnnp_dataframe_copy_to_gpu, lib/nnc/ccv_cnnp_dataframe_addons.c:21-120).  `slots` pinned host buffers and as many device buffers of
 * `slot_bytes`, one copy stream of its own.  A loader thread fills slot s on the host (..._host), hands it over (..._submit: asynchronous
 * host-to-device copy; the copy first waits -- on the device -- for the consumer that last read the slot's device buffer); the training stream
 * takes it (..._acquire: the consumer stream waits for the copy, the host does not), runs its kernels on ..._device(s) (nnc_mi355x_jitter_batch),
 * and gives it back (..._release: marks the point on the consumer stream behind which the device buffer may be overwritten).  ..._host blocks
 * until the previous copy OUT of that pinned buffer has finished, so a slot is refilled while other slots' copies and kernels run.
 * Returns 0 / a null pointer on a bad slot or an allocation failure. */
void* nnc_mi355x_staging_ring_new(int device, int slots, size_t slot_bytes);
void* nnc_mi355x_staging_ring_host(void* ring, int slot);
void* nnc_mi355x_staging_ring_device(void* ring, int slot);
int   nnc_mi355x_staging_ring_submit(void* ring, int slot, size_t bytes);
int   nnc_mi355x_staging_ring_acquire(void* ring, int slot, ccv_nnc_stream_context_t* const consumer);
int   nnc_mi355x_staging_ring_release(void* ring, int slot, ccv_nnc_stream_context_t* const consumer);
void  nnc_mi355x_staging_ring_free(void* ring);

#ifdef __cplusplus
}
#endif
#endif
