/* The reference-side binding of the GPU data pipeline: ONE dataframe stage that stands where the
 * trainers chain four -- ccv_cnnp_dataframe_image_random_jitter + ccv_cnnp_dataframe_one_hot + ccv_cnnp_dataframe_combine_new +
 * ccv_cnnp_dataframe_copy_to_gpu (bin/nnc/imagenet.c:389-406, test/int/nnc/cifar.tests.c:100-126).  It is host code: it is compiled INTO the reference
 * host (against its ccv.h / nnc/ccv_nnc.h and its SFMT generator), and calls the backend's C-ABI (include/nnc_mi355x.h: the pinned staging ring and the
 * jitter / one-hot batch kernels).  Nothing in libnnc_mi355x.so depends on it. */
#ifndef NNC_MI355X_DATAFRAME_H
#define NNC_MI355X_DATAFRAME_H
#include "nnc/ccv_nnc.h"

typedef struct {
	ccv_nnc_tensor_t* images; /* [batch][3][rows][cols] (NCHW) or [batch][rows][cols][3] (NHWC) on the device, CCV_32F or CCV_16F */
	ccv_nnc_tensor_t* one_hot; /* [batch][range] on the device, or 0 when no label column was given */
	int count;                 /* images in this batch (the last batch of an epoch may be short) */
} nnc_mi355x_batch_t;

/* A new dataframe whose ONLY column (index 0) yields nnc_mi355x_batch_t*: `batch_size` consecutive rows of `dataframe`, decoded 8-bit 3-channel images from
 * `image_column` (ccv_dense_matrix_t*, what ccv_cnnp_dataframe_read_image / extract_value produce) put through the reference's random jitter -- the
 * decisions drawn on the host from the reference's own SFMT stream in the reference's order (lib/nnc/ccv_cnnp_dataframe_addons.c:265-330, :216-253), the
 * pixels on the device -- and, when label_column >= 0, the int at `label_structof` of that column's row object as a one-hot row (:400-).  Raw images travel
 * through a pinned staging ring of `slots` buffers (their copy overlaps the previous batch's kernels).  With the same seed the images are the ones
 * ccv_cnnp_dataframe_image_random_jitter yields row by row (integration test: tools/host_dataframe_test.c). */
ccv_cnnp_dataframe_t* nnc_mi355x_dataframe_jitter_batch_new(ccv_cnnp_dataframe_t* const dataframe, const int image_column, const int label_column, const off_t label_structof,
	const int batch_size, const ccv_cnnp_random_jitter_t random_jitter, const int one_hot_range, const float onval, const float offval,
	const int datatype, const int format, const int device_id, const int slots);
#endif
