// CCV_NNC_GEMM_FORWARD / BACKWARD on gfx950 (fp32 MFMA contraction core).
// Oracle semantics: lib/nnc/cmd/blas/ccv_nnc_gemm_cpu_ref.c:110-184 (forward), :318-466 (backward);
// matrix view rules: lib/nnc/ccv_nnc_easy.h:421-444 (ccv_nnc_tensor_get_matrix_params).  Replaces the cuBLAS calls of
// lib/nnc/cmd/blas/gpu/ccv_nnc_gemm_gpu_cublas.cu:232-390, :616-795 -- the bias add is an epilogue of the contraction
// instead of the reference's rank-1 GEMM with a ones vector (:158-162), and dbias is a column reduction (:421-431).
#include "gemm_launch.h"

using namespace nnc;

namespace {

struct matp_t { int batch, rows, cols; long batch_inc, rows_inc, cols_inc; };

// ccv_nnc_tensor_get_matrix_params: the trailing two dims are the matrix, dim[nd-3] (if any) the batch.
static bool matrix_params(const ccv_nnc_tensor_t* t, const int transpose[2], matp_t* m)
{
	const int nd = tensor_nd(t->info.dim);
	if (nd < 1 || nd > 3) return false; // deeper broadcast batches are not on this path
	int st[CCV_NNC_MAX_DIM_ALLOC];
	tensor_strides(t, st);
	const int* d = t->info.dim;
	m->batch = nd < 3 ? 1 : d[nd - 3];
	m->batch_inc = nd < 3 ? 0 : st[nd - 3];
	int rows = nd == 1 ? 1 : d[nd - 2];
	long rows_inc = nd >= 2 ? st[nd - 2] : (long)st[0] * d[0];
	int cols = d[nd - 1];
	long cols_inc = st[nd - 1];
	if (transpose[0] != transpose[1]) {
		if (nd < 2) return false;
		const int lo = nd == 2 ? 0 : nd - 2, hi = nd == 2 ? 1 : nd - 1;
		if (!((transpose[0] == lo && transpose[1] == hi) || (transpose[1] == lo && transpose[0] == hi))) return false;
		int ti = rows; rows = cols; cols = ti;
		long tl = rows_inc; rows_inc = cols_inc; cols_inc = tl;
	}
	m->rows = rows; m->cols = cols; m->rows_inc = rows_inc; m->cols_inc = cols_inc;
	return true;
}

static const int no_transpose[2] = { 0, 0 };

static int _gemm_forw(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	if (input_size < 2 || output_size < 1 || !inputs[0] || !inputs[1] || !outputs[0]) return CCV_NNC_EXEC_INVALID;
	const ccv_nnc_tensor_t* a = inputs[0];
	const ccv_nnc_tensor_t* w = inputs[1];
	const ccv_nnc_tensor_t* bias = input_size > 2 ? inputs[2] : 0;
	ccv_nnc_tensor_t* b = outputs[0];
	if (CCV_GET_DATA_TYPE(a->info.datatype) != CCV_32F) return CCV_NNC_EXEC_INVALID;
	matp_t am, wm, bm;
	if (!matrix_params(a, cmd.info.blas.transpose_a, &am) || !matrix_params(w, cmd.info.blas.transpose_b, &wm) || !matrix_params(b, no_transpose, &bm)) return CCV_NNC_EXEC_INVALID;
	if ((am.batch > wm.batch ? am.batch : wm.batch) != bm.batch || am.rows != bm.rows || am.cols != wm.rows || wm.cols != bm.cols) return CCV_NNC_EXEC_INVALID;
	if ((am.batch != bm.batch && am.batch != 1) || (wm.batch != bm.batch && wm.batch != 1)) return CCV_NNC_EXEC_INVALID;
	if (am.batch == 1) am.batch_inc = 0;
	if (wm.batch == 1) wm.batch_inc = 0;
	const float* biasp = 0;
	long bias_z = 0;
	if (bias) {
		matp_t sm;
		if (!matrix_params(bias, no_transpose, &sm) || sm.cols != bm.cols || sm.cols_inc != 1 || sm.rows != 1) return CCV_NNC_EXEC_INVALID;
		if (sm.batch != 1 && sm.batch != bm.batch) return CCV_NNC_EXEC_INVALID;
		biasp = bias->data.f32;
		bias_z = sm.batch == 1 ? 0 : sm.batch_inc;
	}
	const MatOperand A = { a->data.f32, am.rows_inc, am.cols_inc, am.rows, am.cols };
	const MatOperand B = { w->data.f32, wm.cols_inc, wm.rows_inc, wm.cols, wm.rows }; // rows of B-as-loader are output columns
	const GemmOut out = { b->data.f32, bm.rows_inc, bm.cols_inc, biasp, 1.f, 0 };
	return gemm_strided("gemm_fwd", A, B, out, bm.batch, am.batch_inc, wm.batch_inc, bm.batch_inc, bias_z, flags, stream_context);
}

static int _gemm_back(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	// inputs: g, a, [w]; outputs: [h], [dw], [dbias]   (lib/nnc/cmd/blas/ccv_nnc_blas.c:23-45)
	if (input_size < 2 || output_size < 1 || !inputs[0]) return CCV_NNC_EXEC_INVALID;
	const ccv_nnc_tensor_t* g = inputs[0];
	const ccv_nnc_tensor_t* a = inputs[1];
	const ccv_nnc_tensor_t* w = input_size > 2 ? inputs[2] : 0;
	ccv_nnc_tensor_t* h = outputs[0];
	ccv_nnc_tensor_t* dw = output_size > 1 ? outputs[1] : 0;
	ccv_nnc_tensor_t* dbias = output_size > 2 ? outputs[2] : 0;
	if (CCV_GET_DATA_TYPE(g->info.datatype) != CCV_32F) return CCV_NNC_EXEC_INVALID;
	const int acc = (flags & CCV_NNC_ACCUMULATE_OUTPUT) ? 1 : 0;
	matp_t gm;
	if (!matrix_params(g, no_transpose, &gm)) return CCV_NNC_EXEC_INVALID;
	int ret;
	if (dbias) {
		matp_t sm;
		if (!matrix_params(dbias, no_transpose, &sm) || sm.cols != gm.cols || sm.cols_inc != 1 || sm.rows != 1 || gm.cols_inc != 1) return CCV_NNC_EXEC_INVALID;
		if (sm.batch != 1 && sm.batch != gm.batch) return CCV_NNC_EXEC_INVALID;
		for (int z = 0; z < gm.batch; z++) {
			float* dst = dbias->data.f32 + (sm.batch == 1 ? 0 : (long)z * sm.batch_inc);
			const int accz = acc || (sm.batch == 1 && z > 0);
			if ((ret = colsum_f32(g->data.f32 + (long)z * gm.batch_inc, gm.rows, gm.cols, gm.rows_inc, dst, accz, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
		}
	}
	if (dw) {
		if (!a) return CCV_NNC_EXEC_INVALID;
		matp_t am, dm;
		if (!matrix_params(a, cmd.info.blas.transpose_a, &am) || !matrix_params(dw, cmd.info.blas.transpose_b, &dm)) return CCV_NNC_EXEC_INVALID;
		if (am.rows != gm.rows || am.cols != dm.rows || dm.cols != gm.cols) return CCV_NNC_EXEC_INVALID;
		if ((am.batch != gm.batch && am.batch != 1) || (dm.batch != gm.batch && dm.batch != 1)) return CCV_NNC_EXEC_INVALID;
		if (am.batch == 1) am.batch_inc = 0;
		// dw(k, n) = sum_m a(m, k) * g(m, n)
		const MatOperand A = { a->data.f32, am.cols_inc, am.rows_inc, am.cols, am.rows };
		const MatOperand B = { g->data.f32, gm.cols_inc, gm.rows_inc, gm.cols, gm.rows };
		if (dm.batch == gm.batch) {
			const GemmOut out = { dw->data.f32, dm.rows_inc, dm.cols_inc, 0, 1.f, acc };
			if ((ret = gemm_strided("gemm_dw", A, B, out, gm.batch, am.batch_inc, gm.batch_inc, dm.batch_inc, 0, flags, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
		} else { // shared weight across the batch: accumulate every batch entry into the single dw
			for (int z = 0; z < gm.batch; z++) {
				MatOperand Az = A, Bz = B;
				Az.p += (long)z * am.batch_inc; Bz.p += (long)z * gm.batch_inc;
				const GemmOut out = { dw->data.f32, dm.rows_inc, dm.cols_inc, 0, 1.f, acc || z > 0 };
				if ((ret = gemm_strided("gemm_dw", Az, Bz, out, 1, 0, 0, 0, 0, flags, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
			}
		}
	}
	if (h) {
		if (!w) return CCV_NNC_EXEC_INVALID;
		matp_t hm, wm;
		if (!matrix_params(h, cmd.info.blas.transpose_a, &hm) || !matrix_params(w, cmd.info.blas.transpose_b, &wm)) return CCV_NNC_EXEC_INVALID;
		if (hm.cols != wm.rows || wm.cols != gm.cols || hm.rows != gm.rows) return CCV_NNC_EXEC_INVALID;
		if ((hm.batch != gm.batch && hm.batch != 1) || (wm.batch != gm.batch && wm.batch != 1)) return CCV_NNC_EXEC_INVALID;
		if (wm.batch == 1) wm.batch_inc = 0;
		// h(m, k) = sum_n g(m, n) * w(k, n)
		const MatOperand A = { g->data.f32, gm.rows_inc, gm.cols_inc, gm.rows, gm.cols };
		const MatOperand B = { w->data.f32, wm.rows_inc, wm.cols_inc, wm.rows, wm.cols };
		if (hm.batch == gm.batch) {
			const GemmOut out = { h->data.f32, hm.rows_inc, hm.cols_inc, 0, 1.f, acc };
			if ((ret = gemm_strided("gemm_dx", A, B, out, gm.batch, gm.batch_inc, wm.batch_inc, hm.batch_inc, 0, flags, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
		} else {
			for (int z = 0; z < gm.batch; z++) {
				MatOperand Az = A, Bz = B;
				Az.p += (long)z * gm.batch_inc; Bz.p += (long)z * wm.batch_inc;
				const GemmOut out = { h->data.f32, hm.rows_inc, hm.cols_inc, 0, 1.f, acc || z > 0 };
				if ((ret = gemm_strided("gemm_dx", Az, Bz, out, 1, 0, 0, 0, 0, flags, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
			}
		}
	}
	return CCV_NNC_EXEC_SUCCESS;
}

} // namespace

extern "C" void _register_command_CCV_NNC_GEMM_FORWARD_backend_CCV_NNC_BACKEND_GPU_CUBLAS(ccv_nnc_cmd_backend_registry_t* const registry)
{
	registry->tensor_formats = CCV_TENSOR_FORMAT_NCHW | CCV_TENSOR_FORMAT_NHWC;
	registry->tensor_datatypes = CCV_32F;
	registry->tensor_memory = CCV_TENSOR_GPU_MEMORY;
	registry->algorithms = 1;
	registry->exec = _gemm_forw;
}

extern "C" void _register_command_CCV_NNC_GEMM_BACKWARD_backend_CCV_NNC_BACKEND_GPU_CUBLAS(ccv_nnc_cmd_backend_registry_t* const registry)
{
	registry->tensor_formats = CCV_TENSOR_FORMAT_NCHW | CCV_TENSOR_FORMAT_NHWC;
	registry->tensor_datatypes = CCV_32F;
	registry->tensor_memory = CCV_TENSOR_GPU_MEMORY;
	registry->algorithms = 1;
	registry->exec = _gemm_back;
}
