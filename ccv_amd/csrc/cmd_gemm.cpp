// CCV_NNC_GEMM_FORWARD / BACKWARD on gfx950 (fp32 MFMA contraction core; CCV_16F tensors on the half-precision core, mfma_gemm_f16.h).
// Oracle semantics: lib/nnc/cmd/blas/ccv_nnc_gemm_cpu_ref.c:110-184 (forward), :318-466 (backward);
// matrix view rules: lib/nnc/ccv_nnc_easy.h:421-444 (ccv_nnc_tensor_get_matrix_params).  Replaces the cuBLAS calls of
// lib/nnc/cmd/blas/gpu/ccv_nnc_gemm_gpu_cublas.cu:232-390, :616-795 -- the bias add is an epilogue of the contraction
// instead of the reference's rank-1 GEMM with a ones vector (:158-162), and dbias is a column reduction (:421-431).
#include "gemm_launch.h"

using namespace nnc;

namespace {

template <class T>
struct matp_t {
	int batch, rows, cols;          // inner batch = dim[nd-3]
	long batch_inc, rows_inc, cols_inc;
	int outer_nd;                   // leading dims in front of the inner batch (generalized batched GEMM, gemm_cpu_ref.c _ccv_nnc_gbmm)
	int outer_dim[CCV_NNC_MAX_DIM_ALLOC];
	long outer_inc[CCV_NNC_MAX_DIM_ALLOC];
	T* p;
};

// ccv_nnc_tensor_get_matrix_params (lib/nnc/ccv_nnc_easy.h:421-444): the trailing two dims are the matrix, dim[nd-3]
// (if any) the batch, anything in front of that an outer batch that is walked (and broadcast when 1) by the caller.
template <class T>
static bool matrix_params(const ccv_nnc_tensor_t* t, const int transpose[2], matp_t<T>* m)
{
	const int nd = tensor_nd(t->info.dim);
	if (nd < 1) return false;
	int st[CCV_NNC_MAX_DIM_ALLOC];
	tensor_strides(t, st);
	const int* d = t->info.dim;
	m->p = (T*)t->data.u8;
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
	m->outer_nd = nd > 3 ? nd - 3 : 0;
	for (int i = 0; i < m->outer_nd; i++) { m->outer_dim[i] = d[i]; m->outer_inc[i] = st[i]; }
	return true;
}

static const int no_transpose[2] = { 0, 0 };

// Walks the outer batch index space of `ref` (the operand that carries every outer dim at full extent) and hands the
// callback each operand's element offset; operands with fewer outer dims are right-aligned, extent-1 dims broadcast.
struct outer_walk_t {
	int nd;
	int dim[CCV_NNC_MAX_DIM_ALLOC];
	int idx[CCV_NNC_MAX_DIM_ALLOC];
	template <class T> void init(const matp_t<T>& ref) { nd = ref.outer_nd; for (int i = 0; i < nd; i++) { dim[i] = ref.outer_dim[i]; idx[i] = 0; } }
	template <class T> bool compatible(const matp_t<T>& m) const
	{
		if (m.outer_nd > nd) return false;
		for (int i = 0; i < m.outer_nd; i++) { const int e = m.outer_dim[i], r = dim[nd - m.outer_nd + i]; if (e != r && e != 1) return false; }
		return true;
	}
	template <class T> long offset(const matp_t<T>& m) const
	{
		long o = 0;
		for (int i = 0; i < m.outer_nd; i++) if (m.outer_dim[i] != 1) o += (long)idx[nd - m.outer_nd + i] * m.outer_inc[i];
		return o;
	}
	// true when this operand is broadcast over some outer axis (an output then accumulates over that axis)
	template <class T> bool revisits(const matp_t<T>& m) const
	{
		for (int i = 0; i < nd; i++) {
			const int j = i - (nd - m.outer_nd);
			if ((j < 0 || m.outer_dim[j] == 1) && dim[i] > 1 && idx[i] > 0) return true;
		}
		return false;
	}
	bool next() { for (int i = nd - 1; i >= 0; i--) { if (++idx[i] < dim[i]) return true; idx[i] = 0; } return false; }
};

// T = float or half_t (every tensor of the command of that type).  check_only: run nothing, return CCV_NNC_EXEC_SUCCESS iff every
// contraction of the command can be read in 4-element chunks -- the only form the half-precision core has; the caller routes
// the command through the fp32 images otherwise (half_stage.cpp).
template <class T>
static int gemm_forw_t(const ccv_nnc_cmd_t cmd, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context, const bool check_only)
{
	typedef typename gemm_out_of<T>::type Out;
	if (input_size < 2 || output_size < 1 || !inputs[0] || !inputs[1] || !outputs[0]) return CCV_NNC_EXEC_INVALID;
	const ccv_nnc_tensor_t* a = inputs[0];
	const ccv_nnc_tensor_t* w = inputs[1];
	const ccv_nnc_tensor_t* bias = input_size > 2 ? inputs[2] : 0;
	ccv_nnc_tensor_t* b = outputs[0];
	matp_t<T> am, wm, bm, sm;
	if (!matrix_params(a, cmd.info.blas.transpose_a, &am) || !matrix_params(w, cmd.info.blas.transpose_b, &wm) || !matrix_params(b, no_transpose, &bm)) return CCV_NNC_EXEC_INVALID;
	if ((am.batch > wm.batch ? am.batch : wm.batch) != bm.batch || am.rows != bm.rows || am.cols != wm.rows || wm.cols != bm.cols) return CCV_NNC_EXEC_INVALID;
	if ((am.batch != bm.batch && am.batch != 1) || (wm.batch != bm.batch && wm.batch != 1)) return CCV_NNC_EXEC_INVALID;
	if (am.batch == 1) am.batch_inc = 0;
	if (wm.batch == 1) wm.batch_inc = 0;
	long bias_z = 0, bias_ldm = 0;
	if (bias) { // a row vector, or a full rows x cols matrix, optionally batched (gemm_cpu_ref.c:158-172)
		if (!matrix_params(bias, no_transpose, &sm) || sm.cols != bm.cols || sm.cols_inc != 1) return CCV_NNC_EXEC_INVALID;
		if ((sm.batch != 1 && sm.batch != bm.batch) || (sm.rows != 1 && sm.rows != bm.rows)) return CCV_NNC_EXEC_INVALID;
		bias_z = sm.batch == 1 ? 0 : sm.batch_inc;
		bias_ldm = sm.rows == 1 ? 0 : sm.rows_inc;
	}
	outer_walk_t walk;
	walk.init(bm);
	if (!walk.compatible(am) || !walk.compatible(wm) || (bias && !walk.compatible(sm))) return CCV_NNC_EXEC_INVALID;
	do {
		const MatOperandT<T> A = { am.p + walk.offset(am), am.rows_inc, am.cols_inc, am.rows, am.cols };
		const MatOperandT<T> B = { wm.p + walk.offset(wm), wm.cols_inc, wm.rows_inc, wm.cols, wm.rows }; // rows of B-as-loader are output columns
		if (check_only) { if (!gemm_strided_vec(A, B, bm.batch, am.batch_inc, wm.batch_inc)) return CCV_NNC_EXEC_NO_KERNEL; continue; }
		const Out out = { bm.p + walk.offset(bm), bm.rows_inc, bm.cols_inc, bias ? sm.p + walk.offset(sm) : 0, 1.f, 0, bias_ldm };
		const int ret = gemm_strided<T>("gemm_fwd", A, B, out, bm.batch, am.batch_inc, wm.batch_inc, bm.batch_inc, bias_z, flags, stream_context);
		if (ret != CCV_NNC_EXEC_SUCCESS) return ret;
	} while (walk.next());
	return CCV_NNC_EXEC_SUCCESS;
}

static int colsum_any(const half_t* x, long rows, int cols, long ld, half_t* out, int accumulate, ccv_nnc_stream_context_t* ctx) { return colsum_f16(x, rows, cols, ld, out, accumulate, ctx); }
static int colsum_any(const float* x, long rows, int cols, long ld, float* out, int accumulate, ccv_nnc_stream_context_t* ctx) { return colsum_f32(x, rows, cols, ld, out, accumulate, ctx); }

template <class T>
static int gemm_back_t(const ccv_nnc_cmd_t cmd, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context, const bool check_only)
{
	typedef typename gemm_out_of<T>::type Out;
	// inputs: g, a, [w]; outputs: [h], [dw], [dbias]   (lib/nnc/cmd/blas/ccv_nnc_blas.c:23-45)
	if (input_size < 2 || output_size < 1 || !inputs[0]) return CCV_NNC_EXEC_INVALID;
	const ccv_nnc_tensor_t* g = inputs[0];
	const ccv_nnc_tensor_t* a = inputs[1];
	const ccv_nnc_tensor_t* w = input_size > 2 ? inputs[2] : 0;
	ccv_nnc_tensor_t* h = outputs[0];
	ccv_nnc_tensor_t* dw = output_size > 1 ? outputs[1] : 0;
	ccv_nnc_tensor_t* dbias = output_size > 2 ? outputs[2] : 0;
	const int acc = (flags & CCV_NNC_ACCUMULATE_OUTPUT) ? 1 : 0;
	matp_t<T> gm, am, dm, hm, wm, sm;
	if (!matrix_params(g, no_transpose, &gm)) return CCV_NNC_EXEC_INVALID;
	if (dbias && (!matrix_params(dbias, no_transpose, &sm) || sm.cols != gm.cols || sm.cols_inc != 1 || sm.rows != 1 || gm.cols_inc != 1 || (sm.batch != 1 && sm.batch != gm.batch))) return CCV_NNC_EXEC_INVALID;
	if (dw) {
		if (!a || !matrix_params(a, cmd.info.blas.transpose_a, &am) || !matrix_params(dw, cmd.info.blas.transpose_b, &dm)) return CCV_NNC_EXEC_INVALID;
		if (am.rows != gm.rows || am.cols != dm.rows || dm.cols != gm.cols) return CCV_NNC_EXEC_INVALID;
		if ((am.batch != gm.batch && am.batch != 1) || (dm.batch != gm.batch && dm.batch != 1)) return CCV_NNC_EXEC_INVALID;
		if (am.batch == 1) am.batch_inc = 0;
	}
	if (h) {
		if (!w || !matrix_params(h, cmd.info.blas.transpose_a, &hm) || !matrix_params(w, cmd.info.blas.transpose_b, &wm)) return CCV_NNC_EXEC_INVALID;
		if (hm.cols != wm.rows || wm.cols != gm.cols || hm.rows != gm.rows) return CCV_NNC_EXEC_INVALID;
		if ((hm.batch != gm.batch && hm.batch != 1) || (wm.batch != gm.batch && wm.batch != 1)) return CCV_NNC_EXEC_INVALID;
		if (wm.batch == 1) wm.batch_inc = 0;
	}
	outer_walk_t walk;
	walk.init(gm);
	if ((dbias && !walk.compatible(sm)) || (dw && (!walk.compatible(am) || !walk.compatible(dm))) || (h && (!walk.compatible(hm) || !walk.compatible(wm)))) return CCV_NNC_EXEC_INVALID;
	int ret;
	do {
		const T* gp = gm.p + walk.offset(gm);
		if (dbias && !check_only) {
			T* sp = sm.p + walk.offset(sm);
			const int acc_o = acc || walk.revisits(sm);
			for (int z = 0; z < gm.batch; z++) {
				T* dst = sp + (sm.batch == 1 ? 0 : (long)z * sm.batch_inc);
				const int accz = acc_o || (sm.batch == 1 && z > 0);
				if ((ret = colsum_any(gp + (long)z * gm.batch_inc, gm.rows, gm.cols, gm.rows_inc, dst, accz, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
			}
		}
		if (dw) {
			// dw(k, n) = sum_m a(m, k) * g(m, n)
			const int acc_o = acc || walk.revisits(dm);
			const MatOperandT<T> A = { am.p + walk.offset(am), am.cols_inc, am.rows_inc, am.cols, am.rows };
			const MatOperandT<T> B = { gp, gm.cols_inc, gm.rows_inc, gm.cols, gm.rows };
			T* dp = dm.p + walk.offset(dm);
			if (dm.batch == gm.batch) {
				if (check_only) { if (!gemm_strided_vec(A, B, gm.batch, am.batch_inc, gm.batch_inc)) return CCV_NNC_EXEC_NO_KERNEL; }
				else {
					const Out out = { dp, dm.rows_inc, dm.cols_inc, 0, 1.f, acc_o, 0 };
					if ((ret = gemm_strided<T>("gemm_dw", A, B, out, gm.batch, am.batch_inc, gm.batch_inc, dm.batch_inc, 0, flags, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
				}
			} else { // shared weight across the batch: accumulate every batch entry into the single dw
				for (int z = 0; z < gm.batch; z++) {
					MatOperandT<T> Az = A, Bz = B;
					Az.p += (long)z * am.batch_inc; Bz.p += (long)z * gm.batch_inc;
					if (check_only) { if (!gemm_strided_vec(Az, Bz, 1, 0, 0)) return CCV_NNC_EXEC_NO_KERNEL; continue; }
					const Out out = { dp, dm.rows_inc, dm.cols_inc, 0, 1.f, acc_o || z > 0, 0 };
					if ((ret = gemm_strided<T>("gemm_dw", Az, Bz, out, 1, 0, 0, 0, 0, flags, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
				}
			}
		}
		if (h) {
			// h(m, k) = sum_n g(m, n) * w(k, n)
			const int acc_o = acc || walk.revisits(hm);
			const MatOperandT<T> A = { gp, gm.rows_inc, gm.cols_inc, gm.rows, gm.cols };
			const MatOperandT<T> B = { wm.p + walk.offset(wm), wm.rows_inc, wm.cols_inc, wm.rows, wm.cols };
			T* hp = hm.p + walk.offset(hm);
			if (hm.batch == gm.batch) {
				if (check_only) { if (!gemm_strided_vec(A, B, gm.batch, gm.batch_inc, wm.batch_inc)) return CCV_NNC_EXEC_NO_KERNEL; }
				else {
					const Out out = { hp, hm.rows_inc, hm.cols_inc, 0, 1.f, acc_o, 0 };
					if ((ret = gemm_strided<T>("gemm_dx", A, B, out, gm.batch, gm.batch_inc, wm.batch_inc, hm.batch_inc, 0, flags, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
				}
			} else {
				for (int z = 0; z < gm.batch; z++) {
					MatOperandT<T> Az = A, Bz = B;
					Az.p += (long)z * gm.batch_inc; Bz.p += (long)z * wm.batch_inc;
					if (check_only) { if (!gemm_strided_vec(Az, Bz, 1, 0, 0)) return CCV_NNC_EXEC_NO_KERNEL; continue; }
					const Out out = { hp, hm.rows_inc, hm.cols_inc, 0, 1.f, acc_o || z > 0, 0 };
					if ((ret = gemm_strided<T>("gemm_dx", Az, Bz, out, 1, 0, 0, 0, 0, flags, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
				}
			}
		}
	} while (walk.next());
	return CCV_NNC_EXEC_SUCCESS;
}

// 1 = every tensor CCV_32F, 2 = every tensor CCV_16F, 0 = mixed / other
static int uniform_float_type(ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size)
{
	int seen = 0;
	for (int i = 0; i < input_size + output_size; i++) {
		const ccv_nnc_tensor_t* t = i < input_size ? inputs[i] : outputs[i - input_size];
		if (!t) continue;
		const int dt = CCV_GET_DATA_TYPE(t->info.datatype);
		seen |= dt == CCV_32F ? 1 : (dt == CCV_16F ? 2 : 4);
	}
	return seen == 1 ? 1 : (seen == 2 ? 2 : 0);
}

static int _gemm_forw_f32(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	return gemm_forw_t<float>(cmd, flags, inputs, input_size, outputs, output_size, stream_context, false);
}
static int _gemm_back_f32(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	return gemm_back_t<float>(cmd, flags, inputs, input_size, outputs, output_size, stream_context, false);
}
// fp32 -> the fp32 core; half precision throughout and readable in 4-element chunks -> the half-precision core
// (v_mfma_f32_32x32x16_f16); anything else with a half tensor (mixed types, odd strides) -> the fp32 core on fp32 images
static int _gemm_forw(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	MarkerScope marker(cmd.cmd);
	const int ut = uniform_float_type(inputs, input_size, outputs, output_size);
	if (ut == 1) return _gemm_forw_f32(cmd, hint, flags, inputs, input_size, outputs, output_size, stream_context);
	if (ut == 2 && gemm_forw_t<half_t>(cmd, flags, inputs, input_size, outputs, output_size, stream_context, true) == CCV_NNC_EXEC_SUCCESS)
		return gemm_forw_t<half_t>(cmd, flags, inputs, input_size, outputs, output_size, stream_context, false);
	if (!any_half_tensor(inputs, input_size, outputs, output_size)) return CCV_NNC_EXEC_INVALID;
	return half_staged_exec(_gemm_forw_f32, cmd, hint, flags, inputs, input_size, outputs, output_size, stream_context);
}
static int gemm_back_any(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context);
static int _gemm_back(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	const int r = gemm_back_any(cmd, hint, flags, inputs, input_size, outputs, output_size, stream_context);
	// the weight / bias gradients just enqueued (cmd_comm.cpp "Overlap": deployment (b)'s all-reduce of each starts behind its own writer)
	if (r == CCV_NNC_EXEC_SUCCESS && g_comm_overlap_on.load(std::memory_order_relaxed))
	{
		if (flags & CCV_NNC_ACCUMULATE_OUTPUT) { for (int i = 1; i < output_size && i < 3; i++) comm_gradient_touched(outputs[i]); }
		else if (output_size > 1) comm_gradients_written(outputs + 1, output_size > 3 ? 2 : output_size - 1, stream_context);
	}
	return r;
}
static int gemm_back_any(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	MarkerScope marker(cmd.cmd);
	const int ut = uniform_float_type(inputs, input_size, outputs, output_size);
	if (ut == 1) return _gemm_back_f32(cmd, hint, flags, inputs, input_size, outputs, output_size, stream_context);
	if (ut == 2 && gemm_back_t<half_t>(cmd, flags, inputs, input_size, outputs, output_size, stream_context, true) == CCV_NNC_EXEC_SUCCESS)
		return gemm_back_t<half_t>(cmd, flags, inputs, input_size, outputs, output_size, stream_context, false);
	if (!any_half_tensor(inputs, input_size, outputs, output_size)) return CCV_NNC_EXEC_INVALID;
	return half_staged_exec(_gemm_back_f32, cmd, hint, flags, inputs, input_size, outputs, output_size, stream_context);
}

} // namespace

extern "C" void _register_command_CCV_NNC_GEMM_FORWARD_backend_CCV_NNC_BACKEND_GPU_CUBLAS(ccv_nnc_cmd_backend_registry_t* const registry)
{
	registry->tensor_formats = CCV_TENSOR_FORMAT_NCHW | CCV_TENSOR_FORMAT_NHWC;
	registry->tensor_datatypes = CCV_32F | CCV_16F;
	registry->tensor_memory = CCV_TENSOR_GPU_MEMORY;
	registry->algorithms = 1;
	registry->exec = _gemm_forw;
	NNC_DEPALETTIZED(registry, _gemm_forw); // palettized a / w (ccv_nnc_gemm_gpu_cublas.cu:289-334)
}

extern "C" void _register_command_CCV_NNC_GEMM_BACKWARD_backend_CCV_NNC_BACKEND_GPU_CUBLAS(ccv_nnc_cmd_backend_registry_t* const registry)
{
	registry->tensor_formats = CCV_TENSOR_FORMAT_NCHW | CCV_TENSOR_FORMAT_NHWC;
	registry->tensor_datatypes = CCV_32F | CCV_16F;
	registry->tensor_memory = CCV_TENSOR_GPU_MEMORY;
	registry->algorithms = 1;
	registry->exec = _gemm_back;
	NNC_DEPALETTIZED(registry, _gemm_back); // palettized a / w (ccv_nnc_gemm_gpu_cublas.cu:658-752)
}
