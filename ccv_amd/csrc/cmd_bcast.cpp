// Broadcasting element-wise and reduction commands on gfx950: ADD, MUL (forward / backward), REDUCE_SUM, REDUCE_MEAN
// (forward / backward), and the unary / binary math commands EWDIV, EWEXP, EWLOG, EWSQRT, CLAMP.  All HBM-bound.
// Oracle semantics:
//   add     lib/nnc/cmd/blas/ccv_nnc_add_cpu_ref.c:16-300      c = p*a + q*b, size-1 dims broadcast; backward reduces g over them
//   mul     lib/nnc/cmd/blas/ccv_nnc_mul_cpu_ref.c:16-415      c = p*a*b;     backward da = p*g*b, db = p*g*a (reduced)
//   reduce  lib/nnc/cmd/reduce/ccv_nnc_reduce_sum_cpu_ref.c:16-117, ccv_nnc_reduce_mean_cpu_ref.c
//   ew      lib/nnc/cmd/ew/ccv_nnc_ew_cpu_ref.c:235-1420
// Replaces blas/gpu/ccv_nnc_{add,mul}_gpu_cudnn.cu, reduce/gpu/*.cu, ew/gpu/ccv_nnc_ew_gpu_ref.cu.
//
// One generic kernel each way, over shapes right-aligned to 4 dims with stride 0 on broadcast axes:
//   bcast_map_kernel     out[i] = f(a[ia], b[ib])               one lane per OUTPUT element (coalesced stores)
//   bcast_reduce_kernel  out[o] = sum over the reduced axes of f(x[ix], y[iy]) in row-major order -- one lane per output
//                        element, serial over the reduced sub-space: deterministic and in the reference's summation order.
//                        (Reductions on the training hot path -- bias gradients, batch-norm statistics -- do NOT come
//                        through here: they use the two-stage column reducers of cmd_ew.cpp / cmd_norm.cpp.)
#include "common.h"
#include <math.h>

using namespace nnc;

namespace {

struct shape4_t { int d[4]; long s[4]; };

// dims right-aligned to 4, element strides, stride 0 where the tensor's extent is 1 (broadcast)
static bool shape4(const ccv_nnc_tensor_t* t, shape4_t* o)
{
	const int nd = tensor_nd(t->info.dim);
	if (nd > 4) return false;
	int st[CCV_NNC_MAX_DIM_ALLOC];
	tensor_strides(t, st);
	for (int k = 0; k < 4; k++) {
		const int j = k - (4 - nd);
		o->d[k] = j >= 0 ? t->info.dim[j] : 1;
		o->s[k] = (j >= 0 && o->d[k] != 1) ? st[j] : 0;
	}
	return true;
}

struct map_args_t { int d[4]; long sa[4], sb[4], so[4]; };

template <class F>
__global__ void __launch_bounds__(256) bcast_map_kernel(F f, const float* a, const float* b, float* out, const map_args_t m, const size_t n)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += stride) {
		size_t r = idx;
		const int i3 = (int)(r % m.d[3]); r /= m.d[3];
		const int i2 = (int)(r % m.d[2]); r /= m.d[2];
		const int i1 = (int)(r % m.d[1]); r /= m.d[1];
		const int i0 = (int)r;
		const float av = a ? a[i0 * m.sa[0] + i1 * m.sa[1] + i2 * m.sa[2] + i3 * m.sa[3]] : 0.f;
		const float bv = b ? b[i0 * m.sb[0] + i1 * m.sb[1] + i2 * m.sb[2] + i3 * m.sb[3]] : 0.f;
		out[i0 * m.so[0] + i1 * m.so[1] + i2 * m.so[2] + i3 * m.so[3]] = f(av, bv);
	}
}

struct reduce_args_t { int od[4]; int rd[4]; long sx[4], sy[4], so[4]; }; // od: output extents; rd: extents of the reduced sub-space (1 where kept)

enum { RED_SUM = 0, RED_MAX = 1, RED_MIN = 2, RED_NORM2 = 3 }; // NORM2: sum of f() then square root
template <class F, int RED = RED_SUM>
__global__ void __launch_bounds__(256) bcast_reduce_kernel(F f, const float* x, const float* y, float* out, const reduce_args_t m, const size_t n)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += stride) {
		size_t r = idx;
		int o[4];
		o[3] = (int)(r % m.od[3]); r /= m.od[3];
		o[2] = (int)(r % m.od[2]); r /= m.od[2];
		o[1] = (int)(r % m.od[1]); r /= m.od[1];
		o[0] = (int)r;
		float s = RED == RED_MAX ? -INFINITY : RED == RED_MIN ? INFINITY : 0.f;
		for (int j0 = 0; j0 < m.rd[0]; j0++) for (int j1 = 0; j1 < m.rd[1]; j1++) for (int j2 = 0; j2 < m.rd[2]; j2++) for (int j3 = 0; j3 < m.rd[3]; j3++) {
			const int i0 = o[0] + j0, i1 = o[1] + j1, i2 = o[2] + j2, i3 = o[3] + j3; // a reduced axis has od == 1 -> o == 0
			const float xv = x[i0 * m.sx[0] + i1 * m.sx[1] + i2 * m.sx[2] + i3 * m.sx[3]];
			const float yv = y ? y[i0 * m.sy[0] + i1 * m.sy[1] + i2 * m.sy[2] + i3 * m.sy[3]] : 0.f;
			const float v = f(xv, yv);
			if (RED == RED_MAX) s = v > s ? v : s;
			else if (RED == RED_MIN) s = v < s ? v : s;
			else s += v;
		}
		out[o[0] * m.so[0] + o[1] * m.so[1] + o[2] * m.so[2] + o[3] * m.so[3]] = RED == RED_NORM2 ? sqrtf(s) : s;
	}
}

// Large reduced sub-spaces (ADVICE round 1 / VERDICT round 2: one lane serial over the whole reduced space starves the chip when few outputs each
// fold many elements -- a loss averaged over a batch, a norm of a parameter tensor).  Two stages, deterministic: workgroup (o, slice) folds the
// elements j = slice * per .. of output o's reduced sub-space (256 lanes striding it, then a fixed LDS tree) into partial[o][slice]; the second
// kernel folds an output's slices in order.  The serial kernel above keeps the small cases, where its order is the reference's.
template <int RED> __device__ __forceinline__ float red_id() { return RED == RED_MAX ? -INFINITY : RED == RED_MIN ? INFINITY : 0.f; }
template <int RED> __device__ __forceinline__ float red_op(const float a, const float b) { return RED == RED_MAX ? (b > a ? b : a) : RED == RED_MIN ? (b < a ? b : a) : a + b; }
template <class F, int RED>
__global__ void __launch_bounds__(256) bcast_reduce_slices_kernel(F f, const float* x, const float* y, float* partial, const reduce_args_t m, const long R, const long per, const int slices)
{
	__shared__ float red[256];
	size_t r = blockIdx.x;
	int o[4];
	o[3] = (int)(r % m.od[3]); r /= m.od[3];
	o[2] = (int)(r % m.od[2]); r /= m.od[2];
	o[1] = (int)(r % m.od[1]); r /= m.od[1];
	o[0] = (int)r;
	const long j_begin = (long)blockIdx.y * per, j_end = j_begin + per < R ? j_begin + per : R;
	float s = red_id<RED>();
	for (long j = j_begin + threadIdx.x; j < j_end; j += 256) {
		long q = j;
		const int j3 = (int)(q % m.rd[3]); q /= m.rd[3];
		const int j2 = (int)(q % m.rd[2]); q /= m.rd[2];
		const int j1 = (int)(q % m.rd[1]); q /= m.rd[1];
		const int i0 = o[0] + (int)q, i1 = o[1] + j1, i2 = o[2] + j2, i3 = o[3] + j3;
		const float xv = x[i0 * m.sx[0] + i1 * m.sx[1] + i2 * m.sx[2] + i3 * m.sx[3]];
		const float yv = y ? y[i0 * m.sy[0] + i1 * m.sy[1] + i2 * m.sy[2] + i3 * m.sy[3]] : 0.f;
		s = red_op<RED>(s, f(xv, yv));
	}
	red[threadIdx.x] = s;
	__syncthreads();
	for (int w = 128; w > 0; w >>= 1) {
		if ((int)threadIdx.x < w) red[threadIdx.x] = red_op<RED>(red[threadIdx.x], red[threadIdx.x + w]);
		__syncthreads();
	}
	if (threadIdx.x == 0) partial[(size_t)blockIdx.x * slices + blockIdx.y] = red[0];
}
template <int RED>
__global__ void __launch_bounds__(256) bcast_reduce_fold_kernel(const float* partial, float* out, const reduce_args_t m, const size_t n, const int slices)
{
	const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (idx >= n) return;
	float s = red_id<RED>();
	for (int k = 0; k < slices; k++) s = red_op<RED>(s, partial[idx * slices + k]);
	size_t r = idx;
	const int o3 = (int)(r % m.od[3]); r /= m.od[3];
	const int o2 = (int)(r % m.od[2]); r /= m.od[2];
	const int o1 = (int)(r % m.od[1]); r /= m.od[1];
	out[r * m.so[0] + o1 * m.so[1] + o2 * m.so[2] + o3 * m.so[3]] = RED == RED_NORM2 ? sqrtf(s) : s;
}
constexpr long REDUCE_TWO_STAGE_MIN = 4096; // reduced elements per output from which the two-stage path is taken

struct FAdd { float p, q; __device__ float operator()(float a, float b) const { return p * a + q * b; } };
struct FScale { float p; __device__ float operator()(float a, float) const { return p * a; } };
struct FScaleFirst { float p; __device__ float operator()(float a, float) const { return p * a; } }; // second operand only lends its shape
struct FMul { float p; __device__ float operator()(float a, float b) const { return p * a * b; } };
struct FDiv { float p; __device__ float operator()(float a, float b) const { return p * a / b; } };       // p * a / b
struct FRecip { float p; __device__ float operator()(float, float b) const { return p / b; } };           // p / b
struct FNegMulDiv { __device__ float operator()(float a, float b) const { return -a / b; } };
struct FExp { __device__ float operator()(float a, float) const { return expf(a); } };
struct FLog { __device__ float operator()(float a, float) const { return logf(a); } };
struct FSqrt { __device__ float operator()(float a, float) const { return sqrtf(a); } };
struct FCopy { __device__ float operator()(float a, float) const { return a; } };
struct FClamp { float lo, hi; int has_lo, has_hi; __device__ float operator()(float a, float) const { float v = a; if (has_hi) v = v < hi ? v : hi; if (has_lo) v = v > lo ? v : lo; return v; } };
// clamp backward (ew_cpu_ref.c:1351-): h = g unless the forward OUTPUT b sits on a bound
struct FClampBack { float lo, hi; int has_lo, has_hi; __device__ float operator()(float g, float b) const { if (has_hi && b >= hi) return 0.f; if (has_lo && b <= lo) return 0.f; return g; } };

// out = f(a, b) with a / b broadcast against out's shape
template <class F>
static int bcast_map(F f, const ccv_nnc_tensor_t* a, const ccv_nnc_tensor_t* b, ccv_nnc_tensor_t* out, ccv_nnc_stream_context_t* ctx)
{
	shape4_t so, sa, sb;
	if (!shape4(out, &so)) return CCV_NNC_EXEC_INVALID;
	if (a && !shape4(a, &sa)) return CCV_NNC_EXEC_INVALID;
	if (b && !shape4(b, &sb)) return CCV_NNC_EXEC_INVALID;
	map_args_t m;
	for (int k = 0; k < 4; k++) {
		m.d[k] = so.d[k];
		if (a && sa.d[k] != so.d[k] && sa.d[k] != 1) return CCV_NNC_EXEC_INVALID;
		if (b && sb.d[k] != so.d[k] && sb.d[k] != 1) return CCV_NNC_EXEC_INVALID;
		m.sa[k] = a ? sa.s[k] : 0; m.sb[k] = b ? sb.s[k] : 0;
		// the output may itself be a view: use its real strides (extent-1 axes contribute nothing either way)
		m.so[k] = so.s[k];
	}
	const size_t n = (size_t)m.d[0] * m.d[1] * m.d[2] * m.d[3];
	if (n == 0) return CCV_NNC_EXEC_SUCCESS;
	hipLaunchKernelGGL(HIP_KERNEL_NAME(bcast_map_kernel<F>), dim3(grid_for(n, 256)), dim3(256), 0, stream_of(ctx), f, a ? (const float*)a->data.f32 : 0, b ? (const float*)b->data.f32 : 0, out->data.f32, m, n);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}

// out[o] = sum_{reduced axes} f(x, y) over the FULL shape = the broadcast of x, y and out; x / y broadcast into it (stride 0
// on their extent-1 axes), out has extent 1 exactly on the axes that are summed away.
template <class F, int RED = RED_SUM>
static int bcast_reduce(F f, const ccv_nnc_tensor_t* x, const ccv_nnc_tensor_t* y, ccv_nnc_tensor_t* out, ccv_nnc_stream_context_t* ctx)
{
	shape4_t sx, sy, so;
	if (!shape4(x, &sx) || !shape4(out, &so) || (y && !shape4(y, &sy))) return CCV_NNC_EXEC_INVALID;
	reduce_args_t m;
	for (int k = 0; k < 4; k++) {
		int full = sx.d[k];
		if (y && sy.d[k] > full) full = sy.d[k];
		if (so.d[k] > full) full = so.d[k];
		if ((sx.d[k] != full && sx.d[k] != 1) || (y && sy.d[k] != full && sy.d[k] != 1) || (so.d[k] != full && so.d[k] != 1)) return CCV_NNC_EXEC_INVALID;
		m.od[k] = so.d[k];
		m.rd[k] = so.d[k] == full ? 1 : full;
		m.sx[k] = sx.s[k]; m.sy[k] = y ? sy.s[k] : 0; m.so[k] = so.s[k];
	}
	const size_t n = (size_t)m.od[0] * m.od[1] * m.od[2] * m.od[3];
	if (n == 0) return CCV_NNC_EXEC_SUCCESS;
	const long R = (long)m.rd[0] * m.rd[1] * m.rd[2] * m.rd[3];
	if (R >= REDUCE_TWO_STAGE_MIN && n <= 0x7fffffffUL) {
		// enough workgroups to fill the chip: slices per output so that n * slices >= ~4 per CU, each slice at least 2048 elements
		long slices = (4L * device_cu_count() + (long)n - 1) / (long)n;
		const long max_slices = (R + 2047) / 2048;
		if (slices > max_slices) slices = max_slices;
		if (slices > 65535) slices = 65535;
		if (slices < 1) slices = 1;
		const long per = (R + slices - 1) / slices;
		slices = (R + per - 1) / per;
		float* const partial = (float*)workspace_of(ctx, sizeof(float) * n * (size_t)slices);
		if (!partial) return CCV_NNC_EXEC_OOM;
		hipStream_t stream = stream_of(ctx);
		hipLaunchKernelGGL(HIP_KERNEL_NAME(bcast_reduce_slices_kernel<F, RED>), dim3((unsigned)n, (unsigned)slices), dim3(256), 0, stream, f, (const float*)x->data.f32, y ? (const float*)y->data.f32 : 0, partial, m, R, per, (int)slices);
		HIP_ENFORCE(hipGetLastError());
		hipLaunchKernelGGL(HIP_KERNEL_NAME(bcast_reduce_fold_kernel<RED>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const float*)partial, out->data.f32, m, n, (int)slices);
		HIP_ENFORCE(hipGetLastError());
		return CCV_NNC_EXEC_SUCCESS;
	}
	hipLaunchKernelGGL(HIP_KERNEL_NAME(bcast_reduce_kernel<F, RED>), dim3(grid_for(n, 256)), dim3(256), 0, stream_of(ctx), f, (const float*)x->data.f32, y ? (const float*)y->data.f32 : 0, out->data.f32, m, n);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}

static bool same_shape(const ccv_nnc_tensor_t* a, const ccv_nnc_tensor_t* b)
{
	shape4_t sa, sb;
	if (!shape4(a, &sa) || !shape4(b, &sb)) return false;
	for (int k = 0; k < 4; k++) if (sa.d[k] != sb.d[k]) return false;
	return true;
}
static bool f32(const ccv_nnc_tensor_t* t) { return !t || CCV_GET_DATA_TYPE(t->info.datatype) == CCV_32F; }

#define EXEC_ARGS const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context

static int _add_forw(EXEC_ARGS)
{
	if (input_size < 1 || output_size < 1 || !inputs[0] || !outputs[0] || !f32(inputs[0])) return CCV_NNC_EXEC_INVALID;
	const ccv_nnc_tensor_t* b = input_size > 1 ? inputs[1] : 0;
	if (!b) { FScale f; f.p = cmd.info.blas.a[0]; return bcast_map(f, inputs[0], 0, outputs[0], stream_context); }
	FAdd f; f.p = cmd.info.blas.a[0]; f.q = cmd.info.blas.a[1];
	return bcast_map(f, inputs[0], b, outputs[0], stream_context);
}
// inputs (g, ...), outputs (da, db): da = p * g, db = q * g, each summed over the axes it was broadcast along; g NULL = ones
static int _add_back(EXEC_ARGS)
{
	const ccv_nnc_tensor_t* g = input_size > 0 ? inputs[0] : 0;
	const float pq[2] = { cmd.info.blas.a[0], cmd.info.blas.a[1] };
	for (int i = 0; i < 2 && i < output_size; i++) {
		ccv_nnc_tensor_t* o = outputs[i];
		if (!o) continue;
		if (!f32(o)) return CCV_NNC_EXEC_INVALID;
		int ret;
		if (!g) { if (!tensor_contiguous(o)) return CCV_NNC_EXEC_INVALID; ret = fill_f32(o->data.f32, tensor_count(o->info), pq[i], stream_context); }
		else { FScale f; f.p = pq[i]; ret = same_shape(g, o) ? bcast_map(f, g, 0, o, stream_context) : bcast_reduce(f, g, 0, o, stream_context); }
		if (ret != CCV_NNC_EXEC_SUCCESS) return ret;
	}
	return CCV_NNC_EXEC_SUCCESS;
}
static int _mul_forw(EXEC_ARGS)
{
	if (input_size < 1 || output_size < 1 || !inputs[0] || !outputs[0] || !f32(inputs[0])) return CCV_NNC_EXEC_INVALID;
	const ccv_nnc_tensor_t* b = input_size > 1 ? inputs[1] : 0;
	if (!b) { FScale f; f.p = cmd.info.blas.a[0]; return bcast_map(f, inputs[0], 0, outputs[0], stream_context); }
	FMul f; f.p = cmd.info.blas.a[0];
	return bcast_map(f, inputs[0], b, outputs[0], stream_context);
}
// inputs (g, a, b), outputs (da, db): da = p * g * b, db = p * g * a (reduced over broadcast axes); g NULL = ones
static int _mul_back(EXEC_ARGS)
{
	if (input_size < 3) return CCV_NNC_EXEC_INVALID;
	const ccv_nnc_tensor_t* g = inputs[0];
	const float p = cmd.info.blas.a[0];
	for (int i = 0; i < 2 && i < output_size; i++) {
		ccv_nnc_tensor_t* o = outputs[i];
		if (!o) continue;
		const ccv_nnc_tensor_t* other = inputs[2 - i]; // da needs b, db needs a
		if (!other || !f32(o)) return CCV_NNC_EXEC_INVALID;
		int ret;
		if (g) {
			FMul f; f.p = p;
			ret = same_shape(g, o) ? bcast_map(f, g, other, o, stream_context) : bcast_reduce(f, g, other, o, stream_context);
		} else { // g = ones of the broadcast shape of (a, b): the other operand summed over the axes `o` was broadcast along
			FScaleFirst f; f.p = p;
			const ccv_nnc_tensor_t* self = inputs[1 + i];
			ret = same_shape(other, o) && (!self || same_shape(self, o)) ? bcast_map(f, other, 0, o, stream_context) : bcast_reduce(f, other, self, o, stream_context);
		}
		if (ret != CCV_NNC_EXEC_SUCCESS) return ret;
	}
	return CCV_NNC_EXEC_SUCCESS;
}

static int _reduce_sum_forw(EXEC_ARGS)
{
	if (input_size < 1 || output_size < 1 || !inputs[0] || !outputs[0] || !f32(inputs[0])) return CCV_NNC_EXEC_INVALID;
	FCopy f;
	return bcast_reduce(f, inputs[0], 0, outputs[0], stream_context);
}
static int _reduce_sum_back(EXEC_ARGS)
{ // h = g broadcast back to the input shape (reduce_sum_cpu_ref.c:60-117); g NULL = ones
	if (output_size < 1 || !outputs[0]) return CCV_NNC_EXEC_INVALID;
	if (input_size < 1 || !inputs[0]) { if (!tensor_contiguous(outputs[0])) return CCV_NNC_EXEC_INVALID; return fill_f32(outputs[0]->data.f32, tensor_count(outputs[0]->info), 1.f, stream_context); }
	FCopy f;
	return bcast_map(f, inputs[0], 0, outputs[0], stream_context);
}
static int _reduce_mean_forw(EXEC_ARGS)
{
	if (input_size < 1 || output_size < 1 || !inputs[0] || !outputs[0] || !f32(inputs[0])) return CCV_NNC_EXEC_INVALID;
	const size_t no = tensor_count(outputs[0]->info);
	FScale f; f.p = no ? 1.f / (float)(tensor_count(inputs[0]->info) / no) : 1.f; // sum of x/count: same order as the reference up to the scaling point
	return bcast_reduce(f, inputs[0], 0, outputs[0], stream_context);
}
static int _reduce_mean_back(EXEC_ARGS)
{
	if (output_size < 1 || !outputs[0]) return CCV_NNC_EXEC_INVALID;
	ccv_nnc_tensor_t* h = outputs[0];
	const ccv_nnc_tensor_t* g = input_size > 0 ? inputs[0] : 0;
	size_t ng = 1;
	if (g) ng = tensor_count(g->info);
	else { // the reduced count comes from the command's axes
		shape4_t sh;
		if (!shape4(h, &sh)) return CCV_NNC_EXEC_INVALID;
		size_t cnt = 1;
		const int nd = tensor_nd(h->info.dim);
		for (int i = 0; i < cmd.info.reduce.count; i++) if (cmd.info.reduce.axis[i] < nd) cnt *= h->info.dim[cmd.info.reduce.axis[i]];
		if (!tensor_contiguous(h)) return CCV_NNC_EXEC_INVALID;
		return fill_f32(h->data.f32, tensor_count(h->info), 1.f / (float)cnt, stream_context);
	}
	FScale f; f.p = 1.f / (float)(tensor_count(h->info) / (ng ? ng : 1));
	return bcast_map(f, g, 0, h, stream_context);
}

static int _ewdiv_forw(EXEC_ARGS)
{ // c = a / b; a NULL = reciprocal (ew_cpu_ref.c:633)
	if (input_size < 2 || output_size < 1 || !inputs[1] || !outputs[0]) return CCV_NNC_EXEC_INVALID;
	if (!inputs[0]) { FRecip f; f.p = 1.f; return bcast_map(f, 0, inputs[1], outputs[0], stream_context); }
	FDiv f; f.p = 1.f;
	return bcast_map(f, inputs[0], inputs[1], outputs[0], stream_context);
}
// inputs (g, a, b, c = a / b), outputs (ha, hb): ha = g / b, hb = -g * c / b   (ew_cpu_ref.c:639-)
static int _ewdiv_back(EXEC_ARGS)
{
	if (input_size < 3 || output_size < 1 || !inputs[2]) return CCV_NNC_EXEC_INVALID;
	const ccv_nnc_tensor_t* g = inputs[0];
	const ccv_nnc_tensor_t* b = inputs[2];
	int ret;
	if (outputs[0]) {
		if (g) { FDiv f; f.p = 1.f; ret = bcast_map(f, g, b, outputs[0], stream_context); }
		else { FRecip f; f.p = 1.f; ret = bcast_map(f, 0, b, outputs[0], stream_context); }
		if (ret != CCV_NNC_EXEC_SUCCESS) return ret;
	}
	if (output_size > 1 && outputs[1]) {
		if (input_size < 4 || !inputs[3]) return CCV_NNC_EXEC_INVALID;
		FNegMulDiv f;
		if ((ret = bcast_map(f, inputs[3], b, outputs[1], stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
		if (g) { FMul m; m.p = 1.f; if ((ret = bcast_map(m, outputs[1], g, outputs[1], stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret; }
	}
	return CCV_NNC_EXEC_SUCCESS;
}
template <class F>
static int unary_forw(F f, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const ctx)
{
	if (input_size < 1 || output_size < 1 || !inputs[0] || !outputs[0] || !f32(inputs[0])) return CCV_NNC_EXEC_INVALID;
	return bcast_map(f, inputs[0], 0, outputs[0], ctx);
}
static int _ewexp_forw(EXEC_ARGS) { return unary_forw(FExp(), inputs, input_size, outputs, output_size, stream_context); }
static int _ewlog_forw(EXEC_ARGS) { return unary_forw(FLog(), inputs, input_size, outputs, output_size, stream_context); }
static int _ewsqrt_forw(EXEC_ARGS) { return unary_forw(FSqrt(), inputs, input_size, outputs, output_size, stream_context); }
static int _ewexp_back(EXEC_ARGS)
{ // inputs (g, _, b = exp(a)): h = g * b   (:1040-1050)
	if (input_size < 3 || output_size < 1 || !inputs[2] || !outputs[0]) return CCV_NNC_EXEC_INVALID;
	if (!inputs[0]) { FCopy f; return bcast_map(f, inputs[2], 0, outputs[0], stream_context); }
	FMul f; f.p = 1.f;
	return bcast_map(f, inputs[0], inputs[2], outputs[0], stream_context);
}
static int _ewlog_back(EXEC_ARGS)
{ // inputs (g, a): h = g / a   (:1118-1123)
	if (input_size < 2 || output_size < 1 || !inputs[1] || !outputs[0]) return CCV_NNC_EXEC_INVALID;
	if (!inputs[0]) { FRecip f; f.p = 1.f; return bcast_map(f, 0, inputs[1], outputs[0], stream_context); }
	FDiv f; f.p = 1.f;
	return bcast_map(f, inputs[0], inputs[1], outputs[0], stream_context);
}
static int _ewsqrt_back(EXEC_ARGS)
{ // inputs (g, _, b = sqrt(a)): h = 0.5 * g / b   (:1191-1196)
	if (input_size < 3 || output_size < 1 || !inputs[2] || !outputs[0]) return CCV_NNC_EXEC_INVALID;
	if (!inputs[0]) { FRecip f; f.p = 0.5f; return bcast_map(f, 0, inputs[2], outputs[0], stream_context); }
	FDiv f; f.p = 0.5f;
	return bcast_map(f, inputs[0], inputs[2], outputs[0], stream_context);
}
static int _clamp_forw(EXEC_ARGS)
{ // NaN bound = unbounded on that side (:1198-)
	FClamp f; f.lo = cmd.info.clamp.min; f.hi = cmd.info.clamp.max; f.has_lo = !isnan(f.lo); f.has_hi = !isnan(f.hi);
	if (!f.has_lo && !f.has_hi) return CCV_NNC_EXEC_INVALID;
	return unary_forw(f, inputs, input_size, outputs, output_size, stream_context);
}
static int _clamp_back(EXEC_ARGS)
{ // inputs (g, _, b): h = g where b is strictly inside the bounds, else 0; g NULL = ones   (:1351-)
	if (input_size < 3 || output_size < 1 || !inputs[2] || !outputs[0]) return CCV_NNC_EXEC_INVALID;
	FClampBack f; f.lo = cmd.info.clamp.min; f.hi = cmd.info.clamp.max; f.has_lo = !isnan(f.lo); f.has_hi = !isnan(f.hi);
	if (inputs[0]) return bcast_map(f, inputs[0], inputs[2], outputs[0], stream_context);
	int ret;
	if (!tensor_contiguous(outputs[0])) return CCV_NNC_EXEC_INVALID;
	if ((ret = fill_f32(outputs[0]->data.f32, tensor_count(outputs[0]->info), 1.f, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
	return bcast_map(f, outputs[0], inputs[2], outputs[0], stream_context);
}

// ---- REDUCE_MAX / MIN / NORM2, element-wise MIN / MAX, ARGMAX / ARGMIN --------------------------------
//   reduce max/min  lib/nnc/cmd/reduce/ccv_nnc_reduce_max_cpu_ref.c, _min_: forward = extremum over the reduced axes; backward
//                   (g, a, b) -> h = (a == b) ? g : 0 with b, g broadcast back (every position that ties receives the gradient); g absent = 1
//   reduce norm2    lib/nnc/cmd/reduce/ccv_nnc_reduce_norm2_cpu_ref.c: b = sqrt(sum a^2); backward h = g a / b
//   min / max       lib/nnc/cmd/compare/ccv_nnc_min_cpu_ref.c, _max_: c = min(a, b); backward (g, a, b) -> (ha, hb): the smaller
//                   (larger) operand takes g, a tie gives g to both
//   argmax / argmin lib/nnc/cmd/reduce/ccv_nnc_argmax_cpu_ref.c:16-80: index of the FIRST extremum along one axis, int32 or fp32
struct FSquare { __device__ float operator()(float a, float) const { return a * a; } };
struct FMin { __device__ float operator()(float a, float b) const { return a < b ? a : b; } };
struct FMax { __device__ float operator()(float a, float b) const { return a > b ? a : b; } };

static int _reduce_max_forw(EXEC_ARGS)
{
	if (input_size < 1 || output_size < 1 || !inputs[0] || !outputs[0] || !f32(inputs[0])) return CCV_NNC_EXEC_INVALID;
	return bcast_reduce<FCopy, RED_MAX>(FCopy(), inputs[0], 0, outputs[0], stream_context);
}
static int _reduce_min_forw(EXEC_ARGS)
{
	if (input_size < 1 || output_size < 1 || !inputs[0] || !outputs[0] || !f32(inputs[0])) return CCV_NNC_EXEC_INVALID;
	return bcast_reduce<FCopy, RED_MIN>(FCopy(), inputs[0], 0, outputs[0], stream_context);
}
static int _reduce_norm2_forw(EXEC_ARGS)
{
	if (input_size < 1 || output_size < 1 || !inputs[0] || !outputs[0] || !f32(inputs[0])) return CCV_NNC_EXEC_INVALID;
	return bcast_reduce<FSquare, RED_NORM2>(FSquare(), inputs[0], 0, outputs[0], stream_context);
}

// h = f(g, a, b) over h's shape, every operand broadcast into it; g may be absent (= 1)
struct map3_args_t { int d[4]; long sg[4], sa[4], sb[4], so[4]; };
template <class F>
__global__ void __launch_bounds__(256) bcast_map3_kernel(F f, const float* g, const float* a, const float* b, float* out, float* out2, const map3_args_t m, const size_t n)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += stride) {
		size_t r = idx;
		const int i3 = (int)(r % m.d[3]); r /= m.d[3];
		const int i2 = (int)(r % m.d[2]); r /= m.d[2];
		const int i1 = (int)(r % m.d[1]); r /= m.d[1];
		const int i0 = (int)r;
		const float gv = g ? g[i0 * m.sg[0] + i1 * m.sg[1] + i2 * m.sg[2] + i3 * m.sg[3]] : 1.f;
		const float av = a[i0 * m.sa[0] + i1 * m.sa[1] + i2 * m.sa[2] + i3 * m.sa[3]];
		const float bv = b[i0 * m.sb[0] + i1 * m.sb[1] + i2 * m.sb[2] + i3 * m.sb[3]];
		const long o = i0 * m.so[0] + i1 * m.so[1] + i2 * m.so[2] + i3 * m.so[3];
		float second;
		const float first = f(gv, av, bv, second);
		if (out) out[o] = first;
		if (out2) out2[o] = second;
	}
}
template <class F>
static int bcast_map3(F f, const ccv_nnc_tensor_t* g, const ccv_nnc_tensor_t* a, const ccv_nnc_tensor_t* b, ccv_nnc_tensor_t* out, ccv_nnc_tensor_t* out2, ccv_nnc_stream_context_t* ctx)
{
	ccv_nnc_tensor_t* const ref = out ? out : out2;
	if (!ref || !a || !b || !f32(g) || !f32(a) || !f32(b) || !f32(ref)) return CCV_NNC_EXEC_INVALID;
	shape4_t so, sg, sa, sb, s2;
	if (!shape4(ref, &so) || !shape4(a, &sa) || !shape4(b, &sb) || (g && !shape4(g, &sg))) return CCV_NNC_EXEC_INVALID;
	if (out && out2) { if (!shape4(out2, &s2)) return CCV_NNC_EXEC_INVALID; for (int k = 0; k < 4; k++) if (s2.d[k] != so.d[k] || s2.s[k] != so.s[k]) return CCV_NNC_EXEC_INVALID; }
	map3_args_t m;
	for (int k = 0; k < 4; k++) {
		m.d[k] = so.d[k];
		if ((sa.d[k] != so.d[k] && sa.d[k] != 1) || (sb.d[k] != so.d[k] && sb.d[k] != 1) || (g && sg.d[k] != so.d[k] && sg.d[k] != 1)) return CCV_NNC_EXEC_INVALID;
		m.sg[k] = g ? sg.s[k] : 0; m.sa[k] = sa.s[k]; m.sb[k] = sb.s[k]; m.so[k] = so.s[k];
	}
	const size_t n = (size_t)m.d[0] * m.d[1] * m.d[2] * m.d[3];
	if (n == 0) return CCV_NNC_EXEC_SUCCESS;
	hipLaunchKernelGGL(HIP_KERNEL_NAME(bcast_map3_kernel<F>), dim3(grid_for(n, 256)), dim3(256), 0, stream_of(ctx), f, g ? (const float*)g->data.f32 : 0, (const float*)a->data.f32, (const float*)b->data.f32,
		out ? out->data.f32 : 0, out2 ? out2->data.f32 : 0, m, n);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}
struct F3Select { __device__ float operator()(float g, float a, float b, float& second) const { second = 0.f; return a == b ? g : 0.f; } };
struct F3Norm2Back { __device__ float operator()(float g, float a, float b, float& second) const { second = 0.f; return g * a / b; } };
struct F3MinBack { __device__ float operator()(float g, float a, float b, float& second) const { second = a < b ? 0.f : g; return a > b ? 0.f : g; } };
struct F3MaxBack { __device__ float operator()(float g, float a, float b, float& second) const { second = a > b ? 0.f : g; return a < b ? 0.f : g; } };

static int _reduce_extremum_back(EXEC_ARGS)
{ // (g, a, b) -> h
	if (input_size < 3 || output_size < 1) return CCV_NNC_EXEC_INVALID;
	return bcast_map3(F3Select(), inputs[0], inputs[1], inputs[2], outputs[0], 0, stream_context);
}
static int _reduce_norm2_back(EXEC_ARGS)
{
	if (input_size < 3 || output_size < 1) return CCV_NNC_EXEC_INVALID;
	return bcast_map3(F3Norm2Back(), inputs[0], inputs[1], inputs[2], outputs[0], 0, stream_context);
}
static int _min_forw(EXEC_ARGS)
{
	if (input_size < 2 || output_size < 1 || !inputs[0] || !inputs[1] || !outputs[0] || !f32(inputs[0]) || !same_shape(inputs[0], inputs[1]) || !same_shape(inputs[0], outputs[0])) return CCV_NNC_EXEC_INVALID;
	return bcast_map(FMin(), inputs[0], inputs[1], outputs[0], stream_context);
}
static int _max_forw(EXEC_ARGS)
{
	if (input_size < 2 || output_size < 1 || !inputs[0] || !inputs[1] || !outputs[0] || !f32(inputs[0]) || !same_shape(inputs[0], inputs[1]) || !same_shape(inputs[0], outputs[0])) return CCV_NNC_EXEC_INVALID;
	return bcast_map(FMax(), inputs[0], inputs[1], outputs[0], stream_context);
}
static int _min_back(EXEC_ARGS)
{ // (g, a, b) -> (ha, hb)
	if (input_size < 3 || output_size < 1) return CCV_NNC_EXEC_INVALID;
	return bcast_map3(F3MinBack(), inputs[0], inputs[1], inputs[2], outputs[0], output_size > 1 ? outputs[1] : 0, stream_context);
}
static int _max_back(EXEC_ARGS)
{
	if (input_size < 3 || output_size < 1) return CCV_NNC_EXEC_INVALID;
	return bcast_map3(F3MaxBack(), inputs[0], inputs[1], inputs[2], outputs[0], output_size > 1 ? outputs[1] : 0, stream_context);
}

template <bool MAXIMUM, class OUT>
__global__ void __launch_bounds__(256) argext_kernel(const float* a, OUT* b, const int before, const int axis_dim, const int after)
{
	const size_t n = (size_t)before * after, stride = (size_t)gridDim.x * blockDim.x;
	for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += stride) {
		const size_t i = idx / after, j = idx - i * after;
		const float* const p = a + i * (size_t)axis_dim * after + j;
		float best = p[0];
		int at = 0;
		for (int k = 1; k < axis_dim; k++) {
			const float v = p[(size_t)k * after];
			if (MAXIMUM ? v > best : v < best) { best = v; at = k; }
		}
		b[idx] = (OUT)at;
	}
}
template <bool MAXIMUM>
static int argext(const ccv_nnc_cmd_t& cmd, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const ctx)
{
	if (input_size < 1 || output_size < 1 || !inputs[0] || !outputs[0] || !tensor_contiguous(inputs[0]) || !tensor_contiguous(outputs[0]) || CCV_GET_DATA_TYPE(inputs[0]->info.datatype) != CCV_32F) return CCV_NNC_EXEC_INVALID;
	const ccv_nnc_tensor_t* a = inputs[0];
	const int axis = cmd.info.reduce.axis[0], nd = tensor_nd(a->info.dim);
	if (cmd.info.reduce.count != 1 || axis < 0 || axis >= nd) return CCV_NNC_EXEC_INVALID;
	const int axis_dim = a->info.dim[axis];
	long after = 1;
	for (int i = axis + 1; i < nd; i++) after *= a->info.dim[i];
	const size_t total = tensor_count(a->info);
	if (axis_dim <= 0 || total == 0) return CCV_NNC_EXEC_SUCCESS;
	const long before = (long)(total / axis_dim / after);
	if (tensor_count(outputs[0]->info) != total / axis_dim) return CCV_NNC_EXEC_INVALID;
	const int odt = CCV_GET_DATA_TYPE(outputs[0]->info.datatype);
	const unsigned grid = grid_for((size_t)before * after, 256);
	if (odt == CCV_32S) hipLaunchKernelGGL(HIP_KERNEL_NAME(argext_kernel<MAXIMUM, int>), dim3(grid), dim3(256), 0, stream_of(ctx), (const float*)a->data.f32, outputs[0]->data.i32, (int)before, axis_dim, (int)after);
	else if (odt == CCV_32F) hipLaunchKernelGGL(HIP_KERNEL_NAME(argext_kernel<MAXIMUM, float>), dim3(grid), dim3(256), 0, stream_of(ctx), (const float*)a->data.f32, outputs[0]->data.f32, (int)before, axis_dim, (int)after);
	else return CCV_NNC_EXEC_INVALID;
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}
// ---- MASKED_FILL (lib/nnc/cmd/util/ccv_nnc_util_cpu_ref.c:1280-1367 float mask, :1369- int32 mask; GPU file util/gpu/ccv_nnc_util_gpu_ref.cu)
// c = (mask == p) ? q : a, a and mask broadcast against each other; backward: h = (mask == p) ? 0 : g (no gradient to the mask).
struct FMaskFillF { float p, q; __device__ float operator()(float a, float b) const { return b == p ? q : a; } };
struct FMaskFillI { int p; float q; __device__ float operator()(float a, float b) const { return __float_as_int(b) == p ? q : a; } }; // the mask's bits ARE an int32
static int masked_fill(const float p, const float q, const ccv_nnc_tensor_t* a, const ccv_nnc_tensor_t* mask, ccv_nnc_tensor_t* c, ccv_nnc_stream_context_t* ctx)
{
	if (!a || !mask || !c || CCV_GET_DATA_TYPE(a->info.datatype) != CCV_32F || CCV_GET_DATA_TYPE(c->info.datatype) != CCV_32F) return CCV_NNC_EXEC_INVALID;
	const int mdt = CCV_GET_DATA_TYPE(mask->info.datatype);
	if (mdt == CCV_32S) { FMaskFillI f; f.p = (int)p; f.q = q; return bcast_map(f, a, mask, c, ctx); }
	if (mdt != CCV_32F) return CCV_NNC_EXEC_INVALID;
	FMaskFillF f; f.p = p; f.q = q;
	return bcast_map(f, a, mask, c, ctx);
}
static int _masked_fill_forw(EXEC_ARGS)
{
	if (input_size < 2 || output_size < 1) return CCV_NNC_EXEC_INVALID;
	return masked_fill(cmd.info.blas.a[0], cmd.info.blas.a[1], inputs[0], inputs[1], outputs[0], stream_context);
}
static int _masked_fill_back(EXEC_ARGS)
{ // inputs (g, a, mask): util_gpu_ref.cu:211-218
	if (input_size < 3 || output_size < 1) return CCV_NNC_EXEC_INVALID;
	return masked_fill(cmd.info.blas.a[0], 0.f, inputs[0], inputs[2], outputs[0], stream_context);
}

// ---- REDUCE_ISNAN (lib/nnc/cmd/isnan/ccv_nnc_reduce_isnan_cpu_ref.c:16-80): int32 output, 1 where any element of the reduced
// sub-space is NaN.  A max-reduction over the int32 flag carried in the float lanes (bit pattern 1 is a positive subnormal, 0 is
// zero: ordered as the ints are; fp32 subnormals are not flushed on this target).
struct FIsNan { __device__ float operator()(float x, float) const { return __int_as_float(x != x ? 1 : 0); } };
static int _reduce_isnan_forw(EXEC_ARGS)
{
	if (input_size < 1 || output_size < 1 || !inputs[0] || !outputs[0]) return CCV_NNC_EXEC_INVALID;
	if (CCV_GET_DATA_TYPE(inputs[0]->info.datatype) != CCV_32F || CCV_GET_DATA_TYPE(outputs[0]->info.datatype) != CCV_32S) return CCV_NNC_EXEC_INVALID;
	return bcast_reduce<FIsNan, RED_MAX>(FIsNan(), inputs[0], 0, outputs[0], stream_context);
}

// Backward rows the reference registers with an exec function that returns CCV_NNC_EXEC_INVALID (optimizers, argmax / argmin,
// reduce-isnan and transposed convolution have no gradient command: lib/nnc/cmd/sgd/gpu/ccv_nnc_sgd_gpu_ref.cu:97-100,
// adam/gpu/...:129-132, lamb, rmsprop, reduce/gpu/ccv_nnc_argmax_gpu_ref.cu:76-79, isnan/gpu/...:63-66,
// convolution/gpu/ccv_nnc_conv_transpose_gpu_cudnn.cu:174-177): the same here, so that ccv_nnc_cmd_ok() answers as it does there.
static int _no_gradient(EXEC_ARGS) { return CCV_NNC_EXEC_INVALID; }

static int _argmax_forw(EXEC_ARGS) { return argext<true>(cmd, inputs, input_size, outputs, output_size, stream_context); }
static int _argmin_forw(EXEC_ARGS) { return argext<false>(cmd, inputs, input_size, outputs, output_size, stream_context); }

} // namespace


#define NNC_REG(CMD, BACKEND, FORMATS, DATATYPES, MEMORY, EXEC) \
	extern "C" void _register_command_##CMD##_backend_##BACKEND(ccv_nnc_cmd_backend_registry_t* const registry) \
	{ registry->tensor_formats = (FORMATS); registry->tensor_datatypes = (DATATYPES); registry->tensor_memory = (MEMORY); registry->algorithms = 1; registry->exec = EXEC; NNC_HALF_STAGED(registry, EXEC); }
#define ALL_FORMATS (CCV_TENSOR_FORMAT_NCHW | CCV_TENSOR_FORMAT_NHWC | CCV_TENSOR_FORMAT_CHWN)

NNC_REG(CCV_NNC_ADD_FORWARD, CCV_NNC_BACKEND_GPU_CUDNN, ALL_FORMATS, CCV_32F, CCV_TENSOR_GPU_MEMORY, _add_forw)
NNC_REG(CCV_NNC_ADD_BACKWARD, CCV_NNC_BACKEND_GPU_CUDNN, ALL_FORMATS, CCV_32F, CCV_TENSOR_GPU_MEMORY, _add_back)
NNC_REG(CCV_NNC_MUL_FORWARD, CCV_NNC_BACKEND_GPU_CUDNN, ALL_FORMATS, CCV_32F, CCV_TENSOR_GPU_MEMORY, _mul_forw)
NNC_REG(CCV_NNC_MUL_BACKWARD, CCV_NNC_BACKEND_GPU_CUDNN, ALL_FORMATS, CCV_32F, CCV_TENSOR_GPU_MEMORY, _mul_back)
NNC_REG(CCV_NNC_REDUCE_SUM_FORWARD, CCV_NNC_BACKEND_GPU_CUDNN, ALL_FORMATS, CCV_32F, CCV_TENSOR_GPU_MEMORY, _reduce_sum_forw)
NNC_REG(CCV_NNC_REDUCE_SUM_BACKWARD, CCV_NNC_BACKEND_GPU_CUDNN, ALL_FORMATS, CCV_32F, CCV_TENSOR_GPU_MEMORY, _reduce_sum_back)
NNC_REG(CCV_NNC_REDUCE_MEAN_FORWARD, CCV_NNC_BACKEND_GPU_CUDNN, ALL_FORMATS, CCV_32F, CCV_TENSOR_GPU_MEMORY, _reduce_mean_forw)
NNC_REG(CCV_NNC_REDUCE_MEAN_BACKWARD, CCV_NNC_BACKEND_GPU_CUDNN, ALL_FORMATS, CCV_32F, CCV_TENSOR_GPU_MEMORY, _reduce_mean_back)
NNC_REG(CCV_NNC_EWDIV_FORWARD, CCV_NNC_BACKEND_GPU_REF, ALL_FORMATS, CCV_32F, CCV_TENSOR_GPU_MEMORY, _ewdiv_forw)
NNC_REG(CCV_NNC_EWDIV_BACKWARD, CCV_NNC_BACKEND_GPU_REF, ALL_FORMATS, CCV_32F, CCV_TENSOR_GPU_MEMORY, _ewdiv_back)
NNC_REG(CCV_NNC_EWEXP_FORWARD, CCV_NNC_BACKEND_GPU_REF, ALL_FORMATS, CCV_32F, CCV_TENSOR_GPU_MEMORY, _ewexp_forw)
NNC_REG(CCV_NNC_EWEXP_BACKWARD, CCV_NNC_BACKEND_GPU_REF, ALL_FORMATS, CCV_32F, CCV_TENSOR_GPU_MEMORY, _ewexp_back)
NNC_REG(CCV_NNC_EWLOG_FORWARD, CCV_NNC_BACKEND_GPU_REF, ALL_FORMATS, CCV_32F, CCV_TENSOR_GPU_MEMORY, _ewlog_forw)
NNC_REG(CCV_NNC_EWLOG_BACKWARD, CCV_NNC_BACKEND_GPU_REF, ALL_FORMATS, CCV_32F, CCV_TENSOR_GPU_MEMORY, _ewlog_back)
NNC_REG(CCV_NNC_EWSQRT_FORWARD, CCV_NNC_BACKEND_GPU_REF, ALL_FORMATS, CCV_32F, CCV_TENSOR_GPU_MEMORY, _ewsqrt_forw)
NNC_REG(CCV_NNC_EWSQRT_BACKWARD, CCV_NNC_BACKEND_GPU_REF, ALL_FORMATS, CCV_32F, CCV_TENSOR_GPU_MEMORY, _ewsqrt_back)
NNC_REG(CCV_NNC_CLAMP_FORWARD, CCV_NNC_BACKEND_GPU_REF, ALL_FORMATS, CCV_32F, CCV_TENSOR_GPU_MEMORY, _clamp_forw)
NNC_REG(CCV_NNC_CLAMP_BACKWARD, CCV_NNC_BACKEND_GPU_REF, ALL_FORMATS, CCV_32F, CCV_TENSOR_GPU_MEMORY, _clamp_back)
/* REDUCE_MAX / REDUCE_MIN have no GPU row in the reference host's table (lib/nnc/cmd/ccv_nnc_cmd.inc:944-1075): their exec
 * functions above are reachable through nnc_mi355x_cmd_exec only once a host registers them; not registered here. */
NNC_REG(CCV_NNC_REDUCE_NORM2_FORWARD, CCV_NNC_BACKEND_GPU_CUDNN, ALL_FORMATS, CCV_32F, CCV_TENSOR_GPU_MEMORY, _reduce_norm2_forw)
NNC_REG(CCV_NNC_REDUCE_NORM2_BACKWARD, CCV_NNC_BACKEND_GPU_CUDNN, ALL_FORMATS, CCV_32F, CCV_TENSOR_GPU_MEMORY, _reduce_norm2_back)
NNC_REG(CCV_NNC_MIN_FORWARD, CCV_NNC_BACKEND_GPU_REF, ALL_FORMATS, CCV_32F, CCV_TENSOR_GPU_MEMORY, _min_forw)
NNC_REG(CCV_NNC_MIN_BACKWARD, CCV_NNC_BACKEND_GPU_REF, ALL_FORMATS, CCV_32F, CCV_TENSOR_GPU_MEMORY, _min_back)
NNC_REG(CCV_NNC_MAX_FORWARD, CCV_NNC_BACKEND_GPU_REF, ALL_FORMATS, CCV_32F, CCV_TENSOR_GPU_MEMORY, _max_forw)
NNC_REG(CCV_NNC_MAX_BACKWARD, CCV_NNC_BACKEND_GPU_REF, ALL_FORMATS, CCV_32F, CCV_TENSOR_GPU_MEMORY, _max_back)
NNC_REG(CCV_NNC_ARGMAX_FORWARD, CCV_NNC_BACKEND_GPU_REF, ALL_FORMATS, CCV_32F | CCV_32S, CCV_TENSOR_GPU_MEMORY, _argmax_forw)
NNC_REG(CCV_NNC_ARGMIN_FORWARD, CCV_NNC_BACKEND_GPU_REF, ALL_FORMATS, CCV_32F | CCV_32S, CCV_TENSOR_GPU_MEMORY, _argmin_forw)
NNC_REG(CCV_NNC_MASKED_FILL_FORWARD, CCV_NNC_BACKEND_GPU_REF, ALL_FORMATS, CCV_32F | CCV_32S, CCV_TENSOR_GPU_MEMORY, _masked_fill_forw)
NNC_REG(CCV_NNC_MASKED_FILL_BACKWARD, CCV_NNC_BACKEND_GPU_REF, ALL_FORMATS, CCV_32F | CCV_32S, CCV_TENSOR_GPU_MEMORY, _masked_fill_back)
NNC_REG(CCV_NNC_REDUCE_ISNAN_FORWARD, CCV_NNC_BACKEND_GPU_CUDNN, ALL_FORMATS, CCV_32F | CCV_32S, CCV_TENSOR_GPU_MEMORY, _reduce_isnan_forw)
#define NNC_REG_NO_GRADIENT(CMD, BACKEND) \
	extern "C" void _register_command_##CMD##_backend_##BACKEND(ccv_nnc_cmd_backend_registry_t* const registry) \
	{ registry->tensor_formats = ALL_FORMATS; registry->tensor_datatypes = CCV_32F | CCV_16F | CCV_32S; registry->tensor_memory = CCV_TENSOR_GPU_MEMORY; registry->algorithms = 1; registry->exec = _no_gradient; }
NNC_REG_NO_GRADIENT(CCV_NNC_REDUCE_ISNAN_BACKWARD, CCV_NNC_BACKEND_GPU_CUDNN)
NNC_REG_NO_GRADIENT(CCV_NNC_SGD_BACKWARD, CCV_NNC_BACKEND_GPU_REF)
NNC_REG_NO_GRADIENT(CCV_NNC_ADAM_BACKWARD, CCV_NNC_BACKEND_GPU_REF)
NNC_REG_NO_GRADIENT(CCV_NNC_ADAMW_BACKWARD, CCV_NNC_BACKEND_GPU_REF)
NNC_REG_NO_GRADIENT(CCV_NNC_LAMB_BACKWARD, CCV_NNC_BACKEND_GPU_REF)
NNC_REG_NO_GRADIENT(CCV_NNC_RMSPROP_BACKWARD, CCV_NNC_BACKEND_GPU_REF)
NNC_REG_NO_GRADIENT(CCV_NNC_ARGMAX_BACKWARD, CCV_NNC_BACKEND_GPU_REF)
NNC_REG_NO_GRADIENT(CCV_NNC_ARGMIN_BACKWARD, CCV_NNC_BACKEND_GPU_REF)
NNC_REG_NO_GRADIENT(CCV_NNC_CONVOLUTION_TRANSPOSE_BACKWARD, CCV_NNC_BACKEND_GPU_CUDNN)
