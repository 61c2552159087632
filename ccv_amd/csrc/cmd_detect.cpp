// Rows of the detection / signal trainers: complex multiplication, non-maximum suppression, ROI-align and
// the LSSC activation compression.  All four are small, HBM- or latency-bound kernels; none is a contraction.
// Oracle semantics (what the CPU backend computes) and the CUDA files replaced:
//   CMUL       lib/nnc/cmd/blas/ccv_nnc_cmul_cpu_ref.c:16-525                 (blas/gpu/ccv_nnc_cmul_gpu_ref.cu)
//   NMS        lib/nnc/cmd/nms/ccv_nnc_nms_cpu_ref.c:29-225                    (nms/gpu/ccv_nnc_nms_gpu_ref.cu)
//   ROI_ALIGN  lib/nnc/cmd/roi/ccv_nnc_roi_align_cpu_ref.c:18-324              (roi/gpu/ccv_nnc_roi_align_gpu_ref.cu:13-390)
//   LSSC       lib/nnc/cmd/compression/ccv_nnc_lssc_cpu_ref.c:13-150           (compression/gpu/ccv_nnc_lssc_gpu_ref.cu)
#include "common.h"
#include <float.h>

using namespace nnc;

namespace {

#define EXEC_ARGS const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context

static bool f32(const ccv_nnc_tensor_t* t) { return CCV_GET_DATA_TYPE(t->info.datatype) == CCV_32F; }

// =====================================================================================================================
// CMUL: tensors of interleaved (re, im) pairs along the last axis, broadcasting on the other (right-aligned, up to 4) axes.
//   forward   c = a * b
//   backward  da = sum over a's broadcast axes of g * conj(b), db likewise with a   (the PyTorch convention the reference cites);
//             without g: the conjugate of the other operand when nothing is broadcast, and -- as the reference's broadcasting
//             branch does (cmul_cpu_ref.c:372-420) -- the PLAIN sum of the other operand when something is.
// One thread per output pair; the reduced sub-space is walked in the reference's loop order (axis 0 outermost).
struct shape4_t { int d[4]; long s[4]; };
static bool shape4(const ccv_nnc_tensor_t* t, shape4_t* o)
{
	const int nd = tensor_nd(t->info.dim);
	if (nd > 4 || nd < 1) return false;
	int st[CCV_NNC_MAX_DIM_ALLOC];
	tensor_strides(t, st);
	for (int k = 0; k < 4; k++) {
		const int j = k - (4 - nd);
		o->d[k] = j >= 0 ? t->info.dim[j] : 1;
		o->s[k] = (j >= 0 && o->d[k] != 1) ? st[j] : 0;
	}
	return true;
}
struct cmul_args_t { int od[4]; int rd[4]; long sx[4], sy[4], so[4]; }; // axis 3 counted in PAIRS, its strides in floats per pair
enum { CM_MUL = 0, CM_MUL_CONJ = 1, CM_CONJ = 2, CM_SUM = 3 };
template <int MODE>
__global__ void __launch_bounds__(256) cmul_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out, const cmul_args_t m, const size_t n)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += stride) {
		size_t r = idx;
		int o[4];
		o[3] = (int)(r % m.od[3]); r /= m.od[3];
		o[2] = (int)(r % m.od[2]); r /= m.od[2];
		o[1] = (int)(r % m.od[1]); r /= m.od[1];
		o[0] = (int)r;
		float re = 0.f, im = 0.f;
		for (int j0 = 0; j0 < m.rd[0]; j0++) for (int j1 = 0; j1 < m.rd[1]; j1++) for (int j2 = 0; j2 < m.rd[2]; j2++) {
			const int i0 = o[0] + j0, i1 = o[1] + j1, i2 = o[2] + j2;
			const long xo = i0 * m.sx[0] + i1 * m.sx[1] + i2 * m.sx[2] + o[3] * m.sx[3];
			const float x0 = x[xo], x1 = x[xo + 1];
			if (MODE == CM_CONJ) { re += x0; im += -x1; }
			else if (MODE == CM_SUM) { re += x0; im += x1; }
			else {
				const long yo = i0 * m.sy[0] + i1 * m.sy[1] + i2 * m.sy[2] + o[3] * m.sy[3];
				const float y0 = y[yo], y1 = y[yo + 1];
				if (MODE == CM_MUL) { re += x0 * y0 - x1 * y1; im += x0 * y1 + x1 * y0; }
				else { re += x0 * y0 + x1 * y1; im += -x0 * y1 + x1 * y0; }
			}
		}
		const long oo = o[0] * m.so[0] + o[1] * m.so[1] + o[2] * m.so[2] + o[3] * m.so[3];
		out[oo] = re; out[oo + 1] = im;
	}
}
// out = reduce over the axes out is 1 on (and x / y are not) of op(x, y); *broadcast = whether any axis was reduced or broadcast
template <int MODE>
static int cmul_run(const ccv_nnc_tensor_t* x, const ccv_nnc_tensor_t* y, ccv_nnc_tensor_t* out, ccv_nnc_stream_context_t* ctx)
{
	shape4_t sx, sy, so;
	if (!f32(x) || !f32(out) || (y && !f32(y))) return CCV_NNC_EXEC_INVALID;
	if (!shape4(x, &sx) || !shape4(out, &so) || (y && !shape4(y, &sy))) return CCV_NNC_EXEC_INVALID;
	cmul_args_t m;
	for (int k = 0; k < 4; k++) {
		int full = sx.d[k];
		if (y && sy.d[k] > full) full = sy.d[k];
		if (so.d[k] > full) full = so.d[k];
		if ((sx.d[k] != full && sx.d[k] != 1) || (y && sy.d[k] != full && sy.d[k] != 1) || (so.d[k] != full && so.d[k] != 1)) return CCV_NNC_EXEC_INVALID;
		m.od[k] = so.d[k];
		m.rd[k] = so.d[k] == full ? 1 : full;
		m.sx[k] = sx.s[k]; m.sy[k] = y ? sy.s[k] : 0; m.so[k] = so.s[k];
	}
	// the pair axis: even, dense, never broadcast (cmul_cpu_ref.c:94-95)
	if ((so.d[3] & 1) || sx.d[3] != so.d[3] || (y && sy.d[3] != so.d[3]) || so.s[3] != 1 || sx.s[3] != 1 || (y && sy.s[3] != 1)) return CCV_NNC_EXEC_INVALID;
	m.od[3] = so.d[3] / 2; m.rd[3] = 1;
	m.sx[3] = 2; m.sy[3] = 2; m.so[3] = 2;
	const size_t n = (size_t)m.od[0] * m.od[1] * m.od[2] * m.od[3];
	if (n == 0) return CCV_NNC_EXEC_SUCCESS;
	hipLaunchKernelGGL(HIP_KERNEL_NAME(cmul_kernel<MODE>), dim3(grid_for(n, 256)), dim3(256), 0, stream_of(ctx), (const float*)x->data.f32, y ? (const float*)y->data.f32 : 0, out->data.f32, m, n);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}
static bool same_dims(const ccv_nnc_tensor_t* a, const ccv_nnc_tensor_t* b)
{
	shape4_t sa, sb;
	if (!shape4(a, &sa) || !shape4(b, &sb)) return false;
	for (int k = 0; k < 4; k++) if (sa.d[k] != sb.d[k]) return false;
	return true;
}
static int _cmul_forw(EXEC_ARGS)
{
	if (input_size < 2 || output_size < 1 || !inputs[0] || !inputs[1] || !outputs[0]) return CCV_NNC_EXEC_INVALID;
	return cmul_run<CM_MUL>(inputs[0], inputs[1], outputs[0], stream_context);
}
static int _cmul_back(EXEC_ARGS)
{ // inputs (g, a, b), outputs (da, db)
	if (input_size < 3 || output_size < 1) return CCV_NNC_EXEC_INVALID;
	const ccv_nnc_tensor_t* g = inputs[0];
	// "no broadcasting" as the reference decides it (cmul_cpu_ref.c:313-327): each requested output has its other operand's shape
	bool plain = true;
	for (int i = 0; i < 2 && i < output_size; i++)
		if (outputs[i]) { if (!inputs[2 - i]) return CCV_NNC_EXEC_INVALID; plain = plain && same_dims(outputs[i], inputs[2 - i]); }
	for (int i = 0; i < 2 && i < output_size; i++) {
		ccv_nnc_tensor_t* o = outputs[i];
		if (!o) continue;
		const ccv_nnc_tensor_t* other = inputs[2 - i]; // da needs b, db needs a
		int ret;
		if (g) ret = cmul_run<CM_MUL_CONJ>(g, other, o, stream_context);
		else if (plain) ret = cmul_run<CM_CONJ>(other, 0, o, stream_context);
		else ret = cmul_run<CM_SUM>(other, 0, o, stream_context);
		if (ret != CCV_NNC_EXEC_SUCCESS) return ret;
	}
	return CCV_NNC_EXEC_SUCCESS;
}

// =====================================================================================================================
// NMS: per image, m boxes of d >= 5 floats (score, x, y, w, h, ...): sort by score (descending), suppress every box whose IoU
// with an earlier surviving box reaches the threshold, move the survivors to the front; c = the original index of each
// surviving row, -1 behind them.  One workgroup per image:
//   rank   every box counts the boxes that sort before it (higher score; equal score and lower index -- the order of the
//          reference's selection sort) and writes its row to that position: m^2 / 256 comparisons per thread, no sort network
//   sweep  the reference's double loop with the inner loop spread over the workgroup; the outer loop is inherently serial
//          (whether box x suppresses anything depends on whether an earlier box suppressed x)
//   pack   one thread walks the rows (m is a few thousand at most)
__global__ void __launch_bounds__(256) nms_forw_kernel(const float* __restrict__ a, float* __restrict__ b, int* __restrict__ c, const int m, const int d, const long aninc, const long aminc, const long bninc, const long bminc, const long cninc, const float iou_threshold)
{
	const float* const ap = a + blockIdx.x * aninc;
	float* const bp = b + blockIdx.x * bninc;
	int* const cp = c + blockIdx.x * cninc;
	for (int x = threadIdx.x; x < m; x += blockDim.x) {
		const float s = ap[x * aminc];
		int rank = 0;
		for (int y = 0; y < m; y++) {
			const float u = ap[y * aminc];
			rank += (u > s || (u == s && y < x)) ? 1 : 0;
		}
		for (int k = 0; k < d; k++) bp[rank * bminc + k] = ap[x * aminc + k];
		cp[rank] = x;
	}
	for (int x = 0; x < m; x++) {
		__syncthreads();
		const float v = bp[x * bminc];
		if (v == -FLT_MAX) continue; // suppressed (the same value in every thread: read after the barrier)
		const float x1 = bp[x * bminc + 1], y1 = bp[x * bminc + 2], w1 = bp[x * bminc + 3], h1 = bp[x * bminc + 4];
		const float area1 = w1 * h1;
		for (int y = x + 1 + threadIdx.x; y < m; y += blockDim.x) {
			if (bp[y * bminc] == -FLT_MAX) continue;
			const float x2 = bp[y * bminc + 1], y2 = bp[y * bminc + 2], w2 = bp[y * bminc + 3], h2 = bp[y * bminc + 4];
			const float area2 = w2 * h2;
			const float xdiff = fmaxf(0.f, fminf(x1 + w1, x2 + w2) - fmaxf(x1, x2));
			const float ydiff = fmaxf(0.f, fminf(y1 + h1, y2 + h2) - fmaxf(y1, y2));
			const float intersection = xdiff * ydiff;
			const float iou = intersection / (area1 + area2 - intersection);
			if (iou >= iou_threshold) bp[y * bminc] = -FLT_MAX;
		}
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		int y = 0;
		for (int x = 0; x < m; x++)
			if (bp[x * bminc] != -FLT_MAX) {
				if (x != y) {
					for (int j = 0; j < 5; j++) bp[y * bminc + j] = bp[x * bminc + j];
					cp[y] = cp[x];
				}
				++y;
			}
		for (int x = y; x < m; x++) { cp[x] = -1; bp[x * bminc] = -FLT_MAX; }
	}
}
// backward: b = 0, then b[c[x]] = a[x] for the rows that survived
__global__ void __launch_bounds__(256) nms_back_kernel(const float* __restrict__ a, const int* __restrict__ c, float* __restrict__ b, const int m, const int d, const long aninc, const long aminc, const long bninc, const long bminc, const long cninc)
{
	const float* const ap = a + blockIdx.x * aninc;
	float* const bp = b + blockIdx.x * bninc;
	const int* const cp = c + blockIdx.x * cninc;
	for (int i = threadIdx.x; i < m * d; i += blockDim.x) bp[(i / d) * bminc + (i % d)] = 0.f;
	__syncthreads();
	for (int x = threadIdx.x; x < m; x += blockDim.x) {
		const int k = cp[x];
		if (k < 0) continue; // (survivors come first, so everything behind the first -1 is -1 too)
		for (int j = 0; j < d; j++) bp[k * bminc + j] = ap[x * aminc + j];
	}
}
struct nms_geom_t { int n, m, d; long aninc, aminc, bninc, bminc, cninc; };
static bool nms_geometry(const ccv_nnc_tensor_t* a, const ccv_nnc_tensor_t* b, const ccv_nnc_tensor_t* c, nms_geom_t* g)
{
	const int a_nd = tensor_nd(a->info.dim), b_nd = tensor_nd(b->info.dim), c_nd = tensor_nd(c->info.dim);
	if (a_nd != b_nd || a_nd < 1 || a_nd > 3 || c_nd < 1 || c_nd > 2) return false;
	for (int i = 0; i < a_nd; i++) if (a->info.dim[i] != b->info.dim[i]) return false;
	if (!f32(a) || !f32(b) || CCV_GET_DATA_TYPE(c->info.datatype) != CCV_32S) return false;
	int ast[CCV_NNC_MAX_DIM_ALLOC], bst[CCV_NNC_MAX_DIM_ALLOC], cst[CCV_NNC_MAX_DIM_ALLOC];
	tensor_strides(a, ast); tensor_strides(b, bst); tensor_strides(c, cst);
	g->n = a_nd >= 3 ? a->info.dim[0] : 1;
	g->m = a_nd >= 3 ? a->info.dim[1] : a->info.dim[0];
	g->d = a_nd <= 1 ? 1 : a->info.dim[a_nd - 1];
	g->aninc = a_nd >= 3 ? ast[0] : 0; g->bninc = b_nd >= 3 ? bst[0] : 0; g->cninc = c_nd >= 2 ? cst[0] : 0;
	g->aminc = a_nd >= 2 ? ast[a_nd - 2] : 1; g->bminc = b_nd >= 2 ? bst[b_nd - 2] : 1;
	if (c_nd == 1 ? (g->m != c->info.dim[0] || g->n != 1) : (g->n != c->info.dim[0] || g->m != c->info.dim[1])) return false;
	if (a_nd >= 2 && (ast[a_nd - 1] != 1 || bst[b_nd - 1] != 1)) return false;
	return true;
}
static int _nms_forw(EXEC_ARGS)
{
	if (input_size < 1 || output_size < 2 || !inputs[0] || !outputs[0] || !outputs[1]) return CCV_NNC_EXEC_INVALID;
	nms_geom_t g;
	if (!nms_geometry(inputs[0], outputs[0], outputs[1], &g) || g.d < 5) return CCV_NNC_EXEC_INVALID;
	if (g.n == 0 || g.m == 0) return CCV_NNC_EXEC_SUCCESS;
	hipLaunchKernelGGL(nms_forw_kernel, dim3(g.n), dim3(256), 0, stream_of(stream_context), (const float*)inputs[0]->data.f32, outputs[0]->data.f32, outputs[1]->data.i32, g.m, g.d, g.aninc, g.aminc, g.bninc, g.bminc, g.cninc, cmd.info.nms.iou_threshold);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}
static int _nms_back(EXEC_ARGS)
{ // inputs (gradient of the sorted boxes, ., ., ., c), output: gradient of the boxes
	if (input_size < 5 || output_size < 1 || !inputs[0] || !inputs[4] || !outputs[0]) return CCV_NNC_EXEC_INVALID;
	nms_geom_t g;
	if (!nms_geometry(inputs[0], outputs[0], inputs[4], &g)) return CCV_NNC_EXEC_INVALID;
	if (g.n == 0 || g.m == 0) return CCV_NNC_EXEC_SUCCESS;
	hipLaunchKernelGGL(nms_back_kernel, dim3(g.n), dim3(256), 0, stream_of(stream_context), (const float*)inputs[0]->data.f32, (const int*)inputs[4]->data.i32, outputs[0]->data.f32, g.m, g.d, g.aninc, g.aminc, g.bninc, g.bminc, g.cninc);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}

// =====================================================================================================================
// ROI_ALIGN: c[n] = the region b[n % b_n] = (x, y, w, h) (fractions of the map) of a[n % a_n] resampled to pool_h x pool_w: every
// output cell averages bin_h x bin_w bilinear samples (bin = ceil(roi / pool)), samples that fall outside the map are left out
// of the average.  Forward: a thread per output element (lanes along the tensor's contiguous axis).  Backward: the output
// gradient scattered with the same weights -- float atomics, like the backend being replaced (roi_align_gpu_ref.cu:197-315).
struct roi_geom_t {
	int h, w, ch, pool_h, pool_w, a_n, b_n, c_n;
	long a_sn, a_sh, a_sw, a_sc, c_sn, c_sh, c_sw, c_sc, bninc;
	int c_fast; // 1: channels are c's contiguous axis (NHWC), 0: x is (NCHW)
};
struct roi_cell_t { float roi_x, roi_y, scale_x, scale_y; int bin_h, bin_w; };
__device__ __forceinline__ roi_cell_t roi_cell(const float* __restrict__ bp, const roi_geom_t& g, const int n)
{
	const float* const r = bp + (n % g.b_n) * g.bninc;
	roi_cell_t q;
	q.roi_x = r[0] * g.w; q.roi_y = r[1] * g.h;
	const float roi_w = r[2] * g.w, roi_h = r[3] * g.h;
	q.bin_h = (int)ceilf(roi_h / g.pool_h); q.bin_w = (int)ceilf(roi_w / g.pool_w);
	q.scale_y = roi_h / (q.bin_h * g.pool_h); q.scale_x = roi_w / (q.bin_w * g.pool_w);
	return q;
}
// sample coordinate of bin i: the reference's expression, double constants and all (roi_align_cpu_ref.c:38, :71)
__device__ __forceinline__ float roi_at(const float roi, const int i, const float scale) { return (float)(roi + (i + 0.5) * scale - 0.5); }
__device__ __forceinline__ void roi_unflatten(size_t idx, const roi_geom_t& g, int& n, int& y, int& x, int& k)
{
	if (g.c_fast) { k = (int)(idx % g.ch); idx /= g.ch; x = (int)(idx % g.pool_w); idx /= g.pool_w; y = (int)(idx % g.pool_h); n = (int)(idx / g.pool_h); }
	else { x = (int)(idx % g.pool_w); idx /= g.pool_w; y = (int)(idx % g.pool_h); idx /= g.pool_h; k = (int)(idx % g.ch); n = (int)(idx / g.ch); }
}
template <bool BACK>
__global__ void __launch_bounds__(256) roi_align_kernel(const roi_geom_t g, const float* __restrict__ bp, const float* __restrict__ src, float* __restrict__ dst, const size_t total)
{ // forward: src = a, dst = c.  backward: src = gradient of c, dst = gradient of a (zeroed before the launch)
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
		int n, y, x, k;
		roi_unflatten(idx, g, n, y, x, k);
		const roi_cell_t q = roi_cell(bp, g, n);
		const int py = y * q.bin_h, px = x * q.bin_w;
		const long abase = (long)(n % g.a_n) * g.a_sn + k * g.a_sc;
		const long co = (long)n * g.c_sn + y * g.c_sh + x * g.c_sw + k * g.c_sc;
		float v = 0.f;
		int count = 0;
		if (BACK) { // the number of samples inside the map first: the gradient is divided by it
			int cy = 0, cx = 0;
			for (int by = 0; by < q.bin_h; by++) { const int iy = (int)floorf(roi_at(q.roi_y, by + py, q.scale_y)); cy += (iy + 1 < 0 || iy > g.h - 1) ? 0 : 1; }
			for (int bx = 0; bx < q.bin_w; bx++) { const int ix = (int)floorf(roi_at(q.roi_x, bx + px, q.scale_x)); cx += (ix + 1 < 0 || ix > g.w - 1) ? 0 : 1; }
			count = cy * cx;
			if (count == 0) continue;
			v = src[co] / count;
		}
		for (int by = 0; by < q.bin_h; by++) {
			const float ay = roi_at(q.roi_y, by + py, q.scale_y);
			const int iy = (int)floorf(ay);
			if (iy + 1 < 0 || iy > g.h - 1) continue;
			const float ry = ay - iy;
			const int iy0 = iy < 0 ? 0 : (iy > g.h - 1 ? g.h - 1 : iy), iy1 = iy + 1 < 0 ? 0 : (iy + 1 > g.h - 1 ? g.h - 1 : iy + 1);
			for (int bx = 0; bx < q.bin_w; bx++) {
				const float ax = roi_at(q.roi_x, bx + px, q.scale_x);
				const int ix = (int)floorf(ax);
				if (ix + 1 < 0 || ix > g.w - 1) continue;
				const float rx = ax - ix;
				const int ix0 = ix < 0 ? 0 : (ix > g.w - 1 ? g.w - 1 : ix), ix1 = ix + 1 < 0 ? 0 : (ix + 1 > g.w - 1 ? g.w - 1 : ix + 1);
				const float c00 = (1 - ry) * (1 - rx), c01 = (1 - ry) * rx, c10 = ry * (1 - rx), c11 = ry * rx;
				const long o00 = abase + iy0 * g.a_sh + ix0 * g.a_sw, o01 = abase + iy0 * g.a_sh + ix1 * g.a_sw;
				const long o10 = abase + iy1 * g.a_sh + ix0 * g.a_sw, o11 = abase + iy1 * g.a_sh + ix1 * g.a_sw;
				if (BACK) {
					atomicAdd(dst + o00, v * c00); atomicAdd(dst + o01, v * c01);
					atomicAdd(dst + o10, v * c10); atomicAdd(dst + o11, v * c11);
				} else {
					v += src[o00] * c00 + src[o01] * c01 + src[o10] * c10 + src[o11] * c11;
					++count;
				}
			}
		}
		if (!BACK) dst[co] = count > 0 ? v / count : 0.f;
	}
}
static bool roi_geometry(const ccv_nnc_tensor_t* a, const ccv_nnc_tensor_t* b, const ccv_nnc_tensor_t* c, roi_geom_t* g)
{
	Image4 ai, ci;
	if (!f32(a) || !f32(b) || !f32(c) || a->info.format != c->info.format || !image4(a, &ai) || !image4(c, &ci) || ai.c != ci.c) return false;
	const int b_nd = tensor_nd(b->info.dim);
	if (b_nd != 1 && b_nd != 2) return false;
	int bst[CCV_NNC_MAX_DIM_ALLOC];
	tensor_strides(b, bst);
	if (b->info.dim[b_nd - 1] < 4 || bst[b_nd - 1] != 1) return false;
	g->h = ai.h; g->w = ai.w; g->ch = ai.c; g->pool_h = ci.h; g->pool_w = ci.w;
	g->a_n = ai.n; g->b_n = b_nd == 1 ? 1 : b->info.dim[0]; g->c_n = ci.n;
	if (g->c_n != (g->a_n > g->b_n ? g->a_n : g->b_n)) return false;
	g->bninc = b_nd == 1 ? 0 : bst[0];
	g->a_sn = ai.sn; g->a_sh = ai.sh; g->a_sw = ai.sw; g->a_sc = ai.sc;
	g->c_sn = ci.sn; g->c_sh = ci.sh; g->c_sw = ci.sw; g->c_sc = ci.sc;
	g->c_fast = ci.sc == 1 ? 1 : 0;
	return true;
}
static int _roi_align_forw(EXEC_ARGS)
{
	if (input_size < 2 || output_size < 1 || !inputs[0] || !inputs[1] || !outputs[0]) return CCV_NNC_EXEC_INVALID;
	roi_geom_t g;
	if (!roi_geometry(inputs[0], inputs[1], outputs[0], &g)) return CCV_NNC_EXEC_INVALID;
	const size_t total = (size_t)g.c_n * g.pool_h * g.pool_w * g.ch;
	if (!total) return CCV_NNC_EXEC_SUCCESS;
	hipLaunchKernelGGL(HIP_KERNEL_NAME(roi_align_kernel<false>), dim3(grid_for(total, 256)), dim3(256), 0, stream_of(stream_context), g, (const float*)inputs[1]->data.f32, (const float*)inputs[0]->data.f32, outputs[0]->data.f32, total);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}
static int _roi_align_back(EXEC_ARGS)
{ // inputs (gradient of c, ., b), output: gradient of a
	if (input_size < 3 || output_size < 1 || !inputs[0] || !inputs[2] || !outputs[0]) return CCV_NNC_EXEC_INVALID;
	roi_geom_t g;
	if (!roi_geometry(outputs[0], inputs[2], inputs[0], &g) || !tensor_contiguous(outputs[0])) return CCV_NNC_EXEC_INVALID;
	hipStream_t stream = stream_of(stream_context);
	HIP_ENFORCE(hipMemsetAsync(outputs[0]->data.u8, 0, sizeof(float) * tensor_count(outputs[0]->info), stream));
	const size_t total = (size_t)g.c_n * g.pool_h * g.pool_w * g.ch;
	if (!total) return CCV_NNC_EXEC_SUCCESS;
	hipLaunchKernelGGL(HIP_KERNEL_NAME(roi_align_kernel<true>), dim3(grid_for(total, 256)), dim3(256), 0, stream, g, (const float*)inputs[2]->data.f32, (const float*)inputs[0]->data.f32, outputs[0]->data.f32, total);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}

// =====================================================================================================================
// LSSC (lossy 4 x 4 block compression of half-precision NCHW activations): a block becomes 4 halves -- min, max and two 16-bit
// words of sixteen 2-bit indices into {min, 2/3 min + 1/3 max, 1/3 min + 2/3 max, max}.  The arithmetic is the CPU reference's,
// operation for operation (the index scale is computed in double there, lssc_cpu_ref.c:68; its float -> half conversion is the
// table method of lib/ccv_util.c:1434-1440, which TRUNCATES -- restated in to_half_trunc), so compressed and decompressed
// tensors are bit-identical to the oracle's.  A thread per block; blocks of a row are contiguous in both tensors.
typedef _Float16 half_t;
__device__ __forceinline__ unsigned short to_half_trunc(const float f)
{
	const unsigned u = __float_as_uint(f);
	const unsigned s = (u >> 16) & 0x8000u, e = (u >> 23) & 0xffu, m = u & 0x7fffffu;
	if (e < 103) return (unsigned short)s;
	if (e < 113) return (unsigned short)(s | ((0x0400u >> (113 - e)) + (m >> (126 - e))));
	if (e < 143) return (unsigned short)(s | ((e - 112) << 10) | (m >> 13));
	if (e < 255) return (unsigned short)(s | 0x7c00u);
	return (unsigned short)(s | 0x7c00u | (m >> 13));
}
struct lssc_geom_t { int planes, H, W, BH, BW; long a_sp, a_sh, b_sp, b_sh; }; // planes = N * C; BW = blocks per row
__global__ void __launch_bounds__(256) lssc_forw_kernel(const lssc_geom_t g, const half_t* __restrict__ a, unsigned short* __restrict__ b, const size_t total)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
		const int bx = (int)(idx % g.BW), by = (int)((idx / g.BW) % g.BH);
		const long pl = (long)(idx / ((size_t)g.BW * g.BH));
		const half_t* const apz = a + pl * g.a_sp + (long)by * 4 * g.a_sh + bx * 4;
		const int h = (by * 4 + 4 < g.H ? by * 4 + 4 : g.H) - by * 4, w = (bx * 4 + 4 < g.W ? bx * 4 + 4 : g.W) - bx * 4;
		float a32[16];
		const float first = (float)apz[0];
		for (int c = 0; c < 16; c++) a32[c] = first;
		for (int j0 = 0; j0 < h; j0++) for (int j1 = 0; j1 < w; j1++) a32[j0 * 4 + j1] = (float)apz[j0 * g.a_sh + j1];
		float amax = a32[0], amin = a32[0];
		for (int c = 1; c < 16; c++) { amax = a32[c] > amax ? a32[c] : amax; amin = a32[c] < amin ? a32[c] : amin; }
		unsigned short* const bpz = b + pl * g.b_sp + (long)by * g.b_sh + bx * 4;
		bpz[0] = to_half_trunc(amin); bpz[1] = to_half_trunc(amax); // (exact: both are halves already)
		const float abottom = amin * 7 / 6 - amax / 6;
		const double spread = (double)(amax - amin);
		const float ascale = (float)(3 / (spread > 1e-6 ? spread : 1e-6));
		unsigned lo = 0, hi = 0;
		for (int c = 0; c < 8; c++) { int q = (int)((a32[c] - abottom) * ascale); q = q < 0 ? 0 : (q > 3 ? 3 : q); lo |= (unsigned)q << (c << 1); }
		for (int c = 0; c < 8; c++) { int q = (int)((a32[8 + c] - abottom) * ascale); q = q < 0 ? 0 : (q > 3 ? 3 : q); hi |= (unsigned)q << (c << 1); }
		bpz[2] = (unsigned short)lo; bpz[3] = (unsigned short)hi;
	}
}
__global__ void __launch_bounds__(256) lssc_back_kernel(const lssc_geom_t g, const unsigned short* __restrict__ b, unsigned short* __restrict__ a, const size_t total)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
		const int bx = (int)(idx % g.BW), by = (int)((idx / g.BW) % g.BH);
		const long pl = (long)(idx / ((size_t)g.BW * g.BH));
		const unsigned short* const bpz = b + pl * g.b_sp + (long)by * g.b_sh + bx * 4;
		float bm[4];
		bm[0] = (float)*(const half_t*)&bpz[0];
		bm[3] = (float)*(const half_t*)&bpz[1];
		bm[1] = bm[3] / 3 + bm[0] * 2 / 3;
		bm[2] = bm[3] * 2 / 3 + bm[0] / 3;
		unsigned short v[4];
		for (int c = 0; c < 4; c++) v[c] = to_half_trunc(bm[c]);
		const unsigned lo = bpz[2], hi = bpz[3];
		unsigned short* const apz = a + pl * g.a_sp + (long)by * 4 * g.a_sh + bx * 4;
		const int h = (by * 4 + 4 < g.H ? by * 4 + 4 : g.H) - by * 4, w = (bx * 4 + 4 < g.W ? bx * 4 + 4 : g.W) - bx * 4;
		for (int j0 = 0; j0 < h; j0++) for (int j1 = 0; j1 < w; j1++) {
			const int c = j0 * 4 + j1;
			apz[j0 * g.a_sh + j1] = v[((c < 8 ? lo >> (c << 1) : hi >> ((c - 8) << 1))) & 3];
		}
	}
}
static bool lssc_geometry(const ccv_nnc_tensor_t* a, const ccv_nnc_tensor_t* b, lssc_geom_t* g)
{ // a: the activation (.., H, W), b: its compressed form (.., ceil(H / 4), ceil(W / 4) * 4), both NCHW halves
	if (CCV_GET_DATA_TYPE(a->info.datatype) != CCV_16F || CCV_GET_DATA_TYPE(b->info.datatype) != CCV_16F) return false;
	if (a->info.format != CCV_TENSOR_FORMAT_NCHW || b->info.format != CCV_TENSOR_FORMAT_NCHW) return false;
	const int a_nd = tensor_nd(a->info.dim), b_nd = tensor_nd(b->info.dim);
	if ((a_nd != 3 && a_nd != 4) || a_nd != b_nd || !tensor_contiguous(a) || !tensor_contiguous(b)) return false;
	g->planes = 1;
	for (int i = 0; i < a_nd - 2; i++) { if (a->info.dim[i] != b->info.dim[i]) return false; g->planes *= a->info.dim[i]; }
	g->H = a->info.dim[a_nd - 2]; g->W = a->info.dim[a_nd - 1];
	g->BH = b->info.dim[b_nd - 2]; g->BW = b->info.dim[b_nd - 1] / 4;
	if (b->info.dim[b_nd - 1] % 4 || g->BH != (g->H + 3) / 4 || g->BW != (g->W + 3) / 4) return false;
	g->a_sh = g->W; g->a_sp = (long)g->H * g->W; g->b_sh = (long)g->BW * 4; g->b_sp = (long)g->BH * g->BW * 4;
	return true;
}
static int _lssc_forw(EXEC_ARGS)
{
	if (output_size > input_size) return CCV_NNC_EXEC_INVALID;
	for (int i = 0; i < output_size; i++) {
		if (!inputs[i] || !outputs[i]) return CCV_NNC_EXEC_INVALID;
		lssc_geom_t g;
		if (!lssc_geometry(inputs[i], outputs[i], &g)) return CCV_NNC_EXEC_INVALID;
		const size_t total = (size_t)g.planes * g.BH * g.BW;
		if (!total) continue;
		hipLaunchKernelGGL(lssc_forw_kernel, dim3(grid_for(total, 256)), dim3(256), 0, stream_of(stream_context), g, (const half_t*)inputs[i]->data.f16, (unsigned short*)outputs[i]->data.f16, total);
		HIP_ENFORCE(hipGetLastError());
	}
	return CCV_NNC_EXEC_SUCCESS;
}
static int _lssc_back(EXEC_ARGS)
{
	if (output_size > input_size) return CCV_NNC_EXEC_INVALID;
	for (int i = 0; i < output_size; i++) {
		if (!inputs[i] || !outputs[i]) return CCV_NNC_EXEC_INVALID;
		lssc_geom_t g;
		if (!lssc_geometry(outputs[i], inputs[i], &g)) return CCV_NNC_EXEC_INVALID;
		const size_t total = (size_t)g.planes * g.BH * g.BW;
		if (!total) continue;
		hipLaunchKernelGGL(lssc_back_kernel, dim3(grid_for(total, 256)), dim3(256), 0, stream_of(stream_context), g, (const unsigned short*)inputs[i]->data.f16, (unsigned short*)outputs[i]->data.f16, total);
		HIP_ENFORCE(hipGetLastError());
	}
	return CCV_NNC_EXEC_SUCCESS;
}

} // namespace

#define ALL_FORMATS (CCV_TENSOR_FORMAT_NCHW | CCV_TENSOR_FORMAT_NHWC | CCV_TENSOR_FORMAT_CHWN)
#define NNC_REG(CMD, BACKEND, FORMATS, DATATYPES, EXEC) \
	extern "C" void _register_command_##CMD##_backend_##BACKEND(ccv_nnc_cmd_backend_registry_t* const registry) \
	{ registry->tensor_formats = (FORMATS); registry->tensor_datatypes = (DATATYPES); registry->tensor_memory = CCV_TENSOR_GPU_MEMORY; registry->algorithms = 1; registry->exec = EXEC; NNC_HALF_STAGED(registry, EXEC); }

NNC_REG(CCV_NNC_CMUL_FORWARD, CCV_NNC_BACKEND_GPU_REF, ALL_FORMATS, CCV_32F, _cmul_forw)
NNC_REG(CCV_NNC_CMUL_BACKWARD, CCV_NNC_BACKEND_GPU_REF, ALL_FORMATS, CCV_32F, _cmul_back)
NNC_REG(CCV_NNC_NMS_FORWARD, CCV_NNC_BACKEND_GPU_REF, CCV_TENSOR_FORMAT_NCHW | CCV_TENSOR_FORMAT_NHWC, CCV_32F | CCV_32S, _nms_forw)
NNC_REG(CCV_NNC_NMS_BACKWARD, CCV_NNC_BACKEND_GPU_REF, CCV_TENSOR_FORMAT_NCHW | CCV_TENSOR_FORMAT_NHWC, CCV_32F | CCV_32S, _nms_back)
NNC_REG(CCV_NNC_ROI_ALIGN_FORWARD, CCV_NNC_BACKEND_GPU_REF, CCV_TENSOR_FORMAT_NCHW | CCV_TENSOR_FORMAT_NHWC, CCV_32F, _roi_align_forw)
NNC_REG(CCV_NNC_ROI_ALIGN_BACKWARD, CCV_NNC_BACKEND_GPU_REF, CCV_TENSOR_FORMAT_NCHW | CCV_TENSOR_FORMAT_NHWC, CCV_32F, _roi_align_back)
extern "C" void _register_command_CCV_NNC_COMPRESSION_LSSC_FORWARD_backend_CCV_NNC_BACKEND_GPU_REF(ccv_nnc_cmd_backend_registry_t* const registry)
{ registry->tensor_formats = CCV_TENSOR_FORMAT_NCHW; registry->tensor_datatypes = CCV_16F; registry->tensor_memory = CCV_TENSOR_GPU_MEMORY; registry->algorithms = 1; registry->exec = _lssc_forw; }
extern "C" void _register_command_CCV_NNC_COMPRESSION_LSSC_BACKWARD_backend_CCV_NNC_BACKEND_GPU_REF(ccv_nnc_cmd_backend_registry_t* const registry)
{ registry->tensor_formats = CCV_TENSOR_FORMAT_NCHW; registry->tensor_datatypes = CCV_16F; registry->tensor_memory = CCV_TENSOR_GPU_MEMORY; registry->algorithms = 1; registry->exec = _lssc_back; }
