// UPSAMPLE (nearest / bilinear, align_corners or not), forward and backward, NCHW and NHWC, on gfx950 -- the
// resolution changes of the diffusion UNets.  Oracle: lib/nnc/cmd/upsample/ccv_nnc_upsample_cpu_ref.c
//   nearest  :16-208   source index = min((int)((xd + 0.5) r), A - 1), or (int)(xd r + 0.5) with align_corners; backward adds every output
//                      gradient into its source
//   bilinear :214-526  per axis (_ccv_nnc_init_bi_coeffs :218-243): xs = (i + 0.5) s - 0.5 (or i s), taps si0 = (int)xs, si1 = min((int)(xs + 1), A - 1),
//                      weights sc1 = xs - si0, sc0 = 1 - sc1 (xs < 0 gives a NEGATIVE sc1: kept); backward scatters g with the same weights
// Every coefficient is recomputed per thread with the reference's exact mixed float / double expression, so the taps agree bit for bit.
// Backward is a GATHER: one thread per input-gradient element walks the (few) outputs that reference it in (yd, xd, tap) order -- the order
// the reference's sequential scatter adds them -- so sums are bit-identical without atomics.  HBM-bound, 4 bytes per lane.
#include "common.h"

using namespace nnc;

namespace {

enum { UPSAMPLE_NEAREST = 0, UPSAMPLE_BILINEAR = 1 };

struct UpGeom {
	int N, C, AH, AW, BH, BW;      // a = the small (source) image, b = the resampled one
	long an, ac, ah, aw;           // element strides of a for (n, c, y, x)
	long bn, bc, bh, bw;
	float rh, rw;
	int align, nchw;
};

__device__ __forceinline__ int nearest_src(const int d, const float r, const int A, const int align)
{
#pragma clang fp contract(off)
	const int s = align ? (int)((double)((float)d * r) + 0.5) : (int)(((double)d + 0.5) * (double)r);
	return s < A - 1 ? s : A - 1;
}
struct Bi { int si0, si1; float sc0, sc1; };
__device__ __forceinline__ Bi bi_coeff(const int i, const float s, const int A, const int align)
{
#pragma clang fp contract(off)
	Bi c;
	const float xs = align ? (float)i * s : (float)(((double)i + 0.5) * (double)s - 0.5);
	c.si0 = (int)xs;
	const int t = (int)(xs + 1.f);
	c.si1 = t < A - 1 ? t : A - 1;
	c.sc1 = xs - (float)c.si0;
	c.sc0 = (float)(1.0 - (double)c.sc1);
	return c;
}
__device__ __forceinline__ void split4(size_t idx, const int d1, const int d2, const int d3, int& i0, int& i1, int& i2, int& i3)
{
	i3 = (int)(idx % d3); idx /= d3;
	i2 = (int)(idx % d2); idx /= d2;
	i1 = (int)(idx % d1); idx /= d1;
	i0 = (int)idx;
}

template <int TYPE>
__global__ void __launch_bounds__(256) upsample_forw_kernel(const UpGeom g, const float* a, float* b, const size_t total)
{
#pragma clang fp contract(off) // the reference rounds every product and sum separately; hipcc would fuse them into FMAs (last-ulp differences)
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
		int n, c, yd, xd;
		if (g.nchw) split4(idx, g.C, g.BH, g.BW, n, c, yd, xd); else split4(idx, g.BH, g.BW, g.C, n, yd, xd, c);
		const float* const ap = a + (long)n * g.an + (long)c * g.ac;
		float v;
		if (TYPE == UPSAMPLE_NEAREST) v = ap[(long)nearest_src(yd, g.rh, g.AH, g.align) * g.ah + (long)nearest_src(xd, g.rw, g.AW, g.align) * g.aw];
		else {
			const Bi y = bi_coeff(yd, g.rh, g.AH, g.align), x = bi_coeff(xd, g.rw, g.AW, g.align);
			const float a00 = ap[(long)y.si0 * g.ah + (long)x.si0 * g.aw], a01 = ap[(long)y.si0 * g.ah + (long)x.si1 * g.aw];
			const float a10 = ap[(long)y.si1 * g.ah + (long)x.si0 * g.aw], a11 = ap[(long)y.si1 * g.ah + (long)x.si1 * g.aw];
			// the two layouts of the reference round differently: NCHW multiplies tap * sc_x * sc_y (:304-305), NHWC pre-multiplies the weights (:344-355)
			if (g.nchw) v = a00 * x.sc0 * y.sc0 + a01 * x.sc1 * y.sc0 + a10 * x.sc0 * y.sc1 + a11 * x.sc1 * y.sc1;
			else v = a00 * (x.sc0 * y.sc0) + a01 * (x.sc1 * y.sc0) + a10 * (x.sc0 * y.sc1) + a11 * (x.sc1 * y.sc1);
		}
		b[(long)n * g.bn + (long)c * g.bc + (long)yd * g.bh + (long)xd * g.bw] = v;
	}
}

// candidate outputs that can reference source index s along an axis of A sources / B outputs
__device__ __forceinline__ void window(const int s, const float r, const int B, int& lo, int& hi)
{
	const float inv = r > 0.f ? 1.f / r : 1.f;
	lo = (int)floorf(((float)s - 1.5f) * inv) - 2;
	hi = (int)ceilf(((float)s + 1.5f) * inv) + 2;
	if (lo < 0) lo = 0;
	if (hi > B - 1) hi = B - 1;
}
template <int TYPE>
__global__ void __launch_bounds__(256) upsample_back_kernel(const UpGeom g, const float* b, float* a, const size_t total)
{
#pragma clang fp contract(off)
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
		int n, c, ys, xs;
		if (g.nchw) split4(idx, g.C, g.AH, g.AW, n, c, ys, xs); else split4(idx, g.AH, g.AW, g.C, n, ys, xs, c);
		const float* const bp = b + (long)n * g.bn + (long)c * g.bc;
		int ylo, yhi, xlo, xhi;
		window(ys, g.rh, g.BH, ylo, yhi);
		window(xs, g.rw, g.BW, xlo, xhi);
		float sum = 0.f;
		for (int yd = ylo; yd <= yhi; yd++) {
			if (TYPE == UPSAMPLE_NEAREST) {
				if (nearest_src(yd, g.rh, g.AH, g.align) != ys) continue;
				for (int xd = xlo; xd <= xhi; xd++)
					if (nearest_src(xd, g.rw, g.AW, g.align) == xs) sum += bp[(long)yd * g.bh + (long)xd * g.bw];
			} else {
				const Bi y = bi_coeff(yd, g.rh, g.AH, g.align);
				const bool y0 = y.si0 == ys, y1 = y.si1 == ys;
				if (!y0 && !y1) continue;
				for (int xd = xlo; xd <= xhi; xd++) {
					const Bi x = bi_coeff(xd, g.rw, g.AW, g.align);
					const bool x0 = x.si0 == xs, x1 = x.si1 == xs;
					if (!x0 && !x1) continue;
					const float gv = bp[(long)yd * g.bh + (long)xd * g.bw];
					// tap order of the reference's scatter: (y0,x0) (y0,x1) (y1,x0) (y1,x1); NCHW: g * sc_y * sc_x (:425-428), NHWC: g * (sc_x * sc_y) (:470-481)
					if (g.nchw) {
						if (y0 && x0) sum += gv * y.sc0 * x.sc0;
						if (y0 && x1) sum += gv * y.sc0 * x.sc1;
						if (y1 && x0) sum += gv * y.sc1 * x.sc0;
						if (y1 && x1) sum += gv * y.sc1 * x.sc1;
					} else {
						if (y0 && x0) sum += gv * (x.sc0 * y.sc0);
						if (y0 && x1) sum += gv * (x.sc1 * y.sc0);
						if (y1 && x0) sum += gv * (x.sc0 * y.sc1);
						if (y1 && x1) sum += gv * (x.sc1 * y.sc1);
					}
				}
			}
		}
		a[(long)n * g.an + (long)c * g.ac + (long)ys * g.ah + (long)xs * g.aw] = sum;
	}
}

static bool dense_f32(const ccv_nnc_tensor_t* t) { return t && tensor_contiguous(t) && CCV_GET_DATA_TYPE(t->info.datatype) == CCV_32F; }

// small = source-resolution tensor (forward input / backward output), big = resampled tensor
static bool geometry(const ccv_nnc_cmd_t& cmd, const ccv_nnc_tensor_t* small, const ccv_nnc_tensor_t* big, UpGeom* g)
{
	if (small->info.format != big->info.format) return false;
	const int nd = tensor_nd(small->info.dim);
	if (nd != tensor_nd(big->info.dim) || nd < 3 || nd > 4) return false;
	int ad[4], bd[4];
	for (int k = 0; k < 4; k++) { const int j = k - (4 - nd); ad[k] = j >= 0 ? small->info.dim[j] : 1; bd[k] = j >= 0 ? big->info.dim[j] : 1; }
	g->nchw = small->info.format == CCV_TENSOR_FORMAT_NCHW;
	if (g->nchw) { g->N = ad[0]; g->C = ad[1]; g->AH = ad[2]; g->AW = ad[3]; g->BH = bd[2]; g->BW = bd[3]; if (bd[0] != ad[0] || bd[1] != ad[1]) return false; }
	else { g->N = ad[0]; g->AH = ad[1]; g->AW = ad[2]; g->C = ad[3]; g->BH = bd[1]; g->BW = bd[2]; if (bd[0] != ad[0] || bd[3] != ad[3]) return false; }
	if (g->AH < 1 || g->AW < 1 || g->BH < 1 || g->BW < 1) return false;
	if (g->nchw) { g->aw = 1; g->ah = g->AW; g->ac = (long)g->AH * g->AW; g->an = g->ac * g->C; g->bw = 1; g->bh = g->BW; g->bc = (long)g->BH * g->BW; g->bn = g->bc * g->C; }
	else { g->ac = 1; g->aw = g->C; g->ah = (long)g->AW * g->C; g->an = g->ah * g->AH; g->bc = 1; g->bw = g->C; g->bh = (long)g->BW * g->C; g->bn = g->bh * g->BH; }
	g->align = cmd.info.upsample.align_corners;
	g->rh = g->align ? (float)(g->AH - 1) / (float)(g->BH - 1 > 1 ? g->BH - 1 : 1) : (float)g->AH / (float)g->BH;
	g->rw = g->align ? (float)(g->AW - 1) / (float)(g->BW - 1 > 1 ? g->BW - 1 : 1) : (float)g->AW / (float)g->BW;
	return true;
}

#define EXEC_ARGS const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context

static int _upsample_forw(EXEC_ARGS)
{
	if (input_size < 1 || output_size < 1 || !dense_f32(inputs[0]) || !dense_f32(outputs[0])) return CCV_NNC_EXEC_INVALID;
	UpGeom g;
	if (!geometry(cmd, inputs[0], outputs[0], &g)) return CCV_NNC_EXEC_INVALID;
	const int type = cmd.info.upsample.type;
	if (g.rh > 1.f || g.rw > 1.f) return CCV_NNC_EXEC_INVALID; // enlargement only: the reference asserts the same in every variant (:44-45, :273-274)
	const size_t total = tensor_count(outputs[0]->info);
	if (total == 0) return CCV_NNC_EXEC_SUCCESS;
	if (type == UPSAMPLE_NEAREST) hipLaunchKernelGGL(HIP_KERNEL_NAME(upsample_forw_kernel<UPSAMPLE_NEAREST>), dim3(grid_for(total, 256)), dim3(256), 0, stream_of(stream_context), g, (const float*)inputs[0]->data.f32, outputs[0]->data.f32, total);
	else if (type == UPSAMPLE_BILINEAR) hipLaunchKernelGGL(HIP_KERNEL_NAME(upsample_forw_kernel<UPSAMPLE_BILINEAR>), dim3(grid_for(total, 256)), dim3(256), 0, stream_of(stream_context), g, (const float*)inputs[0]->data.f32, outputs[0]->data.f32, total);
	else return CCV_NNC_EXEC_INVALID;
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}
static int _upsample_back(EXEC_ARGS)
{ // inputs[0] = gradient at the resampled resolution -> outputs[0] = gradient at the source resolution
	if (input_size < 1 || output_size < 1 || !dense_f32(inputs[0]) || !dense_f32(outputs[0])) return CCV_NNC_EXEC_INVALID;
	UpGeom g;
	if (!geometry(cmd, outputs[0], inputs[0], &g)) return CCV_NNC_EXEC_INVALID;
	const int type = cmd.info.upsample.type;
	if (g.rh > 1.f || g.rw > 1.f) return CCV_NNC_EXEC_INVALID;
	const size_t total = tensor_count(outputs[0]->info);
	if (total == 0) return CCV_NNC_EXEC_SUCCESS;
	if (type == UPSAMPLE_NEAREST) hipLaunchKernelGGL(HIP_KERNEL_NAME(upsample_back_kernel<UPSAMPLE_NEAREST>), dim3(grid_for(total, 256)), dim3(256), 0, stream_of(stream_context), g, (const float*)inputs[0]->data.f32, outputs[0]->data.f32, total);
	else if (type == UPSAMPLE_BILINEAR) hipLaunchKernelGGL(HIP_KERNEL_NAME(upsample_back_kernel<UPSAMPLE_BILINEAR>), dim3(grid_for(total, 256)), dim3(256), 0, stream_of(stream_context), g, (const float*)inputs[0]->data.f32, outputs[0]->data.f32, total);
	else return CCV_NNC_EXEC_INVALID;
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}

} // namespace

#define NNC_REG(CMD, BACKEND, EXEC) \
	extern "C" void _register_command_##CMD##_backend_##BACKEND(ccv_nnc_cmd_backend_registry_t* const registry) \
	{ registry->tensor_formats = CCV_TENSOR_FORMAT_NCHW | CCV_TENSOR_FORMAT_NHWC; registry->tensor_datatypes = CCV_32F; registry->tensor_memory = CCV_TENSOR_GPU_MEMORY; registry->algorithms = 1; registry->exec = EXEC; NNC_HALF_STAGED(registry, EXEC); }

NNC_REG(CCV_NNC_UPSAMPLE_FORWARD, CCV_NNC_BACKEND_GPU_REF, _upsample_forw)
NNC_REG(CCV_NNC_UPSAMPLE_BACKWARD, CCV_NNC_BACKEND_GPU_REF, _upsample_back)
