// GROUP_NORM forward / backward on gfx950: statistics over blocks of the tensor, the normalisation of the
// diffusion-model UNets.  Oracle: lib/nnc/cmd/norm/ccv_nnc_group_norm_cpu_ref.c:16-222 (forward), :224-510 (backward).
// Geometry as the reference defines it, on tensors right-aligned to 4 axes: the statistics tensor saved_mean has, on every axis, an extent
// that divides the input's; element i along an axis belongs to statistic i * rdim / adim (rdim = groups on the grouped axis, 1 on the
// reduced axes, adim on the kept ones).  scale / bias map the same way through their own extents (normally one per channel).
//   forward   (a [, scale, bias]) -> (b, saved_mean, saved_inv_std): b = (a - mean) inv_std scale + bias, inv_std = 1 / sqrt(var + eps)
//   backward  (g, _, _, a, [scale, _, _,] saved_mean, saved_inv_std) -> (h [, dscale, dbias]), formulas as layer norm (cmd_rownorm.cpp)
// epsilon quirk, kept: BOTH reference backends read cmd.info.lnorm.epsilon here (group_norm_cpu_ref.c:46, gpu/ccv_nnc_group_norm_gpu_cudnn.cu:117),
// which in the parameter union overlays gnorm.reduce_count -- an int of 1..3 seen as a float, i.e. ~1e-45, not the 1e-5 the caller passed.
// Results must match the reference's, so the same field is read (with small groups the difference, eps / (2 var), reaches 5e-5).
// One 256-thread block per statistic (forward, h) or per scale element (parameter gradients); dense tensors only.  HBM-bound.
#include "common.h"

using namespace nnc;

namespace {

__device__ __forceinline__ float block_sum(float v, float* red)
{
	for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
	__syncthreads();
	if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
	__syncthreads();
	return red[0] + red[1] + red[2] + red[3];
}

struct GnGeom {
	int ad[4];  // input extents
	int rd[4];  // statistics extents
	int sd[4];  // scale extents (all 1 when absent)
	int bd[4];  // bias extents
};
struct Coord { int c[4]; };

// j-th element of the sub-block owned by index `own` of a tensor with extents od (block extents ad / od)
__device__ __forceinline__ Coord sub_coord(const GnGeom& g, const int (&od)[4], const int own, int j)
{
	Coord o;
	int r = own;
	int oc[4];
	oc[3] = r % od[3]; r /= od[3];
	oc[2] = r % od[2]; r /= od[2];
	oc[1] = r % od[1]; r /= od[1];
	oc[0] = r;
#pragma unroll
	for (int k = 3; k >= 0; k--) {
		const int e = g.ad[k] / od[k];
		o.c[k] = oc[k] * e + j % e;
		j /= e;
	}
	return o;
}
__device__ __forceinline__ long lin(const Coord& c, const int (&d)[4]) { return (((long)c.c[0] * d[1] + c.c[1]) * d[2] + c.c[2]) * d[3] + c.c[3]; }
// index into a tensor with extents pd that partitions the input block-wise
__device__ __forceinline__ long part(const GnGeom& g, const Coord& c, const int (&pd)[4])
{
	long o = 0;
#pragma unroll
	for (int k = 0; k < 4; k++) o = o * pd[k] + (long)c.c[k] * pd[k] / g.ad[k];
	return o;
}

__global__ void __launch_bounds__(256) gnorm_forw_kernel(const GnGeom g, const float* a, const float* scale, const float* bias, float* b, float* saved_mean, float* saved_inv_std, const int n, const float inv_n, const float epsilon)
{
	__shared__ float red[4];
	const int own = blockIdx.x;
	float s = 0.f;
	for (int j = threadIdx.x; j < n; j += 256) s += a[lin(sub_coord(g, g.rd, own, j), g.ad)];
	const float mean = block_sum(s, red) * inv_n;
	float v = 0.f;
	for (int j = threadIdx.x; j < n; j += 256) { const float w = a[lin(sub_coord(g, g.rd, own, j), g.ad)] - mean; v += w * w; }
	const float inv_std = 1.f / sqrtf(block_sum(v, red) * inv_n + epsilon);
	if (threadIdx.x == 0) { saved_mean[own] = mean; saved_inv_std[own] = inv_std; }
	for (int j = threadIdx.x; j < n; j += 256) {
		const Coord c = sub_coord(g, g.rd, own, j);
		const long i = lin(c, g.ad);
		float y = (a[i] - mean) * inv_std;
		if (scale) y *= scale[part(g, c, g.sd)];
		if (bias) y += bias[part(g, c, g.bd)];
		b[i] = y;
	}
}
__global__ void __launch_bounds__(256) gnorm_back_kernel(const GnGeom g, const float* gr, const float* a, const float* scale, const float* saved_mean, const float* saved_inv_std, float* h, const int n, const float inv_n)
{
	__shared__ float red[4];
	const int own = blockIdx.x;
	const float mean = saved_mean[own], inv_std = saved_inv_std[own];
	float s1 = 0.f, s2 = 0.f;
	for (int j = threadIdx.x; j < n; j += 256) {
		const Coord c = sub_coord(g, g.rd, own, j);
		const long i = lin(c, g.ad);
		const float gss = gr[i] * (scale ? scale[part(g, c, g.sd)] : 1.f) * inv_std;
		s1 += gss;
		s2 += (a[i] - mean) * inv_std * gss;
	}
	const float gssr = block_sum(s1, red), ahgssr = block_sum(s2, red);
	for (int j = threadIdx.x; j < n; j += 256) {
		const Coord c = sub_coord(g, g.rd, own, j);
		const long i = lin(c, g.ad);
		const float ah = (a[i] - mean) * inv_std;
		const float gss = gr[i] * (scale ? scale[part(g, c, g.sd)] : 1.f) * inv_std;
		h[i] = gss - inv_n * (gssr + ah * ahgssr);
	}
}
// one block per element of a parameter tensor with extents pd: out[own] = sum over its sub-block of (WITH_AH ? ah * g : g)
template <bool WITH_AH>
__global__ void __launch_bounds__(256) gnorm_param_grad_kernel(const GnGeom g, const int pd0, const int pd1, const int pd2, const int pd3, const float* gr, const float* a, const float* saved_mean, const float* saved_inv_std, float* out, const int n)
{
	__shared__ float red[4];
	const int pd[4] = { pd0, pd1, pd2, pd3 };
	float s = 0.f;
	for (int j = threadIdx.x; j < n; j += 256) {
		const Coord c = sub_coord(g, pd, blockIdx.x, j);
		const long i = lin(c, g.ad);
		if (WITH_AH) { const long r = part(g, c, g.rd); s += (a[i] - saved_mean[r]) * saved_inv_std[r] * gr[i]; }
		else s += gr[i];
	}
	s = block_sum(s, red);
	if (threadIdx.x == 0) out[blockIdx.x] = s;
}

// The same sums in the REFERENCE'S ORDER: one thread per parameter element walks its sub-block in row-major order with one running fp32 sum,
// products and sums rounded separately -- what norm/ccv_nnc_group_norm_cpu_ref.c:315-357 (dscale[k] += ah[x] * g[x], ah = (a - mean) * inv_std
// stored as a float) and the CPU reduce-sum behind dbias do.  The reference's own test compares these gradients with REQUIRE_TENSOR_EQ
// (test/int/nnc/cudnn.tests.c:1932: 128 ulp OR 1.2e-7 absolute), which a re-associated sum of +-1 terms misses on the elements that happen to
// cancel to ~0; for the sub-block sizes where a serial walk costs nothing (<= GNORM_SEQ_MAX terms) the order is therefore kept.
constexpr int GNORM_SEQ_MAX = 2048;
template <bool WITH_AH>
__global__ void __launch_bounds__(64) gnorm_param_grad_seq_kernel(const GnGeom g, const int pd0, const int pd1, const int pd2, const int pd3, const float* gr, const float* a, const float* saved_mean, const float* saved_inv_std, float* out, const int n, const int P)
{
	const int own = blockIdx.x * 64 + threadIdx.x;
	if (own >= P) return;
	const int pd[4] = { pd0, pd1, pd2, pd3 };
	float s = 0.f;
	for (int j = 0; j < n; j++) {
		const Coord c = sub_coord(g, pd, own, j);
		const long i = lin(c, g.ad);
		if (WITH_AH) {
			const long r = part(g, c, g.rd);
			const float ah = __fmul_rn(__fsub_rn(a[i], saved_mean[r]), saved_inv_std[r]);
			s = __fadd_rn(s, __fmul_rn(ah, gr[i]));
		} else s = __fadd_rn(s, gr[i]);
	}
	out[own] = s;
}

static bool dense_f32(const ccv_nnc_tensor_t* t) { return t && tensor_contiguous(t) && CCV_GET_DATA_TYPE(t->info.datatype) == CCV_32F; }
static bool dims4(const ccv_nnc_tensor_t* t, int (&d)[4])
{
	const int nd = tensor_nd(t->info.dim);
	if (nd > 4) return false;
	for (int k = 0; k < 4; k++) { const int j = k - (4 - nd); d[k] = j >= 0 ? t->info.dim[j] : 1; }
	return true;
}
static bool divides(const int (&ad)[4], const int (&pd)[4]) { for (int k = 0; k < 4; k++) if (pd[k] < 1 || ad[k] % pd[k]) return false; return true; }
static long prod(const int (&d)[4]) { return (long)d[0] * d[1] * d[2] * d[3]; }

static bool geometry(const ccv_nnc_tensor_t* a, const ccv_nnc_tensor_t* stat, const ccv_nnc_tensor_t* scale, const ccv_nnc_tensor_t* bias, GnGeom* g)
{
	const int one[4] = { 1, 1, 1, 1 };
	if (!dims4(a, g->ad) || !dims4(stat, g->rd) || !divides(g->ad, g->rd)) return false;
	for (int k = 0; k < 4; k++) { g->sd[k] = one[k]; g->bd[k] = one[k]; }
	if (scale && (!dims4(scale, g->sd) || !divides(g->ad, g->sd))) return false;
	if (bias && (!dims4(bias, g->bd) || !divides(g->ad, g->bd))) return false;
	return prod(g->ad) <= 0x7fffffffL;
}

#define EXEC_ARGS const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context

static int _group_norm_forw(EXEC_ARGS)
{
	if (input_size < 1 || output_size < 3 || !dense_f32(inputs[0]) || !dense_f32(outputs[0]) || !dense_f32(outputs[1]) || !dense_f32(outputs[2])) return CCV_NNC_EXEC_INVALID;
	const int affine = cmd.info.gnorm.elementwise_affine;
	const ccv_nnc_tensor_t* scale = affine && input_size >= 2 ? inputs[1] : 0;
	const ccv_nnc_tensor_t* bias = affine && input_size >= 3 ? inputs[2] : 0;
	if (affine && (!dense_f32(scale) || !dense_f32(bias))) return CCV_NNC_EXEC_INVALID;
	GnGeom g;
	if (!geometry(inputs[0], outputs[1], scale, bias, &g) || tensor_count(outputs[2]->info) != tensor_count(outputs[1]->info) || tensor_count(outputs[0]->info) != tensor_count(inputs[0]->info)) return CCV_NNC_EXEC_INVALID;
	const long R = prod(g.rd), total = prod(g.ad);
	if (R == 0 || total == 0) return CCV_NNC_EXEC_SUCCESS;
	const int n = (int)(total / R);
	hipLaunchKernelGGL(gnorm_forw_kernel, dim3((unsigned)R), dim3(256), 0, stream_of(stream_context), g, (const float*)inputs[0]->data.f32, scale ? (const float*)scale->data.f32 : (const float*)0, bias ? (const float*)bias->data.f32 : (const float*)0,
		outputs[0]->data.f32, outputs[1]->data.f32, outputs[2]->data.f32, n, 1.f / (float)n, cmd.info.lnorm.epsilon);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}
static int _group_norm_back(EXEC_ARGS)
{
	const int affine = cmd.info.gnorm.elementwise_affine;
	const int im = affine ? 7 : 5, is = affine ? 8 : 6;
	if (input_size <= is || output_size < 1 || !dense_f32(inputs[0]) || !dense_f32(inputs[3]) || !dense_f32(inputs[im]) || !dense_f32(inputs[is]) || (affine && !dense_f32(inputs[4]))) return CCV_NNC_EXEC_INVALID;
	const ccv_nnc_tensor_t* gr = inputs[0];
	const ccv_nnc_tensor_t* a = inputs[3];
	const ccv_nnc_tensor_t* scale = affine ? inputs[4] : 0;
	ccv_nnc_tensor_t* h = outputs[0];
	ccv_nnc_tensor_t* dscale = output_size > 1 ? outputs[1] : 0;
	ccv_nnc_tensor_t* dbias = output_size > 2 ? outputs[2] : 0;
	GnGeom g;
	if (!geometry(a, inputs[im], scale, 0, &g) || tensor_count(gr->info) != tensor_count(a->info) || tensor_count(inputs[is]->info) != tensor_count(inputs[im]->info)) return CCV_NNC_EXEC_INVALID;
	const long R = prod(g.rd), total = prod(g.ad);
	if (R == 0 || total == 0) return CCV_NNC_EXEC_SUCCESS;
	const int n = (int)(total / R);
	hipStream_t stream = stream_of(stream_context);
	const float* const gp = (const float*)gr->data.f32;
	const float* const ap = (const float*)a->data.f32;
	const float* const mp = (const float*)inputs[im]->data.f32;
	const float* const ip = (const float*)inputs[is]->data.f32;
	if (h) {
		if (!dense_f32(h) || tensor_count(h->info) != (size_t)total) return CCV_NNC_EXEC_INVALID;
		hipLaunchKernelGGL(gnorm_back_kernel, dim3((unsigned)R), dim3(256), 0, stream, g, gp, ap, scale ? (const float*)scale->data.f32 : (const float*)0, mp, ip, h->data.f32, n, 1.f / (float)n);
		HIP_ENFORCE(hipGetLastError());
	}
	ccv_nnc_tensor_t* const outs[2] = { dscale, dbias };
	for (int w = 0; w < 2; w++) {
		ccv_nnc_tensor_t* const o = outs[w];
		if (!o) continue;
		int pd[4];
		if (!dense_f32(o) || !dims4(o, pd) || !divides(g.ad, pd)) return CCV_NNC_EXEC_INVALID;
		const long P = prod(pd);
		if (P == 0) continue;
		const int pn = (int)(total / P);
		if (pn <= GNORM_SEQ_MAX) { // few terms per element: the reference's summation order (see gnorm_param_grad_seq_kernel)
			if (w == 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(gnorm_param_grad_seq_kernel<true>), dim3((unsigned)((P + 63) / 64)), dim3(64), 0, stream, g, pd[0], pd[1], pd[2], pd[3], gp, ap, mp, ip, o->data.f32, pn, (int)P);
			else hipLaunchKernelGGL(HIP_KERNEL_NAME(gnorm_param_grad_seq_kernel<false>), dim3((unsigned)((P + 63) / 64)), dim3(64), 0, stream, g, pd[0], pd[1], pd[2], pd[3], gp, ap, mp, ip, o->data.f32, pn, (int)P);
		} else if (w == 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(gnorm_param_grad_kernel<true>), dim3((unsigned)P), dim3(256), 0, stream, g, pd[0], pd[1], pd[2], pd[3], gp, ap, mp, ip, o->data.f32, pn);
		else hipLaunchKernelGGL(HIP_KERNEL_NAME(gnorm_param_grad_kernel<false>), dim3((unsigned)P), dim3(256), 0, stream, g, pd[0], pd[1], pd[2], pd[3], gp, ap, mp, ip, o->data.f32, pn);
		HIP_ENFORCE(hipGetLastError());
	}
	return CCV_NNC_EXEC_SUCCESS;
}

} // namespace

#define NNC_REG(CMD, BACKEND, EXEC) \
	extern "C" void _register_command_##CMD##_backend_##BACKEND(ccv_nnc_cmd_backend_registry_t* const registry) \
	{ registry->tensor_formats = CCV_TENSOR_FORMAT_NCHW | CCV_TENSOR_FORMAT_NHWC | CCV_TENSOR_FORMAT_CHWN; registry->tensor_datatypes = CCV_32F; registry->tensor_memory = CCV_TENSOR_GPU_MEMORY; registry->algorithms = 1; registry->exec = EXEC; NNC_HALF_STAGED(registry, EXEC); }

NNC_REG(CCV_NNC_GROUP_NORM_FORWARD, CCV_NNC_BACKEND_GPU_CUDNN, _group_norm_forw)
NNC_REG(CCV_NNC_GROUP_NORM_BACKWARD, CCV_NNC_BACKEND_GPU_CUDNN, _group_norm_back)
