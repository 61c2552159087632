// LAYER_NORM and RMSNORM, forward and backward, on gfx950 -- the transformer-side companions of batch norm.
// Oracle semantics:
//   layer norm  lib/nnc/cmd/norm/ccv_nnc_layer_norm_cpu_ref.c:16-180 (forward), :182-440 (backward)
//               forward (a [, scale, bias]) -> (b, saved_mean, saved_inv_std): mean / centred variance over the axes on which saved_mean
//               has extent 1, inv_std = 1 / sqrt(var + eps), b = (a - mean) inv_std scale + bias
//               backward (g, _, _, a, [scale, _, _,] saved_mean, saved_inv_std) -> (h [, dscale, dbias]):
//               ah = (a - mean) inv_std, gss = g scale inv_std, h = gss - (sum gss + ah sum(ah gss)) / n, dscale = sum_rows ah g, dbias = sum_rows g
//   rms norm    lib/nnc/cmd/norm/ccv_nnc_rmsnorm_cpu_ref.c:16-130, :132-350: no mean, no bias; h = gss - ah sum(ah gss) / n
// Supported geometry (what the transformer layers use): dense tensors, statistics over the TRAILING axes, i.e. [rows][n] with scale / bias of
// n elements (or a single one); other reduce-axis patterns return CCV_NNC_EXEC_INVALID.  One 256-thread block per row; a row is read from
// HBM once and re-read from L1/L2 for the later passes; parameter gradients are row-chunk partial sums folded by colsum_f32 (fixed order).
// HBM-bound: forward 2 |a|, backward 3 |a| bytes.
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

// scale / bias index: per element of the row (stride 1) or one value for all (stride 0)
template <bool CENTER>
__global__ void __launch_bounds__(256) rownorm_forw_kernel(const float* a, const float* scale, const int scale_inc, const float* bias, const int bias_inc, float* b, float* saved_mean, float* saved_inv_std, const int n, const float inv_n, const float epsilon)
{
	__shared__ float red[4];
	const size_t o = (size_t)blockIdx.x * n;
	float mean = 0.f;
	if (CENTER) {
		float s = 0.f;
		for (int j = threadIdx.x; j < n; j += 256) s += a[o + j];
		mean = block_sum(s, red) * inv_n;
	}
	float v = 0.f;
	for (int j = threadIdx.x; j < n; j += 256) { const float w = a[o + j] - mean; v += w * w; }
	const float inv_std = 1.f / sqrtf(block_sum(v, red) * inv_n + epsilon);
	if (threadIdx.x == 0) { if (CENTER) saved_mean[blockIdx.x] = mean; saved_inv_std[blockIdx.x] = inv_std; }
	for (int j = threadIdx.x; j < n; j += 256) {
		float y = (a[o + j] - mean) * inv_std;
		if (scale) y *= scale[j * scale_inc];
		if (bias) y += bias[j * bias_inc];
		b[o + j] = y;
	}
}
template <bool CENTER>
__global__ void __launch_bounds__(256) rownorm_back_kernel(const float* g, const float* a, const float* scale, const int scale_inc, const float* saved_mean, const float* saved_inv_std, float* h, const int n, const float inv_n)
{
	__shared__ float red[4];
	const size_t o = (size_t)blockIdx.x * n;
	const float mean = CENTER ? saved_mean[blockIdx.x] : 0.f, inv_std = saved_inv_std[blockIdx.x];
	float s1 = 0.f, s2 = 0.f;
	for (int j = threadIdx.x; j < n; j += 256) {
		const float ah = (a[o + j] - mean) * inv_std;
		const float gss = g[o + j] * (scale ? scale[j * scale_inc] : 1.f) * inv_std;
		s1 += gss;
		s2 += ah * gss;
	}
	const float gssr = CENTER ? block_sum(s1, red) : 0.f;
	const float ahgssr = block_sum(s2, red);
	for (int j = threadIdx.x; j < n; j += 256) {
		const float ah = (a[o + j] - mean) * inv_std;
		const float gss = g[o + j] * (scale ? scale[j * scale_inc] : 1.f) * inv_std;
		h[o + j] = gss - inv_n * (gssr + ah * ahgssr);
	}
}
// partial[chunk][j] = sum over the chunk's rows of ah * g (the scale gradient before the fold over chunks)
template <bool CENTER>
__global__ void __launch_bounds__(256) rownorm_dscale_partial_kernel(const float* g, const float* a, const float* saved_mean, const float* saved_inv_std, float* partial, const int rows, const int n, const int rows_per_chunk)
{
	const int j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= n) return;
	const int r0 = blockIdx.y * rows_per_chunk;
	const int r1 = r0 + rows_per_chunk < rows ? r0 + rows_per_chunk : rows;
	float s = 0.f;
	for (int r = r0; r < r1; r++) {
		const size_t o = (size_t)r * n + j;
		s += (a[o] - (CENTER ? saved_mean[r] : 0.f)) * saved_inv_std[r] * g[o];
	}
	partial[(size_t)blockIdx.y * n + j] = s;
}

static bool dense_f32(const ccv_nnc_tensor_t* t) { return t && tensor_contiguous(t) && CCV_GET_DATA_TYPE(t->info.datatype) == CCV_32F; }

// rows x n from the statistics tensor: its extents must equal a's on the leading axes and be 1 on the trailing ones
static bool row_geometry(const ccv_nnc_tensor_t* a, const ccv_nnc_tensor_t* stat, int* rows, int* n)
{
	const int nd = tensor_nd(a->info.dim), sd = tensor_nd(stat->info.dim);
	if (nd < 1 || sd > nd) return false;
	long r = 1, m = 1;
	bool trailing = false;
	for (int i = 0; i < nd; i++) {
		const int j = i - (nd - sd);
		const int e = j >= 0 ? stat->info.dim[j] : 1;
		if (e == a->info.dim[i] && !trailing && a->info.dim[i] != 1) r *= e;
		else if (e == 1) { trailing = trailing || a->info.dim[i] != 1; m *= a->info.dim[i]; }
		else return false;
	}
	if (r > 0x7fffffffL || m > 0x7fffffffL || m < 1) return false;
	*rows = (int)r; *n = (int)m;
	return true;
}
static bool param_inc(const ccv_nnc_tensor_t* p, const int n, int* inc)
{
	if (!p) { *inc = 0; return true; }
	if (!dense_f32(p)) return false;
	const size_t c = tensor_count(p->info);
	if (c == (size_t)n) { *inc = 1; return true; }
	if (c == 1) { *inc = 0; return true; }
	return false;
}

#define EXEC_ARGS const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context

template <bool CENTER>
static int rownorm_forw(const ccv_nnc_tensor_t* a, const ccv_nnc_tensor_t* scale, const ccv_nnc_tensor_t* bias, ccv_nnc_tensor_t* b, ccv_nnc_tensor_t* saved_mean, ccv_nnc_tensor_t* saved_inv_std, const float epsilon, ccv_nnc_stream_context_t* const ctx)
{
	if (!dense_f32(a) || !dense_f32(b) || !dense_f32(saved_inv_std) || (CENTER && !dense_f32(saved_mean)) || tensor_count(a->info) != tensor_count(b->info)) return CCV_NNC_EXEC_INVALID;
	int rows, n, sinc, binc;
	if (!row_geometry(a, saved_inv_std, &rows, &n) || !param_inc(scale, n, &sinc) || !param_inc(bias, n, &binc)) return CCV_NNC_EXEC_INVALID;
	if (CENTER && tensor_count(saved_mean->info) != (size_t)rows) return CCV_NNC_EXEC_INVALID;
	if (rows == 0) return CCV_NNC_EXEC_SUCCESS;
	hipLaunchKernelGGL(HIP_KERNEL_NAME(rownorm_forw_kernel<CENTER>), dim3(rows), dim3(256), 0, stream_of(ctx), (const float*)a->data.f32, scale ? (const float*)scale->data.f32 : (const float*)0, sinc,
		bias ? (const float*)bias->data.f32 : (const float*)0, binc, b->data.f32, CENTER ? saved_mean->data.f32 : (float*)0, saved_inv_std->data.f32, n, 1.f / (float)n, epsilon);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}
template <bool CENTER>
static int rownorm_back(const ccv_nnc_tensor_t* g, const ccv_nnc_tensor_t* a, const ccv_nnc_tensor_t* scale, const ccv_nnc_tensor_t* saved_mean, const ccv_nnc_tensor_t* saved_inv_std, ccv_nnc_tensor_t* h, ccv_nnc_tensor_t* dscale, ccv_nnc_tensor_t* dbias, ccv_nnc_stream_context_t* const ctx)
{
	if (!dense_f32(g) || !dense_f32(a) || !dense_f32(saved_inv_std) || (CENTER && !dense_f32(saved_mean)) || tensor_count(g->info) != tensor_count(a->info)) return CCV_NNC_EXEC_INVALID;
	int rows, n, sinc;
	if (!row_geometry(a, saved_inv_std, &rows, &n) || !param_inc(scale, n, &sinc)) return CCV_NNC_EXEC_INVALID;
	if (rows == 0) return CCV_NNC_EXEC_SUCCESS;
	hipStream_t stream = stream_of(ctx);
	const float* const gp = (const float*)g->data.f32;
	const float* const ap = (const float*)a->data.f32;
	const float* const mp = CENTER ? (const float*)saved_mean->data.f32 : (const float*)0;
	const float* const ip = (const float*)saved_inv_std->data.f32;
	if (h) {
		if (!dense_f32(h) || tensor_count(h->info) != tensor_count(a->info)) return CCV_NNC_EXEC_INVALID;
		hipLaunchKernelGGL(HIP_KERNEL_NAME(rownorm_back_kernel<CENTER>), dim3(rows), dim3(256), 0, stream, gp, ap, scale ? (const float*)scale->data.f32 : (const float*)0, sinc, mp, ip, h->data.f32, n, 1.f / (float)n);
		HIP_ENFORCE(hipGetLastError());
	}
	if (dbias) { // sum over rows of g
		if (!dense_f32(dbias) || tensor_count(dbias->info) != (size_t)n) return CCV_NNC_EXEC_INVALID;
		const int r = colsum_f32(gp, rows, n, n, dbias->data.f32, 0, ctx);
		if (r != CCV_NNC_EXEC_SUCCESS) return r;
	}
	if (dscale) {
		if (!dense_f32(dscale) || tensor_count(dscale->info) != (size_t)n) return CCV_NNC_EXEC_INVALID;
		int chunks = (rows + 63) / 64;
		if (chunks > 512) chunks = 512;
		const int rows_per_chunk = (rows + chunks - 1) / chunks;
		chunks = (rows + rows_per_chunk - 1) / rows_per_chunk;
		// [ colsum_f32's own partials (it takes the workspace base) | our chunk partials ]
		const size_t head = (sizeof(float) * (size_t)device_cu_count() * 4 * n + 255) & ~(size_t)255;
		char* ws = (char*)workspace_of(ctx, head + sizeof(float) * (size_t)chunks * n);
		if (!ws) return CCV_NNC_EXEC_OOM;
		float* partial = (float*)(ws + head);
		hipLaunchKernelGGL(HIP_KERNEL_NAME(rownorm_dscale_partial_kernel<CENTER>), dim3((n + 255) / 256, chunks), dim3(256), 0, stream, gp, ap, mp, ip, partial, rows, n, rows_per_chunk);
		HIP_ENFORCE(hipGetLastError());
		const int r = colsum_f32(partial, chunks, n, n, dscale->data.f32, 0, ctx);
		if (r != CCV_NNC_EXEC_SUCCESS) return r;
	}
	return CCV_NNC_EXEC_SUCCESS;
}

static int _layer_norm_forw(EXEC_ARGS)
{
	if (input_size < 1 || output_size < 3 || !inputs[0] || !outputs[0] || !outputs[1] || !outputs[2]) return CCV_NNC_EXEC_INVALID;
	const int affine = cmd.info.lnorm.elementwise_affine;
	const ccv_nnc_tensor_t* scale = affine && input_size >= 2 ? inputs[1] : 0;
	const ccv_nnc_tensor_t* bias = affine && input_size >= 3 ? inputs[2] : 0;
	if (affine && (!scale || !bias)) return CCV_NNC_EXEC_INVALID;
	return rownorm_forw<true>(inputs[0], scale, bias, outputs[0], outputs[1], outputs[2], cmd.info.lnorm.epsilon, stream_context);
}
static int _layer_norm_back(EXEC_ARGS)
{
	const int affine = cmd.info.lnorm.elementwise_affine;
	const int im = affine ? 7 : 5, is = affine ? 8 : 6;
	if (input_size <= is || output_size < 1 || !inputs[0] || !inputs[3] || !inputs[im] || !inputs[is] || (affine && !inputs[4])) return CCV_NNC_EXEC_INVALID;
	return rownorm_back<true>(inputs[0], inputs[3], affine ? inputs[4] : 0, inputs[im], inputs[is], outputs[0], output_size > 1 ? outputs[1] : 0, output_size > 2 ? outputs[2] : 0, stream_context);
}
static int _rmsnorm_forw(EXEC_ARGS)
{
	if (input_size < 2 || output_size < 2 || !inputs[0] || !inputs[1] || !outputs[0] || !outputs[1]) return CCV_NNC_EXEC_INVALID;
	return rownorm_forw<false>(inputs[0], inputs[1], 0, outputs[0], 0, outputs[1], cmd.info.rmsnorm.epsilon, stream_context);
}
static int _rmsnorm_back(EXEC_ARGS)
{ // (g, _, a, scale, _, saved_inv_std) -> (h, dscale)
	if (input_size < 6 || output_size < 1 || !inputs[0] || !inputs[2] || !inputs[3] || !inputs[5]) return CCV_NNC_EXEC_INVALID;
	return rownorm_back<false>(inputs[0], inputs[2], inputs[3], 0, inputs[5], outputs[0], output_size > 1 ? outputs[1] : 0, 0, stream_context);
}

} // namespace

#define NNC_REG(CMD, BACKEND, EXEC) \
	extern "C" void _register_command_##CMD##_backend_##BACKEND(ccv_nnc_cmd_backend_registry_t* const registry) \
	{ registry->tensor_formats = CCV_TENSOR_FORMAT_NCHW | CCV_TENSOR_FORMAT_NHWC | CCV_TENSOR_FORMAT_CHWN; registry->tensor_datatypes = CCV_32F; registry->tensor_memory = CCV_TENSOR_GPU_MEMORY; registry->algorithms = 1; registry->exec = EXEC; NNC_HALF_STAGED(registry, EXEC); }

NNC_REG(CCV_NNC_LAYER_NORM_FORWARD, CCV_NNC_BACKEND_GPU_CUDNN, _layer_norm_forw)
NNC_REG(CCV_NNC_LAYER_NORM_BACKWARD, CCV_NNC_BACKEND_GPU_CUDNN, _layer_norm_back)
NNC_REG(CCV_NNC_RMSNORM_FORWARD, CCV_NNC_BACKEND_GPU_CUDNN, _rmsnorm_forw)
NNC_REG(CCV_NNC_RMSNORM_BACKWARD, CCV_NNC_BACKEND_GPU_CUDNN, _rmsnorm_back)
