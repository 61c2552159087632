// CCV_NNC_SOFTMAX_CROSSENTROPY_FORWARD / BACKWARD on gfx950: one wavefront per row (N x C logits, C = 1000 on the
// hot path), wave-level shuffle reductions, no LDS.  HBM-bound: forward reads a once and writes softmax + loss
// (2|a|), backward reads softmax and writes h (2|a|).
// Oracle: lib/nnc/cmd/softmax_loss/ccv_nnc_softmax_crossentropy_cpu_ref.c:13-181 (forward), :183-360 (backward):
//   loss_i = max_j a_ij - a_i,label    (NOT -log p: the reference stores the un-normalised margin, :55)
//   label forms: fp32 index (rounded +0.5), int32 index, or a dense N x C distribution; label smoothing trim0/trim1
//   softmax_ij = expf(a_ij - max) / sum, the sum accumulated in double (:56-61)
//   backward: h = g_i * (softmax - target); g may be NULL => g_i = 1 (CCV_NNC_CMD_ATTR_NULL_IS_ONES)
// Replaces cudnnSoftmaxForward + 11 small kernels of lib/nnc/cmd/softmax_loss/gpu/ccv_nnc_softmax_crossentropy_gpu_cudnn.cu.
#include "common.h"

using namespace nnc;

namespace {

typedef _Float16 half_t;

enum { LABEL_F32_INDEX = 0, LABEL_I32_INDEX = 1, LABEL_DENSE = 2, LABEL_NONE = 3 };

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) { const float u = __shfl_xor(v, o); v = u > v ? u : v; }
	return v;
}
__device__ __forceinline__ double wave_sum_d(double v)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
	return v;
}
__device__ __forceinline__ float wave_sum_f(float v)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
	return v;
}

// 4 waves per workgroup, one row per wave.  T = float, or half_t: the logits and the softmax stay CCV_16F in their own memory (loads / stores of halves, the arithmetic
// below in fp32 / double as ever, ONE rounding per stored value -- bit for bit what the row computed on fp32 images of the two tensors and converted down: half_stage.cpp
// g_native_half); the loss and the labels are fp32 (or int32) either way.
template <class T>
__global__ void __launch_bounds__(256) softmax_ce_forw_kernel(const T* a, const void* label, const int label_kind, float* loss, T* d, const int rows, const int count, const float trim0, const float trim1)
{
	const int lane = threadIdx.x & 63;
	const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (row >= rows) return; // whole wave exits together
	const T* ap = a + (size_t)row * count;
	T* dp = d + (size_t)row * count;
	float m = -3.402823466e+38f;
	for (int j = lane; j < count; j += 64) { const float v = (float)ap[j]; m = v > m ? v : m; }
	m = wave_max(m);
	if (loss) {
		float p = 0.f;
		if (label_kind == LABEL_DENSE) {
			const float* bp = (const float*)label + (size_t)row * count;
			for (int j = lane; j < count; j += 64) p += bp[j] * (m - (float)ap[j]);
			p = wave_sum_f(p);
		} else {
			const int lb = label_kind == LABEL_F32_INDEX ? (int)(((const float*)label)[row] + 0.5f) : ((const int*)label)[row];
			// a label outside [0, count) is an assert in the reference (softmax_crossentropy_cpu_ref.c); here it must not become a read outside
			// the row: the loss of that row is NaN, which no caller can mistake for a result
			if (trim0 == 0.f && trim1 == 1.f) p = (unsigned)lb < (unsigned)count ? m - (float)ap[lb] : __builtin_nanf("");
			else {
				for (int j = lane; j < count; j += 64) p += (j == lb ? trim1 : trim0) * (m - (float)ap[j]);
				p = wave_sum_f(p);
			}
		}
		if (lane == 0) loss[row] = p;
	}
	double s = 0;
	if (sizeof(T) == sizeof(float)) {
		for (int j = lane; j < count; j += 64) { const float e = expf((float)ap[j] - m); dp[j] = (T)e; s += (double)e; }
		s = wave_sum_d(s);
		const double inv = 1.0 / s;
		for (int j = lane; j < count; j += 64) dp[j] = (T)(float)((double)(float)dp[j] * inv);
	} else { // (a half cannot hold the unnormalised exponential between the passes: it is computed again -- the same value)
		for (int j = lane; j < count; j += 64) s += (double)expf((float)ap[j] - m);
		s = wave_sum_d(s);
		const double inv = 1.0 / s;
		for (int j = lane; j < count; j += 64) dp[j] = (T)(float)((double)expf((float)ap[j] - m) * inv);
	}
}

template <class T>
__global__ void __launch_bounds__(256) softmax_ce_back_kernel(const float* g, const void* label, const int label_kind, const T* d, T* h, const int rows, const int count, const float trim0, const float trim1)
{
	const int lane = threadIdx.x & 63;
	const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (row >= rows) return;
	const T* dp = d + (size_t)row * count;
	T* hp = h + (size_t)row * count;
	if (label_kind == LABEL_DENSE) {
		const float* bp = (const float*)label + (size_t)row * count;
		if (g) { const float gv = g[row]; for (int j = lane; j < count; j += 64) hp[j] = (T)(gv * ((float)dp[j] - bp[j])); }
		else for (int j = lane; j < count; j += 64) hp[j] = (T)((float)dp[j] - bp[j]);
		return;
	}
	const int lb = label_kind == LABEL_F32_INDEX ? (int)(((const float*)label)[row] + 0.5f) : ((const int*)label)[row];
	const bool plain = (trim0 == 0.f && trim1 == 1.f);
	if (g) {
		const float gv = g[row];
		for (int j = lane; j < count; j += 64) {
			float v;
			if (plain) { v = gv * (float)dp[j]; if (j == lb) v -= gv; } // hp[j] = g*d[j]; hp[label] -= g  (:213-215)
			else v = gv * ((float)dp[j] - (j == lb ? trim1 : trim0));
			hp[j] = (T)v;
		}
	} else {
		for (int j = lane; j < count; j += 64) {
			float v = (float)dp[j];
			if (plain) { if (j == lb) v -= 1.f; }
			else v -= (j == lb ? trim1 : trim0);
			hp[j] = (T)v;
		}
	}
}

static int label_kind_of(const ccv_nnc_tensor_t* b, const int batch, const int count)
{
	const int dt = CCV_GET_DATA_TYPE(b->info.datatype);
	if (dt == CCV_32S) return LABEL_I32_INDEX;
	if (dt != CCV_32F) return -1;
	const int nd = tensor_nd(b->info.dim);
	// lib/nnc/ccv_nnc_easy.h ccv_nnc_tensor_get_c: the channel count of a >1-d label tensor
	int range;
	if (nd > 1) range = b->info.format == CCV_TENSOR_FORMAT_NCHW ? (nd == 3 ? b->info.dim[0] : b->info.dim[1]) : b->info.dim[nd - 1];
	else range = (batch == 1 ? b->info.dim[0] : 1);
	if (range == 1) return LABEL_F32_INDEX;
	if (range == count) return LABEL_DENSE;
	return -1;
}

static int _softmax_ce_forw(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	if (input_size < 2 || output_size < 2 || !inputs[0] || !outputs[1]) return CCV_NNC_EXEC_INVALID;
	const ccv_nnc_tensor_t* a = inputs[0];
	const ccv_nnc_tensor_t* b = inputs[1];
	ccv_nnc_tensor_t* c = outputs[0];
	ccv_nnc_tensor_t* d = outputs[1];
	if (!tensor_contiguous(a) || (b && !tensor_contiguous(b)) || !tensor_contiguous(d) || (c && !tensor_contiguous(c))) return CCV_NNC_EXEC_INVALID;
	const int nd = tensor_nd(a->info.dim);
	const int batch = nd < 2 ? 1 : a->info.dim[0];
	const int count = (int)(tensor_count(a->info) / batch);
	if (tensor_count(d->info) != tensor_count(a->info)) return CCV_NNC_EXEC_INVALID;
	int kind = LABEL_NONE;
	if (c) {
		if (!b || (int)tensor_count(c->info) != batch) return CCV_NNC_EXEC_INVALID;
		kind = label_kind_of(b, batch, count);
		if (kind < 0) return CCV_NNC_EXEC_INVALID;
	}
	if (batch == 0 || count == 0) return CCV_NNC_EXEC_SUCCESS;
	const bool half = CCV_GET_DATA_TYPE(a->info.datatype) == CCV_16F; // (half_stage.cpp hands logits and softmax over as they are when BOTH are dense CCV_16F tensors)
	if (half != (CCV_GET_DATA_TYPE(d->info.datatype) == CCV_16F) || (c && CCV_GET_DATA_TYPE(c->info.datatype) != CCV_32F)) return CCV_NNC_EXEC_INVALID;
	if (half) hipLaunchKernelGGL(HIP_KERNEL_NAME(softmax_ce_forw_kernel<half_t>), dim3((batch + 3) / 4), dim3(256), 0, stream_of(stream_context), (const half_t*)a->data.u8, (const void*)(b ? b->data.ptr : 0), kind, c ? c->data.f32 : (float*)0, (half_t*)d->data.u8, batch, count, cmd.info.label_smoothing.trim0, cmd.info.label_smoothing.trim1);
	else hipLaunchKernelGGL(HIP_KERNEL_NAME(softmax_ce_forw_kernel<float>), dim3((batch + 3) / 4), dim3(256), 0, stream_of(stream_context), (const float*)a->data.f32, (const void*)(b ? b->data.ptr : 0), kind, c ? c->data.f32 : (float*)0, d->data.f32, batch, count, cmd.info.label_smoothing.trim0, cmd.info.label_smoothing.trim1);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}

static int _softmax_ce_back(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	// inputs: [0] g (grad of loss, may be null), [3] label, [5] softmax; output [0] h   (cpu_ref.c:185-192)
	if (input_size < 6 || output_size < 1 || !inputs[3] || !inputs[5] || !outputs[0]) return CCV_NNC_EXEC_INVALID;
	const ccv_nnc_tensor_t* g = inputs[0];
	const ccv_nnc_tensor_t* b = inputs[3];
	const ccv_nnc_tensor_t* d = inputs[5];
	ccv_nnc_tensor_t* h = outputs[0];
	if ((g && !tensor_contiguous(g)) || !tensor_contiguous(b) || !tensor_contiguous(d) || !tensor_contiguous(h)) return CCV_NNC_EXEC_INVALID;
	const int nd = tensor_nd(d->info.dim);
	const int batch = nd < 2 ? 1 : d->info.dim[0];
	const int count = (int)(tensor_count(d->info) / batch);
	if (tensor_count(h->info) != tensor_count(d->info) || (g && (int)tensor_count(g->info) != batch)) return CCV_NNC_EXEC_INVALID;
	const int kind = label_kind_of(b, batch, count);
	if (kind < 0) return CCV_NNC_EXEC_INVALID;
	if (batch == 0 || count == 0) return CCV_NNC_EXEC_SUCCESS;
	const bool half = CCV_GET_DATA_TYPE(d->info.datatype) == CCV_16F;
	if (half != (CCV_GET_DATA_TYPE(h->info.datatype) == CCV_16F) || (g && CCV_GET_DATA_TYPE(g->info.datatype) != CCV_32F)) return CCV_NNC_EXEC_INVALID;
	if (half) hipLaunchKernelGGL(HIP_KERNEL_NAME(softmax_ce_back_kernel<half_t>), dim3((batch + 3) / 4), dim3(256), 0, stream_of(stream_context), g ? (const float*)g->data.f32 : (const float*)0, (const void*)b->data.ptr, kind, (const half_t*)d->data.u8, (half_t*)h->data.u8, batch, count, cmd.info.label_smoothing.trim0, cmd.info.label_smoothing.trim1);
	else hipLaunchKernelGGL(HIP_KERNEL_NAME(softmax_ce_back_kernel<float>), dim3((batch + 3) / 4), dim3(256), 0, stream_of(stream_context), g ? (const float*)g->data.f32 : (const float*)0, (const void*)b->data.ptr, kind, (const float*)d->data.f32, h->data.f32, batch, count, cmd.info.label_smoothing.trim0, cmd.info.label_smoothing.trim1);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}

} // namespace

#define NNC_REG(CMD, BACKEND, EXEC) \
	extern "C" void _register_command_##CMD##_backend_##BACKEND(ccv_nnc_cmd_backend_registry_t* const registry) \
	{ registry->tensor_formats = CCV_TENSOR_FORMAT_NCHW | CCV_TENSOR_FORMAT_NHWC; registry->tensor_datatypes = CCV_32F | CCV_32S; registry->tensor_memory = CCV_TENSOR_GPU_MEMORY; registry->algorithms = 1; registry->exec = EXEC; NNC_HALF_STAGED(registry, EXEC); }
NNC_REG(CCV_NNC_SOFTMAX_CROSSENTROPY_FORWARD, CCV_NNC_BACKEND_GPU_CUDNN, _softmax_ce_forw)
NNC_REG(CCV_NNC_SOFTMAX_CROSSENTROPY_BACKWARD, CCV_NNC_BACKEND_GPU_CUDNN, _softmax_ce_back)
