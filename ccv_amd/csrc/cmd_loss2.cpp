// Row losses on gfx950: MSE, smooth-L1 (Huber on the row's L1 sum), binary and categorical cross-entropy, forward and backward
// (SURVEY.md section 8(f).1).  Tensors are [batch][count] (1-d = one row); the loss / incoming gradient is one scalar per row.
// Forward: one 256-thread block per row, wave shuffles + LDS fold (fixed order); backward: element-wise, 4 bytes per lane (the
// row scalar is read through L1).  HBM-bound: forward 2 |a| bytes, backward 3 |a|.
// Oracle semantics:
//   mse                       lib/nnc/cmd/loss/ccv_nnc_mse_cpu_ref.c:13-171                  (reduce_op mean / sum)
//   smooth l1                 lib/nnc/cmd/loss/ccv_nnc_smooth_l1_cpu_ref.c:13-148            backward inputs (g, a, b, c)
//   binary cross-entropy      lib/nnc/cmd/loss/ccv_nnc_binary_crossentropy_cpu_ref.c:13-161   pos_weight
//   categorical cross-entropy lib/nnc/cmd/loss/ccv_nnc_categorical_crossentropy_cpu_ref.c:13-301  labels: fp32 index (+0.5 rounding), int32 index,
//                                                                                             or a dense distribution; label smoothing trim0 / trim1
#include "common.h"

using namespace nnc;

namespace {

enum { MSE_REDUCE_MEAN = 0, MSE_REDUCE_SUM = 1 }; // CCV_NNC_MSE_REDUCE_* (ccv_nnc.h)

__device__ __forceinline__ float block_sum(float v, float* red)
{
	for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
	__syncthreads();
	if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
	__syncthreads();
	return red[0] + red[1] + red[2] + red[3];
}

struct RowGeom { int count; };

// c[row] = F::finish(sum1, sum2) with (sum1, sum2) = sum_j F::term(a[j], b[j])
template <class F>
__global__ void __launch_bounds__(256) row_loss_forw_kernel(const F f, const float* a, const float* b, float* c, const int count)
{
	__shared__ float red[4];
	const size_t o = (size_t)blockIdx.x * count;
	float s1 = 0.f, s2 = 0.f;
	for (int j = threadIdx.x; j < count; j += 256) f.term(a[o + j], b[o + j], s1, s2);
	s1 = block_sum(s1, red);
	s2 = block_sum(s2, red);
	if (threadIdx.x == 0) c[blockIdx.x] = f.finish(s1, s2);
}
// h[row][j] = F::grad(a, b, g[row] or 1, c[row] if given)
template <class F>
__global__ void __launch_bounds__(256) row_loss_back_kernel(const F f, const float* g, const float* a, const float* b, const float* c, float* h, const int count, const size_t n)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
		const size_t row = i / count;
		h[i] = f.grad(a[i], b[i], g ? g[row] : 1.f, c ? c[row] : 0.f);
	}
}

struct Mse {
	float scale, grad_scale, sign;
	__device__ void term(float a, float b, float& s1, float&) const { const float d = b - a; s1 += d * d; }
	__device__ float finish(float s1, float) const { return s1 * scale; }
	__device__ float grad(float a, float b, float g, float) const { return (grad_scale * g) * (sign * (a - b)); }
};
struct SmoothL1 {
	float beta, beta_inv_2, beta_2, inv_beta;
	__device__ void term(float a, float b, float& s1, float& s2) const { const float d = b - a; s1 += fabsf(d); s2 += d * d; }
	__device__ float finish(float s1, float s2) const { return s1 < beta ? s2 * beta_inv_2 : s1 - beta_2; }
	__device__ float grad(float a, float b, float g, float c) const { return c < beta_2 ? (inv_beta * g) * (a - b) : ((a - b) > 0 ? 1.f : -1.f) * g; }
};
struct Bce {
	float pos_weight, pos_weight_1;
	__device__ void term(float a, float b, float& s1, float& s2) const { s1 += (b - 1.f) * logf(1.f - a); s2 += b * logf(a); }
	__device__ float finish(float s1, float s2) const { return s1 - s2 * pos_weight; }
	__device__ float grad(float a, float b, float g, float) const { return g * (a * b * pos_weight_1 + a - pos_weight * b) / fmaxf((1.f - a) * a, 1e-12f); }
};

static int rows_of(const ccv_nnc_tensor_t* a, int* batch, int* count)
{
	const int nd = tensor_nd(a->info.dim);
	if (nd < 1) return 0;
	*batch = nd < 2 ? 1 : a->info.dim[0];
	const size_t n = tensor_count(a->info);
	*count = *batch > 0 ? (int)(n / *batch) : 0;
	return 1;
}
static bool dense_f32(const ccv_nnc_tensor_t* t) { return t && tensor_contiguous(t) && CCV_GET_DATA_TYPE(t->info.datatype) == CCV_32F; }

template <class F>
static int loss_forw(const F f, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const ctx)
{
	if (input_size < 2 || output_size < 1 || !dense_f32(inputs[0]) || !dense_f32(inputs[1]) || !dense_f32(outputs[0])) return CCV_NNC_EXEC_INVALID;
	int batch, count;
	if (!rows_of(inputs[0], &batch, &count) || tensor_count(inputs[1]->info) != tensor_count(inputs[0]->info) || tensor_count(outputs[0]->info) != (size_t)batch) return CCV_NNC_EXEC_INVALID;
	if (batch == 0) return CCV_NNC_EXEC_SUCCESS;
	hipLaunchKernelGGL(HIP_KERNEL_NAME(row_loss_forw_kernel<F>), dim3(batch), dim3(256), 0, stream_of(ctx), f, (const float*)inputs[0]->data.f32, (const float*)inputs[1]->data.f32, outputs[0]->data.f32, count);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}
template <class F>
static int loss_back(const F f, const ccv_nnc_tensor_t* g, const ccv_nnc_tensor_t* a, const ccv_nnc_tensor_t* b, const ccv_nnc_tensor_t* c, ccv_nnc_tensor_t* h, ccv_nnc_stream_context_t* const ctx)
{
	if (!h) return CCV_NNC_EXEC_SUCCESS;
	if (!dense_f32(a) || !dense_f32(b) || !dense_f32(h) || (g && !dense_f32(g)) || (c && !dense_f32(c))) return CCV_NNC_EXEC_INVALID;
	int batch, count;
	if (!rows_of(a, &batch, &count)) return CCV_NNC_EXEC_INVALID;
	const size_t n = tensor_count(a->info);
	if (tensor_count(b->info) != n || tensor_count(h->info) != n || (g && tensor_count(g->info) != (size_t)batch) || (c && tensor_count(c->info) != (size_t)batch)) return CCV_NNC_EXEC_INVALID;
	if (n == 0) return CCV_NNC_EXEC_SUCCESS;
	hipLaunchKernelGGL(HIP_KERNEL_NAME(row_loss_back_kernel<F>), dim3(grid_for(n, 256)), dim3(256), 0, stream_of(ctx), f, g ? (const float*)g->data.f32 : (const float*)0, (const float*)a->data.f32, (const float*)b->data.f32,
		c ? (const float*)c->data.f32 : (const float*)0, h->data.f32, count, n);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}

#define EXEC_ARGS const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context

static Mse mse_of(const ccv_nnc_cmd_t& cmd, const ccv_nnc_tensor_t* a, const float sign)
{
	int batch = 1, count = 1;
	rows_of(a, &batch, &count);
	const bool mean = cmd.info.mse.reduce_op == MSE_REDUCE_MEAN;
	Mse f = { mean ? 1.f / (float)count : 1.f, mean ? 2.f / (float)count : 2.f, sign };
	return f;
}
static int _mse_forw(EXEC_ARGS)
{
	if (input_size < 1 || !inputs[0]) return CCV_NNC_EXEC_INVALID;
	return loss_forw(mse_of(cmd, inputs[0], 1.f), inputs, input_size, outputs, output_size, stream_context);
}
static int _mse_back(EXEC_ARGS)
{ // (g, a, b) -> (ha, hb)
	if (input_size < 3 || output_size < 1 || !inputs[1] || !inputs[2]) return CCV_NNC_EXEC_INVALID;
	int r = loss_back(mse_of(cmd, inputs[1], 1.f), inputs[0], inputs[1], inputs[2], 0, outputs[0], stream_context);
	if (r != CCV_NNC_EXEC_SUCCESS) return r;
	return output_size >= 2 ? loss_back(mse_of(cmd, inputs[1], -1.f), inputs[0], inputs[1], inputs[2], 0, outputs[1], stream_context) : r;
}
static SmoothL1 sl1_of(const ccv_nnc_cmd_t& cmd) { const float beta = cmd.info.smooth_l1.beta; SmoothL1 f = { beta, 0.5f / beta, 0.5f * beta, 1.f / beta }; return f; }
static int _sl1_forw(EXEC_ARGS) { return loss_forw(sl1_of(cmd), inputs, input_size, outputs, output_size, stream_context); }
static int _sl1_back(EXEC_ARGS)
{ // (g, a, b, c) -> h
	if (input_size < 4 || output_size < 1 || !inputs[1] || !inputs[2] || !inputs[3]) return CCV_NNC_EXEC_INVALID;
	return loss_back(sl1_of(cmd), inputs[0], inputs[1], inputs[2], inputs[3], outputs[0], stream_context);
}
static Bce bce_of(const ccv_nnc_cmd_t& cmd) { Bce f = { cmd.info.binary_crossentropy.pos_weight, cmd.info.binary_crossentropy.pos_weight - 1.f }; return f; }
static int _bce_forw(EXEC_ARGS) { return loss_forw(bce_of(cmd), inputs, input_size, outputs, output_size, stream_context); }
static int _bce_back(EXEC_ARGS)
{ // (g, a, b) -> h
	if (input_size < 3 || output_size < 1 || !inputs[1] || !inputs[2]) return CCV_NNC_EXEC_INVALID;
	return loss_back(bce_of(cmd), inputs[0], inputs[1], inputs[2], 0, outputs[0], stream_context);
}

// ---- sigmoid + binary cross-entropy in one (lib/nnc/cmd/sigmoid_loss/ccv_nnc_sigmoid_binary_crossentropy_cpu_ref.c:13-150) --------
// forward (a, b) -> (c [optional], d = sigmoid a): c = sum (1 - b) a + (1 + b (pw - 1)) log(1 + e^-a)
// backward inputs (g, _, _, b, _, d) -> h = g ((d - 1) b pw + d (1 - b))
__global__ void __launch_bounds__(256) sbce_forw_kernel(const float* a, const float* b, float* c, float* d, const int count, const float pos_weight_1)
{
	__shared__ float red[4];
	const size_t o = (size_t)blockIdx.x * count;
	float s = 0.f;
	for (int j = threadIdx.x; j < count; j += 256) {
		const float av = a[o + j];
		const float e = expf(-av);
		d[o + j] = 1.f / (1.f + e);
		if (c) { const float bv = b[o + j]; s += (1.f - bv) * av + (1.f + bv * pos_weight_1) * logf(1.f + e); }
	}
	if (c) { s = block_sum(s, red); if (threadIdx.x == 0) c[blockIdx.x] = s; }
}
struct SbceBack { // (a := d, b) per element
	float pos_weight;
	__device__ float grad(float d, float b, float g, float) const { return g * ((d - 1.f) * b * pos_weight + d * (1.f - b)); }
};
static int _sbce_forw(EXEC_ARGS)
{
	if (input_size < 2 || output_size < 2 || !dense_f32(inputs[0]) || !dense_f32(outputs[1]) || (outputs[0] && (!dense_f32(outputs[0]) || !dense_f32(inputs[1])))) return CCV_NNC_EXEC_INVALID;
	int batch, count;
	const size_t n = tensor_count(inputs[0]->info);
	if (!rows_of(inputs[0], &batch, &count) || tensor_count(outputs[1]->info) != n) return CCV_NNC_EXEC_INVALID;
	if (outputs[0] && (tensor_count(outputs[0]->info) != (size_t)batch || tensor_count(inputs[1]->info) != n)) return CCV_NNC_EXEC_INVALID;
	if (n == 0) return CCV_NNC_EXEC_SUCCESS;
	hipLaunchKernelGGL(sbce_forw_kernel, dim3(batch), dim3(256), 0, stream_of(stream_context), (const float*)inputs[0]->data.f32, outputs[0] ? (const float*)inputs[1]->data.f32 : (const float*)0,
		outputs[0] ? outputs[0]->data.f32 : (float*)0, outputs[1]->data.f32, count, cmd.info.binary_crossentropy.pos_weight - 1.f);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}
static int _sbce_back(EXEC_ARGS)
{
	if (input_size < 6 || output_size < 1 || !inputs[3] || !inputs[5]) return CCV_NNC_EXEC_INVALID;
	SbceBack f = { cmd.info.binary_crossentropy.pos_weight };
	return loss_back(f, inputs[0], inputs[5], inputs[3], 0, outputs[0], stream_context);
}

// ---- categorical cross-entropy: labels as in softmax-cross-entropy (cmd_loss.cpp) -----------------------------------------------
enum { LABEL_F32_INDEX = 0, LABEL_I32_INDEX = 1, LABEL_DENSE = 2 };
__global__ void __launch_bounds__(256) cce_forw_kernel(const float* a, const void* b, const int kind, float* c, const int count, const float trim0, const float trim1)
{
	__shared__ float red[4];
	const size_t o = (size_t)blockIdx.x * count;
	if (kind != LABEL_DENSE) {
		const int label = kind == LABEL_F32_INDEX ? (int)(((const float*)b)[blockIdx.x] + 0.5f) : ((const int*)b)[blockIdx.x];
		if (trim0 == 0.f && trim1 == 1.f) {
			if (threadIdx.x == 0) c[blockIdx.x] = -logf(a[o + label]);
			return;
		}
		float s = 0.f;
		for (int j = threadIdx.x; j < count; j += 256) s += -(j == label ? trim1 : trim0) * logf(a[o + j]);
		s = block_sum(s, red);
		if (threadIdx.x == 0) c[blockIdx.x] = s;
	} else {
		const float* const bp = (const float*)b + o;
		float s = 0.f;
		for (int j = threadIdx.x; j < count; j += 256) s += -bp[j] * logf(a[o + j]);
		s = block_sum(s, red);
		if (threadIdx.x == 0) c[blockIdx.x] = s;
	}
}
__global__ void __launch_bounds__(256) cce_back_kernel(const float* g, const float* a, const void* b, const int kind, float* h, const int count, const size_t n, const float trim0, const float trim1)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
		const size_t row = i / count;
		const int j = (int)(i - row * count);
		const float gp = g ? g[row] : 1.f;
		float t;
		if (kind == LABEL_DENSE) t = ((const float*)b)[i];
		else {
			const int label = kind == LABEL_F32_INDEX ? (int)(((const float*)b)[row] + 0.5f) : ((const int*)b)[row];
			t = j == label ? trim1 : trim0;
		}
		h[i] = t == 0.f ? 0.f : -gp * t / a[i];
	}
}
static int label_kind(const ccv_nnc_tensor_t* a, const ccv_nnc_tensor_t* b, const int batch, int* kind)
{
	const int dt = CCV_GET_DATA_TYPE(b->info.datatype);
	if (dt == CCV_32S) { *kind = LABEL_I32_INDEX; return tensor_count(b->info) == (size_t)batch; }
	if (dt != CCV_32F) return 0;
	// categorical_crossentropy_cpu_ref.c:27-29: more than one axis => channel count; one axis => the whole thing if batch == 1, else indices
	const int nd = tensor_nd(b->info.dim);
	const int range = nd > 1 ? b->info.dim[nd - 1] : (batch == 1 ? b->info.dim[0] : 1);
	if (range == 1) { *kind = LABEL_F32_INDEX; return tensor_count(b->info) == (size_t)batch; }
	*kind = LABEL_DENSE;
	return tensor_count(b->info) == tensor_count(a->info);
}
static int _cce_forw(EXEC_ARGS)
{
	if (input_size < 2 || output_size < 1 || !dense_f32(inputs[0]) || !inputs[1] || !tensor_contiguous(inputs[1]) || !dense_f32(outputs[0])) return CCV_NNC_EXEC_INVALID;
	int batch, count, kind;
	if (!rows_of(inputs[0], &batch, &count) || !label_kind(inputs[0], inputs[1], batch, &kind) || tensor_count(outputs[0]->info) != (size_t)batch) return CCV_NNC_EXEC_INVALID;
	if (batch == 0) return CCV_NNC_EXEC_SUCCESS;
	hipLaunchKernelGGL(cce_forw_kernel, dim3(batch), dim3(256), 0, stream_of(stream_context), (const float*)inputs[0]->data.f32, (const void*)inputs[1]->data.u8, kind, outputs[0]->data.f32, count, cmd.info.label_smoothing.trim0, cmd.info.label_smoothing.trim1);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}
static int _cce_back(EXEC_ARGS)
{ // (g, a, b) -> h
	if (input_size < 3 || output_size < 1 || !dense_f32(inputs[1]) || !inputs[2] || !tensor_contiguous(inputs[2]) || !dense_f32(outputs[0]) || (inputs[0] && !dense_f32(inputs[0]))) return CCV_NNC_EXEC_INVALID;
	int batch, count, kind;
	const size_t n = tensor_count(inputs[1]->info);
	if (!rows_of(inputs[1], &batch, &count) || !label_kind(inputs[1], inputs[2], batch, &kind) || tensor_count(outputs[0]->info) != n || (inputs[0] && tensor_count(inputs[0]->info) != (size_t)batch)) return CCV_NNC_EXEC_INVALID;
	if (n == 0) return CCV_NNC_EXEC_SUCCESS;
	hipLaunchKernelGGL(cce_back_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream_of(stream_context), inputs[0] ? (const float*)inputs[0]->data.f32 : (const float*)0, (const float*)inputs[1]->data.f32, (const void*)inputs[2]->data.u8, kind,
		outputs[0]->data.f32, count, n, cmd.info.label_smoothing.trim0, cmd.info.label_smoothing.trim1);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}

} // namespace

#define NNC_REG(CMD, BACKEND, DATATYPES, EXEC) \
	extern "C" void _register_command_##CMD##_backend_##BACKEND(ccv_nnc_cmd_backend_registry_t* const registry) \
	{ registry->tensor_formats = CCV_TENSOR_FORMAT_NCHW | CCV_TENSOR_FORMAT_NHWC | CCV_TENSOR_FORMAT_CHWN; registry->tensor_datatypes = (DATATYPES); registry->tensor_memory = CCV_TENSOR_GPU_MEMORY; registry->algorithms = 1; registry->exec = EXEC; NNC_HALF_STAGED(registry, EXEC); }

NNC_REG(CCV_NNC_MSE_FORWARD, CCV_NNC_BACKEND_GPU_REF, CCV_32F, _mse_forw)
NNC_REG(CCV_NNC_MSE_BACKWARD, CCV_NNC_BACKEND_GPU_REF, CCV_32F, _mse_back)
NNC_REG(CCV_NNC_SMOOTH_L1_FORWARD, CCV_NNC_BACKEND_GPU_REF, CCV_32F, _sl1_forw)
NNC_REG(CCV_NNC_SMOOTH_L1_BACKWARD, CCV_NNC_BACKEND_GPU_REF, CCV_32F, _sl1_back)
NNC_REG(CCV_NNC_BINARY_CROSSENTROPY_FORWARD, CCV_NNC_BACKEND_GPU_REF, CCV_32F, _bce_forw)
NNC_REG(CCV_NNC_BINARY_CROSSENTROPY_BACKWARD, CCV_NNC_BACKEND_GPU_REF, CCV_32F, _bce_back)
NNC_REG(CCV_NNC_SIGMOID_BINARY_CROSSENTROPY_FORWARD, CCV_NNC_BACKEND_GPU_REF, CCV_32F, _sbce_forw)
NNC_REG(CCV_NNC_SIGMOID_BINARY_CROSSENTROPY_BACKWARD, CCV_NNC_BACKEND_GPU_REF, CCV_32F, _sbce_back)
NNC_REG(CCV_NNC_CATEGORICAL_CROSSENTROPY_FORWARD, CCV_NNC_BACKEND_GPU_REF, CCV_32F | CCV_32S, _cce_forw)
NNC_REG(CCV_NNC_CATEGORICAL_CROSSENTROPY_BACKWARD, CCV_NNC_BACKEND_GPU_REF, CCV_32F | CCV_32S, _cce_back)
