// Activations, plain softmax and the Adam / AdamW / RMSProp update on gfx950 -- the element-wise rows the reference's other
// trainers need beside SGD.  All HBM-bound: grid-stride, 16 bytes per lane when aligned.
// Oracle semantics (CPU reference, fp32 storage, arithmetic promoted to double there; float here, within 1e-6):
//   sigmoid     lib/nnc/cmd/sigmoid/ccv_nnc_sigmoid_cpu_ref.c:13-66        b = 1/(1+e^-a);  h = g b (1-b)      (g may be absent: ones)
//   tanh        lib/nnc/cmd/tanh/ccv_nnc_tanh_cpu_ref.c:13-62              b = tanh a;      h = g (1-b^2)
//   gelu        lib/nnc/cmd/gelu/ccv_nnc_gelu_cpu_ref.c:13-91              erf form and the tanh approximation (cmd.info.gelu.tanh)
//   swish       lib/nnc/cmd/swish/ccv_nnc_swish_cpu_ref.c:13-79            b = a/(1+e^-a);  h = g (a (y - y^2) + y), y = sigmoid a
//   leaky relu  lib/nnc/cmd/leaky_relu/ccv_nnc_leaky_relu_cpu_ref.c:13-60  b = a >= 0 ? a : s a;  h = b >= 0 ? g : s g
//   softmax     lib/nnc/cmd/softmax/ccv_nnc_softmax_cpu_ref.c:13-73        rows = dim[0] (1-d: one row); h = (g - sum(g b)) b
//   adam        lib/nnc/cmd/adam/ccv_nnc_adam_cpu_ref.c:16-175             inputs (g, a, m, v[, vm]) -> (b, n, u[, um]); L2 decay inside the gradient
//   adamw       lib/nnc/cmd/adam/ccv_nnc_adamw_cpu_ref.c:16-174            decoupled decay: b = a - rate decay a - ...
//   rmsprop     lib/nnc/cmd/rmsprop/ccv_nnc_rmsprop_cpu_ref.c:16-108       inputs (g, a, m, v) -> (b, n, u)
//   lamb        lib/nnc/cmd/lamb/ccv_nnc_lamb_cpu_ref.c:16-140             Adam-style update scaled per TENSOR by |w| / |update| (norms in double)
#include "common.h"

using namespace nnc;

namespace {

constexpr int EW_THREADS = 256;

// out[i] = f(x[i], y[i]) over contiguous fp32 tensors; NIN = how many inputs are read
template <class F, int NIN>
__global__ void __launch_bounds__(EW_THREADS) act_map_kernel(F f, float* out, const float* in0, const float* in1, const size_t n4, const size_t n)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	for (size_t i = tid; i < n4; i += stride) {
		const float4 a = ((const float4*)in0)[i];
		const float4 b = NIN > 1 ? ((const float4*)in1)[i] : make_float4(0, 0, 0, 0);
		((float4*)out)[i] = make_float4(f(a.x, b.x), f(a.y, b.y), f(a.z, b.z), f(a.w, b.w));
	}
	for (size_t i = n4 * 4 + tid; i < n; i += stride) out[i] = f(in0[i], NIN > 1 ? in1[i] : 0.f);
}

template <class F, int NIN>
static int act_map(F f, float* out, const float* in0, const float* in1, const size_t n, ccv_nnc_stream_context_t* ctx)
{
	if (n == 0) return CCV_NNC_EXEC_SUCCESS;
	const bool vec = aligned16(out) && aligned16(in0) && (NIN < 2 || aligned16(in1));
	const size_t n4 = vec ? n / 4 : 0;
	hipLaunchKernelGGL(HIP_KERNEL_NAME(act_map_kernel<F, NIN>), dim3(grid_for(vec ? n4 + 3 : n, EW_THREADS)), dim3(EW_THREADS), 0, stream_of(ctx), f, out, in0, in1, n4, n);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}

__device__ __forceinline__ float sigmoidf_(const float x) { return 1.f / (1.f + expf(-x)); }

struct OpSigmoid { __device__ float operator()(float a, float) const { return sigmoidf_(a); } };
struct OpSigmoidBack { __device__ float operator()(float b, float g) const { return g * b * (1.f - b); } };      // (b, g)
struct OpSigmoidBackOnes { __device__ float operator()(float b, float) const { return b * (1.f - b); } };
struct OpTanh { __device__ float operator()(float a, float) const { return tanhf(a); } };
struct OpTanhBack { __device__ float operator()(float b, float g) const { return g * (1.f - b * b); } };
struct OpTanhBackOnes { __device__ float operator()(float b, float) const { return 1.f - b * b; } };
struct OpGeluErf { __device__ float operator()(float x, float) const { return x * 0.5f * (1.f + erff(x * 0.70710678118654752440f)); } };
struct OpGeluTanh { __device__ float operator()(float x, float) const { return 0.5f * x * (1.f + tanhf(0.797884560802865355f * (x + 0.044715f * x * x * x))); } };
struct OpGeluErfBack { // (x, g)
	__device__ float operator()(float x, float g) const
	{
		const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752440f));
		const float pdf = expf(-0.5f * x * x) * 0.797884560802865355f;
		return g * (cdf + x * pdf);
	}
};
struct OpGeluTanhBack {
	__device__ float operator()(float x, float g) const
	{
		const float x_sq = x * x;
		const float t = tanhf(0.797884560802865355f * (x + 0.044715f * x_sq * x));
		const float left_d = 0.5f * (1.f + t);
		const float right_d = 0.5f * x * (1.f - t * t) * 0.797884560802865355f * (1.f + 3.f * 0.044715f * x_sq);
		return g * (left_d + right_d);
	}
};
struct OpSwish { __device__ float operator()(float a, float) const { return a * sigmoidf_(a); } };
struct OpSwishBack { __device__ float operator()(float x, float g) const { const float y = sigmoidf_(x); return g * (x * (y - y * y) + y); } };
struct OpLeaky { float s; __device__ float operator()(float a, float) const { return a >= 0.f ? a : a * s; } };
struct OpLeakyBack { float s; __device__ float operator()(float b, float g) const { return b >= 0.f ? g : s * g; } }; // (b, g)

static bool same_count(const ccv_nnc_tensor_t* a, const ccv_nnc_tensor_t* b) { return tensor_count(a->info) == tensor_count(b->info); }
static bool dense_f32(const ccv_nnc_tensor_t* t) { return t && tensor_contiguous(t) && CCV_GET_DATA_TYPE(t->info.datatype) == CCV_32F; }

// forward: inputs[0] = a -> outputs[0] = b
template <class F>
static int unary_forw(F f, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const ctx)
{
	if (input_size < 1 || output_size < 1 || !dense_f32(inputs[0]) || !dense_f32(outputs[0]) || !same_count(inputs[0], outputs[0])) return CCV_NNC_EXEC_INVALID;
	return act_map<F, 1>(f, outputs[0]->data.f32, inputs[0]->data.f32, 0, tensor_count(inputs[0]->info), ctx);
}
// backward from the forward OUTPUT: inputs (g [may be null], _, b) -> h
template <class F, class FONES>
static int back_from_output(F f, FONES fones, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const ctx)
{
	if (input_size < 3 || output_size < 1 || !dense_f32(inputs[2]) || !dense_f32(outputs[0]) || !same_count(inputs[2], outputs[0])) return CCV_NNC_EXEC_INVALID;
	const ccv_nnc_tensor_t* g = inputs[0];
	const size_t n = tensor_count(inputs[2]->info);
	if (!g) return act_map<FONES, 1>(fones, outputs[0]->data.f32, inputs[2]->data.f32, 0, n, ctx);
	if (!dense_f32(g) || !same_count(g, outputs[0])) return CCV_NNC_EXEC_INVALID;
	return act_map<F, 2>(f, outputs[0]->data.f32, inputs[2]->data.f32, g->data.f32, n, ctx);
}
// backward from the forward INPUT: inputs (g, a) -> h
template <class F>
static int back_from_input(F f, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const ctx)
{
	if (input_size < 2 || output_size < 1 || !dense_f32(inputs[0]) || !dense_f32(inputs[1]) || !dense_f32(outputs[0]) || !same_count(inputs[0], inputs[1]) || !same_count(inputs[0], outputs[0])) return CCV_NNC_EXEC_INVALID;
	return act_map<F, 2>(f, outputs[0]->data.f32, inputs[1]->data.f32, inputs[0]->data.f32, tensor_count(inputs[0]->info), ctx);
}

#define EXEC_ARGS const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context
#define IO inputs, input_size, outputs, output_size, stream_context

static int _sigmoid_forw(EXEC_ARGS) { return unary_forw(OpSigmoid(), IO); }
static int _sigmoid_back(EXEC_ARGS) { return back_from_output(OpSigmoidBack(), OpSigmoidBackOnes(), IO); }
static int _tanh_forw(EXEC_ARGS) { return unary_forw(OpTanh(), IO); }
static int _tanh_back(EXEC_ARGS) { return back_from_output(OpTanhBack(), OpTanhBackOnes(), IO); }
static int _gelu_forw(EXEC_ARGS) { return cmd.info.gelu.tanh ? unary_forw(OpGeluTanh(), IO) : unary_forw(OpGeluErf(), IO); }
static int _gelu_back(EXEC_ARGS) { return cmd.info.gelu.tanh ? back_from_input(OpGeluTanhBack(), IO) : back_from_input(OpGeluErfBack(), IO); }
static int _swish_forw(EXEC_ARGS) { return unary_forw(OpSwish(), IO); }
static int _swish_back(EXEC_ARGS) { return back_from_input(OpSwishBack(), IO); }
static int _leaky_forw(EXEC_ARGS) { OpLeaky f = { cmd.info.leaky_relu.negative_slope }; return unary_forw(f, IO); }
static int _leaky_back(EXEC_ARGS)
{ // (g, _, b) -> h, g required (leaky_relu_cpu_ref.c:38-60)
	if (input_size < 3 || !inputs[0]) return CCV_NNC_EXEC_INVALID;
	OpLeakyBack f = { cmd.info.leaky_relu.negative_slope };
	return back_from_output(f, f, IO);
}

// ---- softmax over rows: one 256-thread block per row, two-pass (max, sum of exp) with the row kept in registers/L2 ---------
__device__ __forceinline__ float block_reduce(float v, float* red, const bool is_max)
{
	for (int o = 32; o > 0; o >>= 1) { const float w = __shfl_xor(v, o); v = is_max ? fmaxf(v, w) : v + w; }
	const int wave = threadIdx.x >> 6;
	__syncthreads();
	if ((threadIdx.x & 63) == 0) red[wave] = v;
	__syncthreads();
	float r = red[0];
	for (int i = 1; i < 4; i++) r = is_max ? fmaxf(r, red[i]) : r + red[i];
	return r;
}
__global__ void __launch_bounds__(256) softmax_forw_kernel(const float* a, float* b, const int count)
{
	__shared__ float red[4];
	const float* const ap = a + (size_t)blockIdx.x * count;
	float* const bp = b + (size_t)blockIdx.x * count;
	float m = -INFINITY;
	for (int j = threadIdx.x; j < count; j += 256) m = fmaxf(m, ap[j]);
	m = block_reduce(m, red, true);
	float s = 0.f;
	for (int j = threadIdx.x; j < count; j += 256) { const float e = expf(ap[j] - m); bp[j] = e; s += e; }
	s = block_reduce(s, red, false);
	const float inv = 1.f / s;
	for (int j = threadIdx.x; j < count; j += 256) bp[j] *= inv;
}
__global__ void __launch_bounds__(256) softmax_back_kernel(const float* g, const float* b, float* h, const int count)
{
	__shared__ float red[4];
	const size_t o = (size_t)blockIdx.x * count;
	float s = 0.f;
	for (int j = threadIdx.x; j < count; j += 256) s += g[o + j] * b[o + j];
	s = block_reduce(s, red, false);
	for (int j = threadIdx.x; j < count; j += 256) h[o + j] = (g[o + j] - s) * b[o + j];
}
static int _softmax_forw(EXEC_ARGS)
{
	if (input_size < 1 || output_size < 1 || !dense_f32(inputs[0]) || !dense_f32(outputs[0]) || !same_count(inputs[0], outputs[0])) return CCV_NNC_EXEC_INVALID;
	const ccv_nnc_tensor_t* a = inputs[0];
	const int batch = tensor_nd(a->info.dim) < 2 ? 1 : a->info.dim[0];
	const size_t n = tensor_count(a->info);
	if (n == 0) return CCV_NNC_EXEC_SUCCESS;
	hipLaunchKernelGGL(softmax_forw_kernel, dim3(batch), dim3(256), 0, stream_of(stream_context), (const float*)a->data.f32, outputs[0]->data.f32, (int)(n / batch));
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}
static int _softmax_back(EXEC_ARGS)
{
	if (input_size < 3 || output_size < 1 || !dense_f32(inputs[0]) || !dense_f32(inputs[2]) || !dense_f32(outputs[0]) || !same_count(inputs[0], inputs[2]) || !same_count(inputs[0], outputs[0])) return CCV_NNC_EXEC_INVALID;
	const ccv_nnc_tensor_t* g = inputs[0];
	const int batch = tensor_nd(g->info.dim) < 2 ? 1 : g->info.dim[0];
	const size_t n = tensor_count(g->info);
	if (n == 0) return CCV_NNC_EXEC_SUCCESS;
	hipLaunchKernelGGL(softmax_back_kernel, dim3(batch), dim3(256), 0, stream_of(stream_context), (const float*)g->data.f32, (const float*)inputs[2]->data.f32, outputs[0]->data.f32, (int)(n / batch));
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}

// ---- optimizers: one pass over (g, a, m, v[, vm]) -> (b, n, u[, um]); 7-9 |p| bytes -------------------------------------------
struct AdamP { float scale, beta1, beta2, decay, epsilon, rate_corr1, inv_corr2, rate_decay; int decoupled, amsgrad; };
__global__ void __launch_bounds__(EW_THREADS) adam_kernel(const AdamP p, const float* g, const float* a, const float* m, const float* v, const float* vm, float* b, float* nm, float* u, float* um, const size_t n)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
		const float av = a[i];
		float grad = p.scale * g[i];
		if (!p.decoupled) grad += p.decay * av;
		const float mom = p.beta1 * m[i] + (1.f - p.beta1) * grad;
		const float vel = p.beta2 * v[i] + (1.f - p.beta2) * grad * grad;
		nm[i] = mom;
		u[i] = vel;
		float denom;
		if (p.amsgrad) {
			const float vel_max_hat = fmaxf(vm[i], vel * p.inv_corr2);
			um[i] = vel_max_hat;
			denom = sqrtf(vel_max_hat) + p.epsilon;
		} else
			denom = sqrtf(vel * p.inv_corr2) + p.epsilon;
		const float base = p.decoupled ? av - p.rate_decay * av : av;
		b[i] = base - (mom * p.rate_corr1) / denom;
	}
}
static int adam_exec(const ccv_nnc_cmd_t& cmd, const int decoupled, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const ctx)
{
	if (input_size < 4 || output_size < 3) return CCV_NNC_EXEC_INVALID;
	for (int i = 0; i < 4; i++) if (!dense_f32(inputs[i])) return CCV_NNC_EXEC_INVALID;
	for (int i = 0; i < 3; i++) if (!dense_f32(outputs[i])) return CCV_NNC_EXEC_INVALID;
	const ccv_nnc_tensor_t* vm = input_size >= 5 ? inputs[4] : 0;
	ccv_nnc_tensor_t* um = output_size >= 4 ? outputs[3] : 0;
	const int ams = cmd.info.adam.amsgrad && vm && um;
	if (ams && (!dense_f32(vm) || !dense_f32(um))) return CCV_NNC_EXEC_INVALID;
	const size_t n = tensor_count(inputs[1]->info);
	for (int i = 0; i < 4; i++) if (tensor_count(inputs[i]->info) != n) return CCV_NNC_EXEC_INVALID;
	for (int i = 0; i < 3; i++) if (tensor_count(outputs[i]->info) != n) return CCV_NNC_EXEC_INVALID;
	if (n == 0) return CCV_NNC_EXEC_SUCCESS;
	AdamP p;
	p.scale = cmd.info.adam.scale; p.beta1 = cmd.info.adam.beta1; p.beta2 = cmd.info.adam.beta2; p.decay = cmd.info.adam.decay; p.epsilon = cmd.info.adam.epsilon;
	p.rate_corr1 = cmd.info.adam.rate / (1 - powf(p.beta1, (float)cmd.info.adam.step));
	p.inv_corr2 = 1.f / (1 - powf(p.beta2, (float)cmd.info.adam.step));
	p.rate_decay = cmd.info.adam.rate * p.decay;
	p.decoupled = decoupled; p.amsgrad = ams;
	hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, EW_THREADS)), dim3(EW_THREADS), 0, stream_of(ctx), p, (const float*)inputs[0]->data.f32, (const float*)inputs[1]->data.f32, (const float*)inputs[2]->data.f32, (const float*)inputs[3]->data.f32,
		ams ? (const float*)vm->data.f32 : (const float*)0, outputs[0]->data.f32, outputs[1]->data.f32, outputs[2]->data.f32, ams ? um->data.f32 : (float*)0, n);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}
static int _adam_forw(EXEC_ARGS) { return adam_exec(cmd, 0, IO); }
static int _adamw_forw(EXEC_ARGS) { return adam_exec(cmd, 1, IO); }

__global__ void __launch_bounds__(EW_THREADS) rmsprop_kernel(const float* g, const float* a, const float* m, const float* v, float* b, float* nm, float* u, const size_t n, const float rate, const float scale, const float decay, const float alpha, const float momentum, const float epsilon)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
		const float av = a[i];
		const float grad = scale * g[i] + decay * av;
		const float vel = alpha * v[i] + (1.f - alpha) * grad * grad;
		const float mom = momentum * m[i] + grad / (sqrtf(vel) + epsilon);
		u[i] = vel;
		nm[i] = mom;
		b[i] = av - rate * mom;
	}
}
static int _rmsprop_forw(EXEC_ARGS)
{
	if (input_size < 4 || output_size < 3) return CCV_NNC_EXEC_INVALID;
	for (int i = 0; i < 4; i++) if (!dense_f32(inputs[i])) return CCV_NNC_EXEC_INVALID;
	for (int i = 0; i < 3; i++) if (!dense_f32(outputs[i])) return CCV_NNC_EXEC_INVALID;
	const size_t n = tensor_count(inputs[1]->info);
	for (int i = 0; i < 4; i++) if (tensor_count(inputs[i]->info) != n) return CCV_NNC_EXEC_INVALID;
	for (int i = 0; i < 3; i++) if (tensor_count(outputs[i]->info) != n) return CCV_NNC_EXEC_INVALID;
	if (n == 0) return CCV_NNC_EXEC_SUCCESS;
	hipLaunchKernelGGL(rmsprop_kernel, dim3(grid_for(n, EW_THREADS)), dim3(EW_THREADS), 0, stream_of(stream_context), (const float*)inputs[0]->data.f32, (const float*)inputs[1]->data.f32, (const float*)inputs[2]->data.f32, (const float*)inputs[3]->data.f32,
		outputs[0]->data.f32, outputs[1]->data.f32, outputs[2]->data.f32, n, cmd.info.rmsprop.rate, cmd.info.rmsprop.scale, cmd.info.rmsprop.decay, cmd.info.rmsprop.alpha, cmd.info.rmsprop.momentum, cmd.info.rmsprop.epsilon);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}

// ---- LAMB: update = mom^ / (sqrt(vel^) + eps) + decay w; b = a - rate (|w| / |update|) update.  Three launches: the element pass writes n, u
// and the update (workspace) and one (sum w^2, sum update^2) pair per block in double; one block folds the pairs in order into the trust
// ratio; the last pass applies it.  Deterministic; 9 |p| bytes. -------------------------------------------------------------------------
struct LambP { float scale, beta1, beta2, decay, epsilon, inv_corr1, inv_corr2; };
__global__ void __launch_bounds__(EW_THREADS) lamb_update_kernel(const LambP p, const float* g, const float* a, const float* m, const float* v, float* nm, float* u, float* update, double* partial, const size_t n)
{
	__shared__ double red[2][4];
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	double wn = 0, un = 0;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
		const float grad = p.scale * g[i], w = a[i];
		const float mom = p.beta1 * m[i] + (1.f - p.beta1) * grad;
		const float vel = p.beta2 * v[i] + (1.f - p.beta2) * grad * grad;
		nm[i] = mom;
		u[i] = vel;
		const float upd = (mom * p.inv_corr1) / (sqrtf(vel * p.inv_corr2) + p.epsilon) + w * p.decay;
		update[i] = upd;
		wn += (double)(w * w);
		un += (double)(upd * upd);
	}
	for (int o = 32; o > 0; o >>= 1) { wn += __shfl_xor(wn, o); un += __shfl_xor(un, o); }
	if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = wn; red[1][threadIdx.x >> 6] = un; }
	__syncthreads();
	if (threadIdx.x == 0) {
		partial[2 * blockIdx.x] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
		partial[2 * blockIdx.x + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
	}
}
__global__ void lamb_trust_kernel(const double* partial, const int blocks, const float rate, float* rate_trust)
{
	double wn = 0, un = 0;
	for (int i = 0; i < blocks; i++) { wn += partial[2 * i]; un += partial[2 * i + 1]; }
	wn = sqrt(wn); un = sqrt(un);
	const float trust = (wn > 0 && un > 0) ? (float)(wn / un) : 1.f;
	*rate_trust = rate * trust;
}
__global__ void __launch_bounds__(EW_THREADS) lamb_apply_kernel(const float* a, const float* update, const float* rate_trust, float* b, const size_t n)
{
	const float rt = *rate_trust;
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) b[i] = a[i] - rt * update[i];
}
static int _lamb_forw(EXEC_ARGS)
{
	if (input_size < 4 || output_size < 3) return CCV_NNC_EXEC_INVALID;
	for (int i = 0; i < 4; i++) if (!dense_f32(inputs[i])) return CCV_NNC_EXEC_INVALID;
	for (int i = 0; i < 3; i++) if (!dense_f32(outputs[i])) return CCV_NNC_EXEC_INVALID;
	const size_t n = tensor_count(inputs[1]->info);
	for (int i = 0; i < 4; i++) if (tensor_count(inputs[i]->info) != n) return CCV_NNC_EXEC_INVALID;
	for (int i = 0; i < 3; i++) if (tensor_count(outputs[i]->info) != n) return CCV_NNC_EXEC_INVALID;
	if (n == 0) return CCV_NNC_EXEC_SUCCESS;
	int blocks = grid_for(n, EW_THREADS); // one partial norm pair per workgroup, folded by one thread: keep them few
	if (blocks > device_cu_count() * 8) blocks = device_cu_count() * 8;
	const size_t head = (sizeof(double) * 2 * (size_t)blocks + sizeof(float) + 255) & ~(size_t)255;
	char* ws = (char*)workspace_of(stream_context, head + sizeof(float) * n);
	if (!ws) return CCV_NNC_EXEC_OOM;
	double* const partial = (double*)ws;
	float* const rate_trust = (float*)(ws + sizeof(double) * 2 * (size_t)blocks);
	float* const update = (float*)(ws + head);
	LambP p;
	p.scale = cmd.info.lamb.scale; p.beta1 = cmd.info.lamb.beta1; p.beta2 = cmd.info.lamb.beta2; p.decay = cmd.info.lamb.decay; p.epsilon = cmd.info.lamb.epsilon;
	p.inv_corr1 = 1.f / (1 - powf(p.beta1, (float)cmd.info.lamb.step));
	p.inv_corr2 = 1.f / (1 - powf(p.beta2, (float)cmd.info.lamb.step));
	hipStream_t stream = stream_of(stream_context);
	hipLaunchKernelGGL(lamb_update_kernel, dim3(blocks), dim3(EW_THREADS), 0, stream, p, (const float*)inputs[0]->data.f32, (const float*)inputs[1]->data.f32, (const float*)inputs[2]->data.f32, (const float*)inputs[3]->data.f32,
		outputs[1]->data.f32, outputs[2]->data.f32, update, partial, n);
	hipLaunchKernelGGL(lamb_trust_kernel, dim3(1), dim3(1), 0, stream, (const double*)partial, blocks, cmd.info.lamb.rate, rate_trust);
	hipLaunchKernelGGL(lamb_apply_kernel, dim3(blocks), dim3(EW_THREADS), 0, stream, (const float*)inputs[1]->data.f32, (const float*)update, (const float*)rate_trust, outputs[0]->data.f32, n);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}

} // namespace

#define NNC_REG(CMD, BACKEND, FORMATS, EXEC) \
	extern "C" void _register_command_##CMD##_backend_##BACKEND(ccv_nnc_cmd_backend_registry_t* const registry) \
	{ registry->tensor_formats = (FORMATS); registry->tensor_datatypes = CCV_32F; registry->tensor_memory = CCV_TENSOR_GPU_MEMORY; registry->algorithms = 1; registry->exec = EXEC; NNC_HALF_STAGED(registry, EXEC); }
#define ALL_FORMATS (CCV_TENSOR_FORMAT_NCHW | CCV_TENSOR_FORMAT_NHWC | CCV_TENSOR_FORMAT_CHWN)

NNC_REG(CCV_NNC_SIGMOID_FORWARD, CCV_NNC_BACKEND_GPU_CUDNN, ALL_FORMATS, _sigmoid_forw)
NNC_REG(CCV_NNC_SIGMOID_BACKWARD, CCV_NNC_BACKEND_GPU_CUDNN, ALL_FORMATS, _sigmoid_back)
NNC_REG(CCV_NNC_TANH_FORWARD, CCV_NNC_BACKEND_GPU_CUDNN, ALL_FORMATS, _tanh_forw)
NNC_REG(CCV_NNC_TANH_BACKWARD, CCV_NNC_BACKEND_GPU_CUDNN, ALL_FORMATS, _tanh_back)
NNC_REG(CCV_NNC_GELU_FORWARD, CCV_NNC_BACKEND_GPU_REF, ALL_FORMATS, _gelu_forw)
NNC_REG(CCV_NNC_GELU_BACKWARD, CCV_NNC_BACKEND_GPU_REF, ALL_FORMATS, _gelu_back)
NNC_REG(CCV_NNC_SWISH_FORWARD, CCV_NNC_BACKEND_GPU_REF, ALL_FORMATS, _swish_forw)
NNC_REG(CCV_NNC_SWISH_BACKWARD, CCV_NNC_BACKEND_GPU_REF, ALL_FORMATS, _swish_back)
NNC_REG(CCV_NNC_LEAKY_RELU_FORWARD, CCV_NNC_BACKEND_GPU_REF, ALL_FORMATS, _leaky_forw)
NNC_REG(CCV_NNC_LEAKY_RELU_BACKWARD, CCV_NNC_BACKEND_GPU_REF, ALL_FORMATS, _leaky_back)
NNC_REG(CCV_NNC_SOFTMAX_FORWARD, CCV_NNC_BACKEND_GPU_CUDNN, ALL_FORMATS, _softmax_forw)
NNC_REG(CCV_NNC_SOFTMAX_BACKWARD, CCV_NNC_BACKEND_GPU_CUDNN, ALL_FORMATS, _softmax_back)
NNC_REG(CCV_NNC_ADAM_FORWARD, CCV_NNC_BACKEND_GPU_REF, ALL_FORMATS, _adam_forw)
NNC_REG(CCV_NNC_ADAMW_FORWARD, CCV_NNC_BACKEND_GPU_REF, ALL_FORMATS, _adamw_forw)
NNC_REG(CCV_NNC_RMSPROP_FORWARD, CCV_NNC_BACKEND_GPU_REF, ALL_FORMATS, _rmsprop_forw)
NNC_REG(CCV_NNC_LAMB_FORWARD, CCV_NNC_BACKEND_GPU_REF, ALL_FORMATS, _lamb_forw)
