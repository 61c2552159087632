// Bandwidth-bound element-wise commands on gfx950: RELU, EWSUM, SCALAR_MUL, SGD, SET, DATA_TRANSFER, plus the
// column-sum used for bias gradients.  16 bytes per lane where the tensors allow it; the element-wise map is a tile per workgroup over a full grid, the others grid-stride
// (coalesced dwordx4), otherwise 4 bytes per lane.  Roofline for every kernel here is HBM (8 TB/s spec);
// algorithmic bytes are listed per kernel.
// Oracle semantics:
//   relu      lib/nnc/cmd/relu/ccv_nnc_relu_cpu_ref.c:13-55            (gpu: relu/gpu/ccv_nnc_relu_gpu_cudnn.cu)
//   ewsum     lib/nnc/cmd/ew/ccv_nnc_ew_cpu_ref.c:15-233               (gpu: ew/gpu/ccv_nnc_ew_gpu_cudnn.cu:13-130)
//   sgd       lib/nnc/cmd/sgd/ccv_nnc_sgd_cpu_ref.c:16-126             (gpu: sgd/gpu/ccv_nnc_sgd_gpu_ref.cu:13-100)
//   set/xfer  lib/nnc/cmd/util/ccv_nnc_util_cpu_ref.c:596-664          (gpu: util/gpu/ccv_nnc_util_gpu_ref.cu:13-62)
//   scalar    lib/nnc/cmd/blas/ccv_nnc_mul_cpu_ref.c:417-430
#include "common.h"

using namespace nnc;

namespace {

constexpr int EW_THREADS = 256;

// ---- generic contiguous map: out[i] = f(in0[i], in1[i], in2[i]) ----------------------------------------------
// T = float or _Float16 (the CCV_16F tensors of the half-precision trainers: loaded and stored as halves, arithmetic in fp32 --
// half the bytes of the fp32 form, a fifth of what running the fp32 kernel on converted images moves).  16 bytes per access.
typedef _Float16 half_t;
template <class T> struct pack16 { typedef T type __attribute__((ext_vector_type(16 / sizeof(T)))); };
// Work split: a workgroup takes ONE contiguous tile of 4 x 256 16-byte vectors (16 KB per tensor), four independent loads per
// input in flight per thread, and the grid covers the tensor -- no grid-stride loop.  Measured on 3.29 GB tensors
// (tools/ew_bw_bench.py, profiles/r02_v8_ew_bw_bench.txt): RELU forward 6.2 TB/s this way against 4.7 TB/s for a grid-stride
// loop capped at 8 (or 16, 32, 64) workgroups per CU -- a few thousand workgroups striding the whole tensor scatter the DRAM
// pages in flight, a front of workgroups walking it in order does not; non-temporal loads / stores made no difference.
constexpr int EW_TILE = 4;
template <class F, int NIN, class T>
__global__ void __launch_bounds__(EW_THREADS) ew_map_kernel(F f, T* out, const T* in0, const T* in1, const T* in2, const size_t nv, const size_t n)
{
	constexpr int W = 16 / sizeof(T);
	typedef typename pack16<T>::type V;
	const size_t base = (size_t)blockIdx.x * (EW_TILE * EW_THREADS) + threadIdx.x;
	if (base + (EW_TILE - 1) * EW_THREADS < nv) {
		V a[EW_TILE], b[EW_TILE], c[EW_TILE];
#pragma unroll
		for (int u = 0; u < EW_TILE; u++) {
			a[u] = NIN > 0 ? ((const V*)in0)[base + u * EW_THREADS] : V{};
			b[u] = NIN > 1 ? ((const V*)in1)[base + u * EW_THREADS] : V{};
			c[u] = NIN > 2 ? ((const V*)in2)[base + u * EW_THREADS] : V{};
		}
#pragma unroll
		for (int u = 0; u < EW_TILE; u++) {
			V r;
#pragma unroll
			for (int e = 0; e < W; e++) r[e] = (T)f((float)a[u][e], (float)b[u][e], (float)c[u][e]);
			((V*)out)[base + u * EW_THREADS] = r;
		}
	} else
		for (int u = 0; u < EW_TILE; u++) {
			const size_t i = base + u * EW_THREADS;
			if (i >= nv) break;
			const V a = NIN > 0 ? ((const V*)in0)[i] : V{};
			const V b = NIN > 1 ? ((const V*)in1)[i] : V{};
			const V c = NIN > 2 ? ((const V*)in2)[i] : V{};
			V r;
#pragma unroll
			for (int e = 0; e < W; e++) r[e] = (T)f((float)a[e], (float)b[e], (float)c[e]);
			((V*)out)[i] = r;
		}
	// what 16-byte vectors do not cover (or everything, when a pointer is not 16-byte aligned: nv == 0)
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t j = nv * W + (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride)
		out[j] = (T)f(NIN > 0 ? (float)in0[j] : 0.f, NIN > 1 ? (float)in1[j] : 0.f, NIN > 2 ? (float)in2[j] : 0.f);
}

template <class F, int NIN, class T>
static int ew_map(F f, T* out, const T* in0, const T* in1, const T* in2, size_t n, ccv_nnc_stream_context_t* ctx)
{
	if (n == 0) return CCV_NNC_EXEC_SUCCESS;
	constexpr int W = 16 / sizeof(T);
	const bool vec = aligned16(out) && (NIN < 1 || aligned16(in0)) && (NIN < 2 || aligned16(in1)) && (NIN < 3 || aligned16(in2));
	const size_t nv = vec ? n / W : 0;
	size_t blocks = vec ? (nv + EW_TILE * EW_THREADS - 1) / (EW_TILE * EW_THREADS) : (n + EW_THREADS - 1) / EW_THREADS;
	if (blocks < 1) blocks = 1;
	if (blocks > 0x7fffffffUL) blocks = 0x7fffffffUL; // (the scalar loop strides; a vector tensor this large does not exist on one GPU)
	hipLaunchKernelGGL(HIP_KERNEL_NAME(ew_map_kernel<F, NIN, T>), dim3((unsigned)blocks), dim3(EW_THREADS), 0, stream_of(ctx), f, out, in0, in1, in2, nv, n);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}
template <class F, int NIN>
static int ew_map(F f, float* out, const float* in0, const float* in1, const float* in2, size_t n, ccv_nnc_stream_context_t* ctx)
{
	return ew_map<F, NIN, float>(f, out, in0, in1, in2, n, ctx);
}
// the tensors' own element type: every tensor handed over is of a's type (half_stage.cpp keeps the big tensors of the rows
// listed in its native table in half precision only when all of them are)
template <class F, int NIN>
static int ew_map_any(F f, const int datatype, void* out, const void* in0, const void* in1, const void* in2, size_t n, ccv_nnc_stream_context_t* ctx)
{
	if (CCV_GET_DATA_TYPE(datatype) == CCV_16F) return ew_map<F, NIN, half_t>(f, (half_t*)out, (const half_t*)in0, (const half_t*)in1, (const half_t*)in2, n, ctx);
	return ew_map<F, NIN, float>(f, (float*)out, (const float*)in0, (const float*)in1, (const float*)in2, n, ctx);
}

struct OpRelu { __device__ float operator()(float a, float, float) const { return a > 0.f ? a : 0.f; } };                  // 2|x| bytes
struct OpReluBack { __device__ float operator()(float g, float b, float) const { return b > 0.f ? g : 0.f; } };           // 3|x|
struct OpReluBackOnes { __device__ float operator()(float b, float, float) const { return b > 0.f ? 1.f : 0.f; } };
struct OpFill { float v; __device__ float operator()(float, float, float) const { return v; } };                           // |x|
struct OpScale { float s; __device__ float operator()(float a, float, float) const { return s * a; } };                    // 2|x|
struct OpAdd2 { __device__ float operator()(float a, float b, float) const { return a + b; } };                            // 3|x|
struct OpAdd3 { __device__ float operator()(float a, float b, float c) const { return a + b + c; } };
struct OpAdd2Relu { __device__ float operator()(float a, float b, float) const { const float t = a + b; return t > 0.f ? t : 0.f; } };       // EWSUM + the ReLU behind it
struct OpAdd3Relu { __device__ float operator()(float a, float b, float c) const { const float t = a + b + c; return t > 0.f ? t : 0.f; } };
struct OpAdd2ReluBack { __device__ float operator()(float a, float b, float m) const { return m > 0.f ? a + b : 0.f; } };                         // EWSUM + the RELU_BACKWARD behind it (4|x| instead of 6|x|)
struct OpCopy { __device__ float operator()(float a, float, float) const { return a; } };

// ---- SGD: n = mu*m + (1-damp)*(scale*g + decay*a); b = a - rate*n   (5|p| bytes: g, a, m in; b, n out) --------
// ONE definition of the arithmetic for every kernel below (scalar, 16-byte, multi-tensor), with contraction off: each product and sum is rounded on its own, as
// the reference's CPU loops do (lib/nnc/cmd/sgd/ccv_nnc_sgd_cpu_ref.c:16-126) -- and the forms are bit-identical to each other by construction.  (Round 6: left
// to the compiler, the multi-tensor kernel fused a multiply-add the 16-byte half kernel did not, and the two differed in the last bit on the MI355X.)
__device__ __forceinline__ void sgd_update(const float g, const float a, const float m, const int nesterov, const float rate, const float scale, const float decay, const float momentum, const float inv_dampening, float* const b, float* const n)
{
#pragma clang fp contract(off)
	if (nesterov) {
		float grad = scale * g;
		const float mom = momentum * m + grad + decay * a;
		*n = mom;
		grad += momentum * mom;
		*b = a - rate * grad;
	} else {
		const float mom = momentum * m + inv_dampening * (scale * g + decay * a);
		*n = mom;
		*b = a - rate * mom;
	}
}
template <class T> // float, or _Float16 when the trainer keeps gradient, parameter and momentum in CCV_16F (arithmetic in fp32)
__global__ void __launch_bounds__(EW_THREADS) sgd_kernel(const T* g, const T* a, const T* m, T* b, T* nm, const size_t n, const int nesterov, const float rate, const float scale, const float decay, const float momentum, const float inv_dampening)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
		float bo, no;
		sgd_update((float)g[i], (float)a[i], (float)m[i], nesterov, rate, scale, decay, momentum, inv_dampening, &bo, &no);
		nm[i] = (T)no;
		b[i] = (T)bo;
	}
}
// The same for CCV_16F tensors in 16-byte accesses (eight halves per lane; round 5): the scalar form above moves 2 bytes per lane and access -- ResNet-50's 161
// half-precision parameter tensors took 1.9 ms of its 48 ms step that way (profiles/r05_v8_resnet50-nchw-bs256-f16_rocprofv3_kernel_stats.md).  Same fp32
// arithmetic and rounding per element: bit-identical results.
__global__ void __launch_bounds__(EW_THREADS) sgd_kernel_h8(const pack16<half_t>::type* g, const pack16<half_t>::type* a, const pack16<half_t>::type* m, pack16<half_t>::type* b, pack16<half_t>::type* nm, const size_t n8, const int nesterov, const float rate, const float scale, const float decay, const float momentum, const float inv_dampening)
{
	typedef pack16<half_t>::type V;
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
		const V gv = g[i], av = a[i], mv = m[i];
		V bo, no;
#pragma unroll
		for (int e = 0; e < 8; e++) {
			float bf, nf;
			sgd_update((float)gv[e], (float)av[e], (float)mv[e], nesterov, rate, scale, decay, momentum, inv_dampening, &bf, &nf);
			no[e] = (half_t)nf;
			bo[e] = (half_t)bf;
		}
		nm[i] = no;
		b[i] = bo;
	}
}
__global__ void __launch_bounds__(EW_THREADS) sgd_kernel_v4(const float4* g, const float4* a, const float4* m, float4* b, float4* nm, const size_t n4, const int nesterov, const float rate, const float scale, const float decay, const float momentum, const float inv_dampening)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
		const float4 gv = g[i], av = a[i], mv = m[i];
		const float gs[4] = { gv.x, gv.y, gv.z, gv.w }, as[4] = { av.x, av.y, av.z, av.w }, ms[4] = { mv.x, mv.y, mv.z, mv.w };
		float bo[4], no[4];
#pragma unroll
		for (int e = 0; e < 4; e++) sgd_update(gs[e], as[e], ms[e], nesterov, rate, scale, decay, momentum, inv_dampening, &bo[e], &no[e]);
		nm[i] = make_float4(no[0], no[1], no[2], no[3]);
		b[i] = make_float4(bo[0], bo[1], bo[2], bo[3]);
	}
}

// ---- multi-tensor SGD (round 6): ONE launch updates up to SGD_MULTI_MAX parameter tensors.  ResNet-50's step ends with ~200 SGD_FORWARD commands (5 - 18 us
// each, most of it launch latency: 4 % of the f16 step, and one host enqueue each); the look-ahead (peephole.cpp) keeps consecutive updates of a stream and
// hands them over together.  A workgroup finds its tensor by walking the table of first-block indices (wave-uniform: scalar loads out of the kernel
// arguments), then does exactly what sgd_kernel_h8 / sgd_kernel_v4 / sgd_kernel do for its vector -- the same expression per element, bit-identical results.
constexpr int SGD_MULTI_MAX = 25; // = one recorded update + a full trail (peephole.cpp TRAIL_MAX)
struct sgd_multi_t { const void* g[SGD_MULTI_MAX]; const void* a[SGD_MULTI_MAX]; const void* m[SGD_MULTI_MAX]; void* b[SGD_MULTI_MAX]; void* nm[SGD_MULTI_MAX]; size_t cnt[SGD_MULTI_MAX]; unsigned first[SGD_MULTI_MAX + 1]; };
template <class T, int W> // W elements per thread: 8 halves / 4 floats (16-byte accesses; every tensor's count a multiple of W, pointers 16-byte aligned), or 1
__global__ void __launch_bounds__(EW_THREADS) sgd_multi_kernel(const sgd_multi_t s, const int n, const int nesterov, const float rate, const float scale, const float decay, const float momentum, const float inv_dampening)
{
	int k = 0;
	while (k + 1 < n && blockIdx.x >= s.first[k + 1]) k++;
	const size_t i = (size_t)(blockIdx.x - s.first[k]) * EW_THREADS + threadIdx.x;
	if (i * W >= s.cnt[k]) return;
	typedef T V __attribute__((ext_vector_type(W)));
	const V gv = ((const V*)s.g[k])[i], av = ((const V*)s.a[k])[i], mv = ((const V*)s.m[k])[i];
	V bo, no;
#pragma unroll
	for (int e = 0; e < W; e++) {
		float bf, nf;
		sgd_update((float)gv[e], (float)av[e], (float)mv[e], nesterov, rate, scale, decay, momentum, inv_dampening, &bf, &nf);
		no[e] = (T)nf;
		bo[e] = (T)bf;
	}
	((V*)s.nm[k])[i] = no;
	((V*)s.b[k])[i] = bo;
}

// ---- column sum: out[c] (+)= sum_r x[r*ld + c].  Stage 1: grid (col tiles of 64, row slices); each wave owns one
// row phase, lanes = 64 consecutive columns (256-byte coalesced rows); stage 2 folds the slices in fixed order. ------
constexpr int CS_COLS = 64, CS_PHASES = 4;
__global__ void __launch_bounds__(256) colsum_partial_kernel(const float* x, const long rows, const int cols, const long ld, const long rows_per_slice, float* partial)
{
	__shared__ float red[CS_PHASES][CS_COLS];
	const int lane = threadIdx.x & 63, phase = threadIdx.x >> 6;
	const int c = blockIdx.x * CS_COLS + lane;
	const long r0 = (long)blockIdx.y * rows_per_slice;
	long r1 = r0 + rows_per_slice;
	if (r1 > rows) r1 = rows;
	float s = 0.f;
	if (c < cols)
		for (long r = r0 + phase; r < r1; r += CS_PHASES) s += x[r * ld + c];
	red[phase][lane] = s;
	__syncthreads();
	if (phase == 0 && c < cols) partial[(long)blockIdx.y * cols + c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}
// 16-byte variant (cols % 4 == 0, ld % 4 == 0, 16-byte aligned): 16 lanes x float4 cover the 64 columns of a tile, the
// other 16 lane groups of the block walk 16 row phases; fixed-order fold of the phases through LDS.
__global__ void __launch_bounds__(256) colsum_partial_v4_kernel(const float* x, const long rows, const int cols, const long ld, const long rows_per_slice, float* partial)
{
	__shared__ float4 red[16][16];
	const int q = threadIdx.x & 15, phase = threadIdx.x >> 4;
	const int c = blockIdx.x * CS_COLS + q * 4;
	const long r0 = (long)blockIdx.y * rows_per_slice;
	long r1 = r0 + rows_per_slice;
	if (r1 > rows) r1 = rows;
	float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
	if (c < cols)
		for (long r = r0 + phase; r < r1; r += 16) {
			const float4 v = *(const float4*)(x + r * ld + c);
			s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
		}
	red[phase][q] = s;
	__syncthreads();
	if (phase == 0 && c < cols) {
		float4 t = red[0][q];
		for (int p = 1; p < 16; p++) { const float4 u = red[p][q]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
		*(float4*)(partial + (long)blockIdx.y * cols + c) = t;
	}
}
__global__ void __launch_bounds__(256) colsum_final_kernel(const float* partial, const int slices, const int cols, float* out, const int accumulate)
{ // 16 columns x 16 phases per workgroup (common.h: fold_slices / fold_phases)
	__shared__ float red[FOLD_PH][FOLD_CH];
	const int ch = threadIdx.x & (FOLD_CH - 1), phase = threadIdx.x / FOLD_CH;
	const int c = blockIdx.x * FOLD_CH + ch;
	red[phase][ch] = c < cols ? fold_slices(partial, slices, cols, c, phase) : 0.f;
	__syncthreads();
	if (phase == 0 && c < cols) {
		const float s = fold_phases(red, ch);
		out[c] = accumulate ? out[c] + s : s;
	}
}

static int _relu_forw(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	if (input_size < 1 || output_size < 1 || !inputs[0] || !outputs[0]) return CCV_NNC_EXEC_INVALID;
	const ccv_nnc_tensor_t* a = inputs[0];
	ccv_nnc_tensor_t* b = outputs[0];
	if (!tensor_contiguous(a) || !tensor_contiguous(b) || tensor_count(a->info) != tensor_count(b->info)) return CCV_NNC_EXEC_INVALID;
	if (a->info.datatype != b->info.datatype) return CCV_NNC_EXEC_INVALID;
	const int folded = deferred_fuse_relu_forw(a, b, stream_context); // in place behind a recorded convolution: that one rectifies as it stores (peephole.cpp)
	if (folded >= 0) return folded;
	return ew_map_any<OpRelu, 1>(OpRelu(), a->info.datatype, b->data.u8, a->data.u8, 0, 0, tensor_count(a->info), stream_context);
}

static int _relu_back(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	// inputs (g, a [unused], b); output h = b > 0 ? g : 0; a null g means ones (reference cuDNN path passes ones)
	if (input_size < 3 || output_size < 1 || !inputs[2] || !outputs[0]) return CCV_NNC_EXEC_INVALID;
	const ccv_nnc_tensor_t* g = inputs[0];
	const ccv_nnc_tensor_t* b = inputs[2];
	ccv_nnc_tensor_t* h = outputs[0];
	if ((g && !tensor_contiguous(g)) || !tensor_contiguous(b) || !tensor_contiguous(h)) return CCV_NNC_EXEC_INVALID;
	const size_t n = tensor_count(b->info);
	if (tensor_count(h->info) != n || (g && tensor_count(g->info) != n)) return CCV_NNC_EXEC_INVALID;
	if (b->info.datatype != h->info.datatype || (g && g->info.datatype != h->info.datatype)) return CCV_NNC_EXEC_INVALID;
	const int folded = deferred_fuse_relu_back(g, b, h, stream_context); // in place on the gradient a recorded command is about to write, masked by the map it read
	if (folded >= 0) return folded;
	if (!g) return ew_map_any<OpReluBackOnes, 1>(OpReluBackOnes(), h->info.datatype, h->data.u8, b->data.u8, 0, 0, n, stream_context);
	return ew_map_any<OpReluBack, 2>(OpReluBack(), h->info.datatype, h->data.u8, g->data.u8, b->data.u8, 0, n, stream_context);
}

// int32 n-ary sum (the reference sums index tensors with it, ew_gpu_cudnn.cu:13-130): out = in0 + in1 + ...
struct i32_ptrs_t { const int* p[8]; };
__global__ void __launch_bounds__(EW_THREADS) ewsum_i32_kernel8(int* out, const i32_ptrs_t in, const int count, const size_t n)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
		int v = in.p[0][i];
		for (int k = 1; k < count; k++) v += in.p[k][i];
		out[i] = v;
	}
}

static int ewsum_forw_entry(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context);
static int _ewsum_forw(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	// recorded when its like has run before: the in-place RELU_FORWARD behind a residual sum folds into the pass that writes the sum (peephole.cpp)
	uint64_t sig;
	if (g_comm_overlap_on.load(std::memory_order_relaxed) && output_size >= 1 && outputs[0]) comm_gradient_touched(outputs[0]); // a gradient summed over several uses: its all-reduce takes stream order (cmd_comm.cpp "Overlap")
	if (const int e = deferred_take_error(stream_context)) return e;
	const bool floats = output_size >= 1 && outputs[0] && (CCV_GET_DATA_TYPE(outputs[0]->info.datatype) == CCV_32F || CCV_GET_DATA_TYPE(outputs[0]->info.datatype) == CCV_16F); // (no ReLU row for CCV_32S: nothing to wait for)
	sig = 0;
	if (floats && deferred_try(_ewsum_forw, DEFER_EWSUM_FORWARD, cmd, hint, flags, inputs, input_size, outputs, output_size, stream_context, &sig)) return CCV_NNC_EXEC_SUCCESS;
	const int r = ewsum_forw_entry(cmd, hint, flags, inputs, input_size, outputs, output_size, stream_context);
	if (r == CCV_NNC_EXEC_SUCCESS) deferred_mark_good(sig);
	return r;
}
static int ewsum_forw_entry(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	const bool relu = cmd.algorithm > 0 && (cmd.algorithm & NNC_MI355X_EWSUM_ALGO_FUSE_RELU); // opt-in, or the look-ahead completing this sum with its ReLU: c = max(0, sum)
	if (input_size < 1 || output_size < 1 || !outputs[0]) return CCV_NNC_EXEC_INVALID;
	ccv_nnc_tensor_t* c = outputs[0];
	if (cmd.algorithm > 0 && (cmd.algorithm & NNC_MI355X_EWSUM_ALGO_FUSE_RELU_BACKWARD)) {
		// (a0, a1, mask) -> c = mask > 0 ? a0 + a1 : 0: the sum rounded to the tensors' type exactly as the plain command rounds it, then masked as RELU_BACKWARD masks
		if (relu || input_size != 3 || !inputs[0] || !inputs[1] || !inputs[2]) return CCV_NNC_EXEC_INVALID;
		const int dt = c->info.datatype;
		if ((CCV_GET_DATA_TYPE(dt) != CCV_32F && CCV_GET_DATA_TYPE(dt) != CCV_16F) || !tensor_contiguous(c)) return CCV_NNC_EXEC_INVALID;
		const size_t n = tensor_count(c->info);
		for (int i = 0; i < 3; i++)
			if (!tensor_contiguous(inputs[i]) || tensor_count(inputs[i]->info) != n || inputs[i]->info.datatype != dt) return CCV_NNC_EXEC_INVALID;
		return ew_map_any<OpAdd2ReluBack, 3>(OpAdd2ReluBack(), dt, c->data.u8, inputs[0]->data.u8, inputs[1]->data.u8, inputs[2]->data.u8, n, stream_context);
	}
	if (CCV_GET_DATA_TYPE(c->info.datatype) == CCV_32S) {
		if (!tensor_contiguous(c) || input_size > 8 || relu) return CCV_NNC_EXEC_INVALID;
		const size_t n = tensor_count(c->info);
		i32_ptrs_t ptrs;
		for (int i = 0; i < input_size; i++) {
			if (!inputs[i] || !tensor_contiguous(inputs[i]) || tensor_count(inputs[i]->info) != n || CCV_GET_DATA_TYPE(inputs[i]->info.datatype) != CCV_32S) return CCV_NNC_EXEC_INVALID;
			ptrs.p[i] = inputs[i]->data.i32;
		}
		if (n == 0) return CCV_NNC_EXEC_SUCCESS;
		hipLaunchKernelGGL(ewsum_i32_kernel8, dim3(grid_for(n, EW_THREADS)), dim3(EW_THREADS), 0, stream_of(stream_context), c->data.i32, ptrs, input_size, n);
		HIP_ENFORCE(hipGetLastError());
		return CCV_NNC_EXEC_SUCCESS;
	}
	if ((CCV_GET_DATA_TYPE(c->info.datatype) != CCV_32F && CCV_GET_DATA_TYPE(c->info.datatype) != CCV_16F) || !tensor_contiguous(c)) return CCV_NNC_EXEC_INVALID;
	const size_t n = tensor_count(c->info);
	for (int i = 0; i < input_size; i++)
		if (!inputs[i] || !tensor_contiguous(inputs[i]) || tensor_count(inputs[i]->info) != n) return CCV_NNC_EXEC_INVALID;
	void* cp = c->data.u8;
	const int dt = c->info.datatype;
	for (int i = 0; i < input_size; i++) if (inputs[i]->info.datatype != dt) return CCV_NNC_EXEC_INVALID;
	// Fold left to right, exactly the association order of the reference: ((in0 + in1) + in2) + ...
	if (input_size == 1) {
		if (relu) return ew_map_any<OpRelu, 1>(OpRelu(), dt, cp, inputs[0]->data.u8, 0, 0, n, stream_context);
		if (inputs[0]->data.u8 == cp) return CCV_NNC_EXEC_SUCCESS;
		return ew_map_any<OpCopy, 1>(OpCopy(), dt, cp, inputs[0]->data.u8, 0, 0, n, stream_context);
	}
	// (the LAST pass of the fold rectifies when asked to: relu && i reaches input_size there)
	int ret, i = 0;
	if (input_size >= 3) {
		if (relu && input_size == 3) ret = ew_map_any<OpAdd3Relu, 3>(OpAdd3Relu(), dt, cp, inputs[0]->data.u8, inputs[1]->data.u8, inputs[2]->data.u8, n, stream_context);
		else ret = ew_map_any<OpAdd3, 3>(OpAdd3(), dt, cp, inputs[0]->data.u8, inputs[1]->data.u8, inputs[2]->data.u8, n, stream_context);
		i = 3;
	} else {
		if (relu) ret = ew_map_any<OpAdd2Relu, 2>(OpAdd2Relu(), dt, cp, inputs[0]->data.u8, inputs[1]->data.u8, 0, n, stream_context);
		else ret = ew_map_any<OpAdd2, 2>(OpAdd2(), dt, cp, inputs[0]->data.u8, inputs[1]->data.u8, 0, n, stream_context);
		i = 2;
	}
	for (; ret == CCV_NNC_EXEC_SUCCESS && i < input_size; ) {
		if (i + 1 < input_size) {
			if (relu && i + 2 == input_size) ret = ew_map_any<OpAdd3Relu, 3>(OpAdd3Relu(), dt, cp, cp, inputs[i]->data.u8, inputs[i + 1]->data.u8, n, stream_context);
			else ret = ew_map_any<OpAdd3, 3>(OpAdd3(), dt, cp, cp, inputs[i]->data.u8, inputs[i + 1]->data.u8, n, stream_context);
			i += 2;
		} else {
			if (relu) ret = ew_map_any<OpAdd2Relu, 2>(OpAdd2Relu(), dt, cp, cp, inputs[i]->data.u8, 0, n, stream_context);
			else ret = ew_map_any<OpAdd2, 2>(OpAdd2(), dt, cp, cp, inputs[i]->data.u8, 0, n, stream_context);
			i += 1;
		}
	}
	return ret;
}

static int _ewsum_back(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	// D[x + y + z, x] = 1: every output gradient is a copy of g (or ones when g is null). ew_cpu_ref.c:216-233
	int ret = CCV_NNC_EXEC_SUCCESS;
	const ccv_nnc_tensor_t* g = input_size > 0 ? inputs[0] : 0;
	for (int i = 0; i < output_size && ret == CCV_NNC_EXEC_SUCCESS; i++) {
		ccv_nnc_tensor_t* o = outputs[i];
		if (!o) continue;
		if (!tensor_contiguous(o)) return CCV_NNC_EXEC_INVALID;
		const size_t n = tensor_count(o->info);
		if (!g) { OpFill f; f.v = 1.f; ret = ew_map_any<OpFill, 0>(f, o->info.datatype, o->data.u8, 0, 0, 0, n, stream_context); }
		else if (g->data.f32 != o->data.f32) {
			if (!tensor_contiguous(g) || tensor_count(g->info) != n || g->info.datatype != o->info.datatype) return CCV_NNC_EXEC_INVALID;
			ret = ew_map_any<OpCopy, 1>(OpCopy(), o->info.datatype, o->data.u8, g->data.u8, 0, 0, n, stream_context);
		}
	}
	return ret;
}

static int _scalar_mul_forw(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	if (input_size < 1 || output_size < 1 || !inputs[0] || !outputs[0]) return CCV_NNC_EXEC_INVALID;
	if (!tensor_contiguous(inputs[0]) || !tensor_contiguous(outputs[0]) || tensor_count(inputs[0]->info) != tensor_count(outputs[0]->info)) return CCV_NNC_EXEC_INVALID;
	OpScale f; f.s = cmd.info.blas.a[0];
	return ew_map<OpScale, 1>(f, outputs[0]->data.f32, inputs[0]->data.f32, 0, 0, tensor_count(inputs[0]->info), stream_context);
}

static int _scalar_mul_back(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	if (output_size < 1 || !outputs[0] || !tensor_contiguous(outputs[0])) return CCV_NNC_EXEC_INVALID;
	const size_t n = tensor_count(outputs[0]->info);
	if (input_size > 0 && inputs[0]) {
		if (!tensor_contiguous(inputs[0]) || tensor_count(inputs[0]->info) != n) return CCV_NNC_EXEC_INVALID;
		OpScale f; f.s = cmd.info.blas.a[0];
		return ew_map<OpScale, 1>(f, outputs[0]->data.f32, inputs[0]->data.f32, 0, 0, n, stream_context);
	}
	OpFill f; f.v = cmd.info.blas.a[0];
	return ew_map<OpFill, 0>(f, outputs[0]->data.f32, 0, 0, 0, n, stream_context);
}

// n validated SGD_FORWARD commands of one stream (each passed _sgd_forw's checks when it arrived) in as few launches as their tensors allow: one per class
// (half / float, 16-byte accesses or single elements).  > 0: not batchable as a whole (different hyper-parameters, or an update that reads what an earlier one
// writes) -- the caller runs them one by one, in order.
static int sgd_multi_run(const ccv_nnc_cmd_t* const* const cmds, ccv_nnc_tensor_t* const* const* const ins, ccv_nnc_tensor_t* const* const* const outs, const int n, ccv_nnc_stream_context_t* const ctx)
{
	if (n < 2 || n > SGD_MULTI_MAX) return 1;
	for (int i = 1; i < n; i++) if (memcmp(&cmds[i]->info.sgd, &cmds[0]->info.sgd, sizeof(cmds[0]->info.sgd)) != 0) return 1;
	for (int j = 1; j < n; j++) // order between the updates must not matter: nothing one writes is read or written by another (in place WITHIN an update is the rule)
		for (int i = 0; i < j; i++)
			for (int o = 0; o < 2; o++) {
				const void* const wj = outs[j][o]->data.u8;
				const void* const wi = outs[i][o]->data.u8;
				for (int q = 0; q < 3; q++) if (ins[i][q]->data.u8 == wj || ins[j][q]->data.u8 == wi) return 1;
				for (int q = 0; q < 2; q++) if (outs[i][q]->data.u8 == wj) return 1;
			}
	const ccv_nnc_cmd_t& cmd = *cmds[0];
	const float inv_dampening = 1 - cmd.info.sgd.dampening;
	hipStream_t stream = stream_of(ctx);
	for (int cls = 0; cls < 4; cls++) { // 0: halves x 8, 1: halves, 2: floats x 4, 3: floats
		sgd_multi_t s;
		int k = 0;
		unsigned blocks = 0;
		for (int i = 0; i < n; i++) {
			const ccv_nnc_tensor_t* const g = ins[i][0]; const ccv_nnc_tensor_t* const a = ins[i][1]; const ccv_nnc_tensor_t* const m = ins[i][2];
			ccv_nnc_tensor_t* const b = outs[i][0]; ccv_nnc_tensor_t* const nn = outs[i][1];
			const size_t cnt = tensor_count(a->info);
			const bool half = CCV_GET_DATA_TYPE(a->info.datatype) == CCV_16F;
			const bool al = aligned16(g->data.u8) && aligned16(a->data.u8) && aligned16(m->data.u8) && aligned16(b->data.u8) && aligned16(nn->data.u8);
			const int mine = half ? ((cnt % 8 == 0 && al) ? 0 : 1) : ((cnt % 4 == 0 && al) ? 2 : 3);
			if (mine != cls) continue;
			const int W = cls == 0 ? 8 : cls == 2 ? 4 : 1;
			const size_t nb = (cnt / W + EW_THREADS - 1) / EW_THREADS;
			if (nb > 0x7fffffffu - blocks) return 1;
			s.g[k] = g->data.u8; s.a[k] = a->data.u8; s.m[k] = m->data.u8; s.b[k] = b->data.u8; s.nm[k] = nn->data.u8; s.cnt[k] = cnt; s.first[k] = blocks;
			blocks += (unsigned)nb;
			k++;
		}
		if (!k) continue;
		s.first[k] = blocks;
#define SGD_MULTI(T, W) hipLaunchKernelGGL(HIP_KERNEL_NAME(sgd_multi_kernel<T, W>), dim3(blocks), dim3(EW_THREADS), 0, stream, s, k, cmd.info.sgd.nesterov, cmd.info.sgd.rate, cmd.info.sgd.scale, cmd.info.sgd.decay, cmd.info.sgd.momentum, inv_dampening)
		if (cls == 0) SGD_MULTI(half_t, 8); else if (cls == 1) SGD_MULTI(half_t, 1); else if (cls == 2) SGD_MULTI(float, 4); else SGD_MULTI(float, 1);
#undef SGD_MULTI
		HIP_ENFORCE(hipGetLastError());
	}
	return CCV_NNC_EXEC_SUCCESS;
}

static int _sgd_forw(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	if (input_size != 3 || output_size != 2) return CCV_NNC_EXEC_INVALID;
	const ccv_nnc_tensor_t* g = inputs[0];
	const ccv_nnc_tensor_t* a = inputs[1];
	const ccv_nnc_tensor_t* m = inputs[2];
	ccv_nnc_tensor_t* b = outputs[0];
	ccv_nnc_tensor_t* n = outputs[1];
	if (!g || !a || !m || !b || !n) return CCV_NNC_EXEC_INVALID;
	if (!tensor_contiguous(g) || !tensor_contiguous(a) || !tensor_contiguous(m) || !tensor_contiguous(b) || !tensor_contiguous(n)) return CCV_NNC_EXEC_INVALID;
	const size_t cnt = tensor_count(a->info);
	if (tensor_count(g->info) != cnt || tensor_count(m->info) != cnt || tensor_count(b->info) != cnt || tensor_count(n->info) != cnt) return CCV_NNC_EXEC_INVALID;
	if (cmd.info.sgd.nesterov && cmd.info.sgd.dampening != 0) return CCV_NNC_EXEC_INVALID;
	if (cnt == 0) return CCV_NNC_EXEC_SUCCESS;
	const float inv_dampening = 1 - cmd.info.sgd.dampening;
	const int dt = CCV_GET_DATA_TYPE(a->info.datatype);
	if (CCV_GET_DATA_TYPE(g->info.datatype) != dt || CCV_GET_DATA_TYPE(m->info.datatype) != dt || CCV_GET_DATA_TYPE(b->info.datatype) != dt || CCV_GET_DATA_TYPE(n->info.datatype) != dt) return CCV_NNC_EXEC_INVALID;
	// an update of this stream that waited in a trail failed when it was replayed: its caller had been told "enqueued", this one is told (once)
	if (const int e = deferred_take_error(stream_context)) return e;
	// every parameter has been checked: behind a recorded CONVOLUTION_BACKWARD whose signal this stream waits for, the update waits with it (peephole.cpp, the trail)
	if (g_deferred_live && deferred_trail_cmd(_sgd_forw, cmd, hint, flags, inputs, input_size, outputs, output_size, stream_context)) return CCV_NNC_EXEC_SUCCESS;
	// ... and an update with nothing recorded in front of it starts a batch of its own: the updates that follow it on this stream join it, one multi-tensor
	// launch for all of them at the stream's next order-observing point (peephole.cpp, "SGD batches")
	if (deferred_sgd_head(_sgd_forw, cmd, hint, flags, inputs, input_size, outputs, output_size, stream_context)) return CCV_NNC_EXEC_SUCCESS;
	hipStream_t stream = stream_of(stream_context);
	deferred_sgd_launched(stream_context);
	if (dt == CCV_16F) {
		if ((cnt % 8 == 0) && aligned16(g->data.u8) && aligned16(a->data.u8) && aligned16(m->data.u8) && aligned16(b->data.u8) && aligned16(n->data.u8)) {
			typedef pack16<half_t>::type V;
			hipLaunchKernelGGL(sgd_kernel_h8, dim3(grid_for(cnt / 8, EW_THREADS)), dim3(EW_THREADS), 0, stream, (const V*)g->data.u8, (const V*)a->data.u8, (const V*)m->data.u8, (V*)b->data.u8, (V*)n->data.u8, cnt / 8, cmd.info.sgd.nesterov, cmd.info.sgd.rate, cmd.info.sgd.scale, cmd.info.sgd.decay, cmd.info.sgd.momentum, inv_dampening);
			HIP_ENFORCE(hipGetLastError());
			return CCV_NNC_EXEC_SUCCESS;
		}
		hipLaunchKernelGGL(HIP_KERNEL_NAME(sgd_kernel<half_t>), dim3(grid_for(cnt, EW_THREADS)), dim3(EW_THREADS), 0, stream, (const half_t*)g->data.f16, (const half_t*)a->data.f16, (const half_t*)m->data.f16, (half_t*)b->data.f16, (half_t*)n->data.f16, cnt, cmd.info.sgd.nesterov, cmd.info.sgd.rate, cmd.info.sgd.scale, cmd.info.sgd.decay, cmd.info.sgd.momentum, inv_dampening);
		HIP_ENFORCE(hipGetLastError());
		return CCV_NNC_EXEC_SUCCESS;
	}
	const bool vec = (cnt % 4 == 0) && aligned16(g->data.f32) && aligned16(a->data.f32) && aligned16(m->data.f32) && aligned16(b->data.f32) && aligned16(n->data.f32);
	if (vec)
		hipLaunchKernelGGL(sgd_kernel_v4, dim3(grid_for(cnt / 4, EW_THREADS)), dim3(EW_THREADS), 0, stream, (const float4*)g->data.f32, (const float4*)a->data.f32, (const float4*)m->data.f32, (float4*)b->data.f32, (float4*)n->data.f32, cnt / 4, cmd.info.sgd.nesterov, cmd.info.sgd.rate, cmd.info.sgd.scale, cmd.info.sgd.decay, cmd.info.sgd.momentum, inv_dampening);
	else
		hipLaunchKernelGGL(HIP_KERNEL_NAME(sgd_kernel<float>), dim3(grid_for(cnt, EW_THREADS)), dim3(EW_THREADS), 0, stream, (const float*)g->data.f32, (const float*)a->data.f32, (const float*)m->data.f32, b->data.f32, n->data.f32, cnt, cmd.info.sgd.nesterov, cmd.info.sgd.rate, cmd.info.sgd.scale, cmd.info.sgd.decay, cmd.info.sgd.momentum, inv_dampening);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}

__global__ void __launch_bounds__(EW_THREADS) fill_f64_kernel(double* p, const size_t n, const double v)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

static int _set_forw(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	for (int i = 0; i < output_size; i++) {
		ccv_nnc_tensor_t* o = outputs[i];
		if (!o) continue;
		if (!tensor_contiguous(o)) return CCV_NNC_EXEC_INVALID;
		const size_t n = tensor_count(o->info);
		const int dt = CCV_GET_DATA_TYPE(o->info.datatype);
		if (cmd.info.blas.a[0] == 0) { HIP_ENFORCE(hipMemsetAsync(o->data.u8, 0, n * datatype_size(dt), stream_of(stream_context))); continue; }
		if (dt == CCV_32F) { const int r = fill_f32(o->data.f32, n, cmd.info.blas.a[0], stream_context); if (r) return r; }
		else if (dt == CCV_16F) { // natively (round 6: the trainers' ~50 SET commands per step used to fill an fp32 image and convert it down): pairs of the half value as one 32-bit pattern
			OpFill f; f.v = cmd.info.blas.a[0];
			const int r = ew_map_any<OpFill, 0>(f, o->info.datatype, o->data.u8, 0, 0, 0, n, stream_context);
			if (r) return r;
		}
		else if (dt == CCV_32S) { union { int i; float f; } u; u.i = (int)cmd.info.blas.a[0]; const int r = fill_f32(o->data.f32, n, u.f, stream_context); if (r) return r; }
		else if (dt == CCV_64F) { // a double is two 32-bit halves: fill pairs
			union { double d; float f[2]; } u; u.d = (double)cmd.info.blas.a[0];
			if (u.f[0] == u.f[1]) { const int r = fill_f32(o->data.f32, n * 2, u.f[0], stream_context); if (r) return r; }
			else { hipLaunchKernelGGL(fill_f64_kernel, dim3(grid_for(n, EW_THREADS)), dim3(EW_THREADS), 0, stream_of(stream_context), o->data.f64, n, u.d); HIP_ENFORCE(hipGetLastError()); }
		} else return CCV_NNC_EXEC_INVALID;
	}
	return CCV_NNC_EXEC_SUCCESS;
}

static int _set_back(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	for (int i = 0; i < output_size; i++) {
		ccv_nnc_tensor_t* o = outputs[i];
		if (!o) continue;
		if (!tensor_contiguous(o)) return CCV_NNC_EXEC_INVALID;
		HIP_ENFORCE(hipMemsetAsync(o->data.u8, 0, tensor_count(o->info) * datatype_size(o->info.datatype), stream_of(stream_context)));
	}
	return CCV_NNC_EXEC_SUCCESS;
}

static int _data_transfer(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	const int cnt = input_size < output_size ? input_size : output_size;
	for (int i = 0; i < cnt; i++) {
		const ccv_nnc_tensor_t* a = inputs[i];
		ccv_nnc_tensor_t* b = outputs[i];
		if (!a || !b || a == b) continue;
		if (!tensor_contiguous(a) || !tensor_contiguous(b) || tensor_count(a->info) != tensor_count(b->info)) return CCV_NNC_EXEC_INVALID;
		size_t size;
		if (CCV_GET_DATA_TYPE(a->info.datatype) == CCV_QX) { // a palettized tensor travels as its byte stream (ccv_nnc_util_gpu_ref.cu:22-26)
			if (a->info.datatype != b->info.datatype || a->info.reserved != b->info.reserved) return CCV_NNC_EXEC_INVALID;
			size = palettized_bytes((a->info.datatype & 0xff) << 12, tensor_count(a->info), (a->info.datatype & 0xf00) >> 8, a->info.reserved);
			if (!size && tensor_count(a->info)) return CCV_NNC_EXEC_INVALID; // (no block size / a bit width outside 4 .. 8: not a stream anybody can size)
		} else {
			if (datatype_size(a->info.datatype) != datatype_size(b->info.datatype)) return CCV_NNC_EXEC_INVALID;
			size = tensor_count(a->info) * datatype_size(a->info.datatype);
		}
		const int am = CCV_TENSOR_GET_MEMORY(a->info.type), bm = CCV_TENSOR_GET_MEMORY(b->info.type);
		const int da = CCV_TENSOR_GET_DEVICE_ID(a->info.type), db = CCV_TENSOR_GET_DEVICE_ID(b->info.type);
		if (stream_context) {
			hipStream_t stream = stream_of(stream_context);
			if (am == CCV_TENSOR_CPU_MEMORY && bm == CCV_TENSOR_GPU_MEMORY) HIP_ENFORCE(hipMemcpyAsync(b->data.u8, a->data.u8, size, hipMemcpyHostToDevice, stream));
			else if (am == CCV_TENSOR_GPU_MEMORY && bm == CCV_TENSOR_CPU_MEMORY) HIP_ENFORCE(hipMemcpyAsync(b->data.u8, a->data.u8, size, hipMemcpyDeviceToHost, stream));
			else if (am == CCV_TENSOR_CPU_MEMORY && bm == CCV_TENSOR_CPU_MEMORY) HIP_ENFORCE(hipMemcpyAsync(b->data.u8, a->data.u8, size, hipMemcpyHostToHost, stream));
			else if (da == db) HIP_ENFORCE(hipMemcpyAsync(b->data.u8, a->data.u8, size, hipMemcpyDeviceToDevice, stream));
			else HIP_ENFORCE(hipMemcpyPeerAsync(b->data.u8, db, a->data.u8, da, size, stream));
		} else
			nnc_mi355x_memcpy(b->data.u8, b->info.type, a->data.u8, a->info.type, size);
	}
	return CCV_NNC_EXEC_SUCCESS;
}

// ---- DROPOUT (lib/nnc/cmd/dropout/ccv_nnc_dropout_cpu_ref.c:17-147 / :149-275; replaces dropout/gpu/ccv_nnc_dropout_gpu_cudnn.cu)
// outputs (b, mask): mask[i] = 1 when element i is dropped (one byte per element in the reserved-space tensor the host sizes,
// ccv_nnc_dropout.c:21-45), b = mask ? 0 : a / (1 - p); `entirety` draws ONE decision for the whole tensor (int32 in mask[0]).
// The draw is a counter-based hash of (per-call seed, element index): parity with the reference is statistical by
// construction (its tests check the drop rate and the 1/(1-p) scaling, test/int/nnc/cudnn.tests.c:3186-3464).
__host__ __device__ __forceinline__ unsigned mix32(unsigned x)
{
	x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
	return x;
}
// `tick`: null, or (inside a captured step, device_rt.cpp "HIP-graph capture") the word the graph's first node increments -- the seed the host drew while the
// step was being recorded is a kernel argument and would repeat with every replay
__global__ void __launch_bounds__(EW_THREADS) dropout_forw_kernel(const float* a, float* b, unsigned char* mask, const size_t n, const unsigned seed0, const unsigned* tick, const float p, const float inv_p)
{
	const unsigned seed = tick ? seed0 + 0x9e3779b9U * tick[0] : seed0;
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
		const unsigned r = mix32(mix32((unsigned)i ^ seed) + (unsigned)(i >> 32) + 0x9e3779b9U);
		const int drop = (float)(r >> 8) * (1.f / 16777216.f) <= p;
		mask[i] = (unsigned char)drop;
		b[i] = drop ? 0.f : a[i] * inv_p;
	}
}
__global__ void __launch_bounds__(EW_THREADS) dropout_back_kernel(const float* g, float* h, const unsigned char* mask, const size_t n, const float inv_p)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) h[i] = mask[i] ? 0.f : g[i] * inv_p;
}
__global__ void __launch_bounds__(64) dropout_decide_kernel(int* decision, const unsigned seed0, const unsigned* tick, const float p)
{ // the whole-tensor decision of a captured step: made on the device from the replay's tick (outside a capture the host decides and copies the word)
	if (threadIdx.x == 0 && blockIdx.x == 0) decision[0] = (float)(mix32(seed0 + 0x9e3779b9U * tick[0]) >> 8) * (1.f / 16777216.f) <= p;
}
__global__ void __launch_bounds__(EW_THREADS) dropout_entire_kernel(const float* a, float* b, const int* decision, const size_t n, const float inv_p)
{
	const int drop = decision[0];
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) b[i] = drop ? 0.f : a[i] * inv_p;
}

// The host's per-stream generator when the reference host is linked in (lib/nnc/ccv_nnc_stream.c:262), else a process counter.
extern "C" uint32_t ccv_nnc_stream_context_genrand_uint32(ccv_nnc_stream_context_t* const stream_context) __attribute__((weak));
static unsigned dropout_seed(ccv_nnc_stream_context_t* ctx)
{
	if (ccv_nnc_stream_context_genrand_uint32) return ccv_nnc_stream_context_genrand_uint32(ctx);
	static unsigned counter = 0x243f6a88U;
	return __sync_add_and_fetch(&counter, 0x9e3779b9U);
}

static int _dropout_forw(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	if (input_size < 1 || output_size < 2 || !inputs[0] || !outputs[0] || !outputs[1]) return CCV_NNC_EXEC_INVALID;
	const ccv_nnc_tensor_t* a = inputs[0];
	if (CCV_GET_DATA_TYPE(a->info.datatype) != CCV_32F || !tensor_contiguous(a) || !tensor_contiguous(outputs[0])) return CCV_NNC_EXEC_INVALID;
	const size_t n = tensor_count(a->info);
	if (tensor_count(outputs[0]->info) != n) return CCV_NNC_EXEC_INVALID;
	const float p = cmd.info.dropout.p, inv_p = 1.f / (1.f - p);
	const size_t mask_bytes = tensor_count(outputs[1]->info) * datatype_size(outputs[1]->info.datatype);
	hipStream_t stream = stream_of(stream_context);
	const unsigned seed = dropout_seed(stream_context);
	const unsigned* const tick = capture_tick_of(stream);
	if (n == 0) return CCV_NNC_EXEC_SUCCESS;
	if (cmd.info.dropout.entirety) {
		if (mask_bytes < sizeof(int)) return CCV_NNC_EXEC_INVALID;
		const int drop = (float)(mix32(seed) >> 8) * (1.f / 16777216.f) <= p;
		if (tick) hipLaunchKernelGGL(dropout_decide_kernel, dim3(1), dim3(64), 0, stream, outputs[1]->data.i32, seed, tick, p); // (a captured copy node would re-read this stack word at every replay)
		else HIP_ENFORCE(hipMemcpyAsync(outputs[1]->data.u8, &drop, sizeof(int), hipMemcpyHostToDevice, stream)); // pageable source: copied before return
		hipLaunchKernelGGL(dropout_entire_kernel, dim3(grid_for(n, EW_THREADS)), dim3(EW_THREADS), 0, stream, (const float*)a->data.f32, outputs[0]->data.f32, (const int*)outputs[1]->data.i32, n, inv_p);
	} else {
		if (mask_bytes < n) return CCV_NNC_EXEC_INVALID;
		hipLaunchKernelGGL(dropout_forw_kernel, dim3(grid_for(n, EW_THREADS)), dim3(EW_THREADS), 0, stream, (const float*)a->data.f32, outputs[0]->data.f32, outputs[1]->data.u8, n, seed, tick, p, inv_p);
	}
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}
static int _dropout_back(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{ // inputs (g, _, _, _, mask), output h   (dropout_cpu_ref.c:149-)
	if (input_size < 5 || output_size < 1 || !inputs[0] || !inputs[4] || !outputs[0]) return CCV_NNC_EXEC_INVALID;
	const ccv_nnc_tensor_t* g = inputs[0];
	if (CCV_GET_DATA_TYPE(g->info.datatype) != CCV_32F || !tensor_contiguous(g) || !tensor_contiguous(outputs[0])) return CCV_NNC_EXEC_INVALID;
	const size_t n = tensor_count(g->info);
	if (tensor_count(outputs[0]->info) != n) return CCV_NNC_EXEC_INVALID;
	if (n == 0) return CCV_NNC_EXEC_SUCCESS;
	const float inv_p = 1.f / (1.f - cmd.info.dropout.p);
	hipStream_t stream = stream_of(stream_context);
	if (cmd.info.dropout.entirety) hipLaunchKernelGGL(dropout_entire_kernel, dim3(grid_for(n, EW_THREADS)), dim3(EW_THREADS), 0, stream, (const float*)g->data.f32, outputs[0]->data.f32, (const int*)inputs[4]->data.i32, n, inv_p);
	else hipLaunchKernelGGL(dropout_back_kernel, dim3(grid_for(n, EW_THREADS)), dim3(EW_THREADS), 0, stream, (const float*)g->data.f32, outputs[0]->data.f32, (const unsigned char*)inputs[4]->data.u8, n, inv_p);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}

// ---- RANDOM_UNIFORM / RANDOM_NORMAL: parameter initialisation of every cnnp model on the device (ccv_cnnp_model init runs
// CMD_RANDOM_UNIFORM_FORWARD on the weights, lib/nnc/ccv_cnnp_model_addons.c) ---------------------------------------------------------
//   uniform  lib/nnc/cmd/rand/ccv_nnc_rand_uniform_cpu_ref.c:17-33   a = r u + (1 - r) l, r in (0, 1), l = blas.a[0], u = blas.a[1]
//   normal   lib/nnc/cmd/rand/ccv_nnc_rand_normal_cpu_ref.c:17-45    Box-Muller pairs, std = blas.a[0], mean = blas.a[1]
// Same counter-based generator as dropout, seeded per call from the stream's generator: parity with the reference is
// statistical by construction (its tests check mean / range, test/int/nnc/random.tests.c).
__device__ __forceinline__ float unit_open(const unsigned r) { return ((float)(r >> 8) + 0.5f) * (1.f / 16777216.f); } // (0, 1)
__global__ void __launch_bounds__(EW_THREADS) random_uniform_kernel(float* a, const size_t n, const unsigned seed, const float l, const float u)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
		const float r = unit_open(mix32(mix32((unsigned)i ^ seed) + (unsigned)(i >> 32) + 0x9e3779b9U));
		a[i] = r * u + (1.f - r) * l;
	}
}
__global__ void __launch_bounds__(EW_THREADS) random_normal_kernel(float* a, const size_t n, const unsigned seed, const float std, const float mean)
{
	const size_t pairs = (n + 1) / 2, stride = (size_t)gridDim.x * blockDim.x;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += stride) {
		const unsigned h = mix32(mix32((unsigned)i ^ seed) + (unsigned)(i >> 32) + 0x9e3779b9U);
		const float r0 = unit_open(h), r1 = unit_open(mix32(h ^ 0x85ebca6bU));
		const float mag = std * sqrtf(-2.f * logf(r0));
		a[2 * i] = mag * cosf(6.283185307179586f * r1) + mean;
		if (2 * i + 1 < n) a[2 * i + 1] = mag * sinf(6.283185307179586f * r1) + mean;
	}
}
static int _random_exec(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	const bool normal = cmd.cmd == CCV_NNC_RANDOM_NORMAL_FORWARD || cmd.cmd == CCV_NNC_RANDOM_NORMAL_BACKWARD;
	for (int k = 0; k < output_size; k++) {
		ccv_nnc_tensor_t* const a = outputs[k];
		if (!a) continue;
		if (!tensor_contiguous(a) || CCV_GET_DATA_TYPE(a->info.datatype) != CCV_32F) return CCV_NNC_EXEC_INVALID;
		const size_t n = tensor_count(a->info);
		if (n == 0) continue;
		const unsigned seed = dropout_seed(stream_context);
		if (normal) hipLaunchKernelGGL(random_normal_kernel, dim3(grid_for((n + 1) / 2, EW_THREADS)), dim3(EW_THREADS), 0, stream_of(stream_context), a->data.f32, n, seed, cmd.info.blas.a[0], cmd.info.blas.a[1]);
		else hipLaunchKernelGGL(random_uniform_kernel, dim3(grid_for(n, EW_THREADS)), dim3(EW_THREADS), 0, stream_of(stream_context), a->data.f32, n, seed, cmd.info.blas.a[0], cmd.info.blas.a[1]);
		HIP_ENFORCE(hipGetLastError());
	}
	return CCV_NNC_EXEC_SUCCESS;
}

} // namespace

namespace nnc {

bool sgd_is_exec(const exec_fn_t fn) { return fn == _sgd_forw; }
int sgd_forw_multi(const ccv_nnc_cmd_t* const* cmds, ccv_nnc_tensor_t* const* const* ins, ccv_nnc_tensor_t* const* const* outs, int n, ccv_nnc_stream_context_t* ctx) { return sgd_multi_run(cmds, ins, outs, n, ctx); }

int fill_f32(float* p, size_t n, float v, ccv_nnc_stream_context_t* ctx)
{
	OpFill f; f.v = v;
	return ew_map<OpFill, 0>(f, p, 0, 0, 0, n, ctx);
}

int relu_inplace(ccv_nnc_tensor_t* t, ccv_nnc_stream_context_t* ctx)
{
	if (!t || !tensor_contiguous(t)) return CCV_NNC_EXEC_INVALID;
	const int dt = CCV_GET_DATA_TYPE(t->info.datatype);
	if (dt != CCV_32F && dt != CCV_16F) return CCV_NNC_EXEC_INVALID;
	return ew_map_any<OpRelu, 1>(OpRelu(), t->info.datatype, t->data.u8, t->data.u8, 0, 0, tensor_count(t->info), ctx);
}

int relu_back_inplace(ccv_nnc_tensor_t* h, const ccv_nnc_tensor_t* b, ccv_nnc_stream_context_t* ctx)
{
	if (!h || !b || !tensor_contiguous(h) || !tensor_contiguous(b) || h->info.datatype != b->info.datatype || tensor_count(h->info) != tensor_count(b->info)) return CCV_NNC_EXEC_INVALID;
	const int dt = CCV_GET_DATA_TYPE(h->info.datatype);
	if (dt != CCV_32F && dt != CCV_16F) return CCV_NNC_EXEC_INVALID;
	return ew_map_any<OpReluBack, 2>(OpReluBack(), h->info.datatype, h->data.u8, h->data.u8, b->data.u8, 0, tensor_count(h->info), ctx);
}

int colsum_f32(const float* x, long rows, int cols, long ld, float* out, int accumulate, ccv_nnc_stream_context_t* ctx)
{
	if (cols <= 0) return CCV_NNC_EXEC_SUCCESS;
	hipStream_t stream = stream_of(ctx);
	const int col_tiles = (cols + CS_COLS - 1) / CS_COLS;
	long slices = ((long)device_cu_count() * 4 + col_tiles - 1) / col_tiles;
	const long max_slices = (rows + 63) / 64;
	if (slices > max_slices) slices = max_slices;
	if (slices < 1) slices = 1;
	const long rows_per_slice = (rows + slices - 1) / slices;
	slices = rows > 0 ? (rows + rows_per_slice - 1) / rows_per_slice : 1;
	float* partial = (float*)workspace_of(ctx, sizeof(float) * (size_t)slices * cols);
	if (!partial) return CCV_NNC_EXEC_OOM;
	if (cols % 4 == 0 && ld % 4 == 0 && aligned16(x) && aligned16(partial))
		hipLaunchKernelGGL(colsum_partial_v4_kernel, dim3(col_tiles, (unsigned)slices), dim3(256), 0, stream, x, rows, cols, ld, rows_per_slice > 0 ? rows_per_slice : 1, partial);
	else
		hipLaunchKernelGGL(colsum_partial_kernel, dim3(col_tiles, (unsigned)slices), dim3(256), 0, stream, x, rows, cols, ld, rows_per_slice > 0 ? rows_per_slice : 1, partial);
	HIP_ENFORCE(hipGetLastError());
	hipLaunchKernelGGL(colsum_final_kernel, dim3((cols + FOLD_CH - 1) / FOLD_CH), dim3(256), 0, stream, (const float*)partial, (int)slices, cols, out, accumulate);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}

} // namespace nnc

#define NNC_REG(CMD, BACKEND, FORMATS, DATATYPES, MEMORY, EXEC) \
	extern "C" void _register_command_##CMD##_backend_##BACKEND(ccv_nnc_cmd_backend_registry_t* const registry) \
	{ registry->tensor_formats = (FORMATS); registry->tensor_datatypes = (DATATYPES); registry->tensor_memory = (MEMORY); registry->algorithms = 1; registry->exec = EXEC; NNC_HALF_STAGED(registry, EXEC); }
#define ALL_FORMATS (CCV_TENSOR_FORMAT_NCHW | CCV_TENSOR_FORMAT_NHWC | CCV_TENSOR_FORMAT_CHWN)

NNC_REG(CCV_NNC_RELU_FORWARD, CCV_NNC_BACKEND_GPU_CUDNN, ALL_FORMATS, CCV_32F, CCV_TENSOR_GPU_MEMORY, _relu_forw)
NNC_REG(CCV_NNC_RELU_BACKWARD, CCV_NNC_BACKEND_GPU_CUDNN, ALL_FORMATS, CCV_32F, CCV_TENSOR_GPU_MEMORY, _relu_back)
NNC_REG(CCV_NNC_EWSUM_FORWARD, CCV_NNC_BACKEND_GPU_CUDNN, ALL_FORMATS, CCV_32F | CCV_32S, CCV_TENSOR_GPU_MEMORY, _ewsum_forw)
NNC_REG(CCV_NNC_EWSUM_BACKWARD, CCV_NNC_BACKEND_GPU_CUDNN, ALL_FORMATS, CCV_32F, CCV_TENSOR_GPU_MEMORY, _ewsum_back)
NNC_REG(CCV_NNC_SCALAR_MUL_FORWARD, CCV_NNC_BACKEND_GPU_CUDNN, ALL_FORMATS, CCV_32F, CCV_TENSOR_GPU_MEMORY, _scalar_mul_forw)
NNC_REG(CCV_NNC_SCALAR_MUL_BACKWARD, CCV_NNC_BACKEND_GPU_CUDNN, ALL_FORMATS, CCV_32F, CCV_TENSOR_GPU_MEMORY, _scalar_mul_back)
NNC_REG(CCV_NNC_DROPOUT_FORWARD, CCV_NNC_BACKEND_GPU_CUDNN, ALL_FORMATS, CCV_32F, CCV_TENSOR_GPU_MEMORY, _dropout_forw)
NNC_REG(CCV_NNC_DROPOUT_BACKWARD, CCV_NNC_BACKEND_GPU_CUDNN, ALL_FORMATS, CCV_32F, CCV_TENSOR_GPU_MEMORY, _dropout_back)
NNC_REG(CCV_NNC_SGD_FORWARD, CCV_NNC_BACKEND_GPU_REF, ALL_FORMATS, CCV_32F, CCV_TENSOR_GPU_MEMORY, _sgd_forw)
NNC_REG(CCV_NNC_SET_FORWARD, CCV_NNC_BACKEND_GPU_CUDNN, ALL_FORMATS, CCV_64F | CCV_32F | CCV_16F | CCV_32S, CCV_TENSOR_GPU_MEMORY, _set_forw)   // (CCV_16F in the row: handled natively, no fp32 images)
NNC_REG(CCV_NNC_SET_BACKWARD, CCV_NNC_BACKEND_GPU_CUDNN, ALL_FORMATS, CCV_64F | CCV_32F | CCV_16F | CCV_32S, CCV_TENSOR_GPU_MEMORY, _set_back)
NNC_REG(CCV_NNC_DATA_TRANSFER_FORWARD, CCV_NNC_BACKEND_GPU_REF, ALL_FORMATS, CCV_64F | CCV_32F | CCV_16F | CCV_64S | CCV_32S | CCV_8U | CCV_QX, CCV_TENSOR_CPU_MEMORY | CCV_TENSOR_GPU_MEMORY, _data_transfer)
NNC_REG(CCV_NNC_DATA_TRANSFER_BACKWARD, CCV_NNC_BACKEND_GPU_REF, ALL_FORMATS, CCV_64F | CCV_32F | CCV_16F | CCV_64S | CCV_32S | CCV_8U | CCV_QX, CCV_TENSOR_CPU_MEMORY | CCV_TENSOR_GPU_MEMORY, _data_transfer)
NNC_REG(CCV_NNC_RANDOM_UNIFORM_FORWARD, CCV_NNC_BACKEND_GPU_REF, ALL_FORMATS, CCV_32F, CCV_TENSOR_GPU_MEMORY, _random_exec)
NNC_REG(CCV_NNC_RANDOM_UNIFORM_BACKWARD, CCV_NNC_BACKEND_GPU_REF, ALL_FORMATS, CCV_32F, CCV_TENSOR_GPU_MEMORY, _random_exec)
NNC_REG(CCV_NNC_RANDOM_NORMAL_FORWARD, CCV_NNC_BACKEND_GPU_REF, ALL_FORMATS, CCV_32F, CCV_TENSOR_GPU_MEMORY, _random_exec)
NNC_REG(CCV_NNC_RANDOM_NORMAL_BACKWARD, CCV_NNC_BACKEND_GPU_REF, ALL_FORMATS, CCV_32F, CCV_TENSOR_GPU_MEMORY, _random_exec)
