// CCV_16F tensors on rows whose kernels compute in fp32: the command runs on fp32 IMAGES of its half-precision tensors.
//
// The reference's GPU rows register CCV_32F | CCV_16F (e.g. lib/nnc/cmd/convolution/gpu/ccv_nnc_conv_gpu_cudnn.cu:478-500,
// lib/nnc/cmd/sgd/gpu/ccv_nnc_sgd_gpu_ref.cu:13-100 with its mixed fp16 / fp32 variants) and its trainers store activations and
// gradients in half precision (test/int/nnc/cifar.tests.c:473).  Rows with a native half-precision kernel (the contraction core:
// GEMM and convolution on v_mfma_f32_32x32x16_f16, cmd_gemm.cpp / cmd_conv.cpp) handle CCV_16F themselves; every other row
// takes this route, so that a half-precision graph never meets a row that cannot run it:
//   1. every distinct CCV_16F tensor of the command gets an fp32 image in the stream's staging arena (device_rt.cpp; separate
//      from the workspace, which the command underneath may grow and thereby move);
//   2. inputs are converted up (exact); outputs that are accumulated into or alias an input are converted up too;
//   3. the fp32 exec function runs on shadow tensor structs (same shape / strides / view flags, datatype CCV_32F);
//   4. outputs are converted down (round to nearest even, as the reference's ccv_float_to_half_precision does) -- a dense tensor as one
//      run, a VIEW element by element through its strides: what lies between a view's rows is not this command's to write.
// Storage is half precision, arithmetic fp32: at least the accuracy the reference's own half-precision kernels have (they
// accumulate in fp32 as well), so its GPU-vs-CPU tolerances hold.  Cost: one extra read + write of each half tensor -- this is
// the coverage path, not the fast one.
#include "common.h"

namespace nnc {

typedef _Float16 half_t;
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// n elements, contiguous.  Four per thread where the alignment allows (8-byte / 16-byte accesses).
static __global__ void __launch_bounds__(256) half_up_kernel(const half_t* __restrict__ in, float* __restrict__ out, const size_t n, const int vec)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	if (vec) {
		const size_t n4 = n >> 2;
		for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
			const half4_t v = ((const half4_t*)in)[i];
			((f32x4_t*)out)[i] = f32x4_t{ (float)v[0], (float)v[1], (float)v[2], (float)v[3] };
		}
		for (size_t i = (n4 << 2) + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (float)in[i];
	} else
		for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (float)in[i];
}
static __global__ void __launch_bounds__(256) half_down_kernel(const float* __restrict__ in, half_t* __restrict__ out, const size_t n, const int vec)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	if (vec) {
		const size_t n4 = n >> 2;
		for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
			const f32x4_t v = ((const f32x4_t*)in)[i];
			((half4_t*)out)[i] = half4_t{ (half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3] };
		}
		for (size_t i = (n4 << 2) + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (half_t)in[i];
	} else
		for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (half_t)in[i];
}

// A VIEW's elements only, from the fp32 image of its span back to the half tensor: the gaps between a view's rows belong to somebody else (sibling
// views of one parent -- channel-concatenated branches the host may run on other streams) and are never written.  One thread per element.
struct view_geom_t { int nd; int dim[CCV_NNC_MAX_DIM_ALLOC]; int stride[CCV_NNC_MAX_DIM_ALLOC]; };
static __global__ void __launch_bounds__(256) half_down_view_kernel(const float* __restrict__ in, half_t* __restrict__ out, const size_t n, const view_geom_t g)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
		size_t r = i, off = 0;
		for (int d = g.nd - 1; d >= 0; d--) { off += (r % (size_t)g.dim[d]) * (size_t)g.stride[d]; r /= (size_t)g.dim[d]; }
		out[off] = (half_t)in[off];
	}
}
static int float_to_half_view(const float* image, void* half, const ccv_nnc_tensor_t* t, ccv_nnc_stream_context_t* ctx)
{
	view_geom_t g;
	int st[CCV_NNC_MAX_DIM_ALLOC];
	tensor_strides(t, st);
	g.nd = tensor_nd(t->info.dim);
	size_t n = 1;
	for (int i = 0; i < g.nd; i++) { g.dim[i] = t->info.dim[i]; g.stride[i] = st[i]; n *= (size_t)t->info.dim[i]; }
	if (!n || !g.nd) return CCV_NNC_EXEC_SUCCESS;
	hipLaunchKernelGGL(half_down_view_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream_of(ctx), image, (half_t*)half, n, g);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}

int half_to_float(const void* in, float* out, size_t n, ccv_nnc_stream_context_t* ctx)
{
	if (!n) return CCV_NNC_EXEC_SUCCESS;
	const int vec = (((uintptr_t)in & 7) == 0 && ((uintptr_t)out & 15) == 0) ? 1 : 0;
	hipLaunchKernelGGL(half_up_kernel, dim3(grid_for(vec ? (n + 3) / 4 : n, 256)), dim3(256), 0, stream_of(ctx), (const half_t*)in, out, n, vec);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}
int float_to_half(const float* in, void* out, size_t n, ccv_nnc_stream_context_t* ctx)
{
	if (!n) return CCV_NNC_EXEC_SUCCESS;
	const int vec = (((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 7) == 0) ? 1 : 0;
	hipLaunchKernelGGL(half_down_kernel, dim3(grid_for(vec ? (n + 3) / 4 : n, 256)), dim3(256), 0, stream_of(ctx), in, (half_t*)out, n, vec);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}

// Bias gradients of the half-precision contraction commands.
// dbias for half precision: out[c] (+)= sum_r x[r * ld + c], fp32 partial sums over row slices, folded in a fixed order
static __global__ void __launch_bounds__(256) colsum_h_partial_kernel(const half_t* x, const long rows, const int cols, const long ld, const long rows_per_slice, float* partial)
{
	__shared__ float red[4][64];
	const int c = blockIdx.x * 64 + (threadIdx.x & 63), rs = threadIdx.x >> 6;
	const long r0 = (long)blockIdx.y * rows_per_slice, r1 = r0 + rows_per_slice < rows ? r0 + rows_per_slice : rows;
	float s = 0.f;
	if (c < cols) for (long r = r0 + rs; r < r1; r += 4) s += (float)x[r * ld + c];
	red[rs][threadIdx.x & 63] = s;
	__syncthreads();
	if (rs == 0 && c < cols) partial[(long)blockIdx.y * cols + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
static __global__ void __launch_bounds__(256) colsum_h_final_kernel(const float* partial, const int slices, const int cols, half_t* out, const int accumulate)
{
	__shared__ float red[FOLD_PH][FOLD_CH]; // 16 columns x 16 phases per workgroup (common.h)
	const int ch = threadIdx.x & (FOLD_CH - 1), phase = threadIdx.x / FOLD_CH;
	const int c = blockIdx.x * FOLD_CH + ch;
	red[phase][ch] = c < cols ? fold_slices(partial, slices, cols, c, phase) : 0.f;
	__syncthreads();
	if (phase == 0 && c < cols) {
		const float s = fold_phases(red, ch);
		out[c] = (half_t)(accumulate ? (float)out[c] + s : s);
	}
}
int colsum_f16(const void* xv, long rows, int cols, long ld, void* outv, int accumulate, ccv_nnc_stream_context_t* ctx)
{
	if (cols <= 0) return CCV_NNC_EXEC_SUCCESS;
	const half_t* x = (const half_t*)xv;
	half_t* out = (half_t*)outv;
	const int col_tiles = (cols + 63) / 64;
	long slices = ((long)device_cu_count() * 4 + col_tiles - 1) / col_tiles;
	const long max_slices = (rows + 63) / 64;
	if (slices > max_slices) slices = max_slices;
	if (slices < 1) slices = 1;
	const long rows_per_slice = (rows + slices - 1) / slices;
	slices = rows > 0 ? (rows + rows_per_slice - 1) / rows_per_slice : 1;
	float* partial = (float*)workspace_of(ctx, sizeof(float) * (size_t)slices * cols);
	if (!partial) return CCV_NNC_EXEC_OOM;
	hipStream_t stream = stream_of(ctx);
	hipLaunchKernelGGL(colsum_h_partial_kernel, dim3(col_tiles, (unsigned)slices), dim3(256), 0, stream, x, rows, cols, ld, rows_per_slice > 0 ? rows_per_slice : 1, partial);
	HIP_ENFORCE(hipGetLastError());
	hipLaunchKernelGGL(colsum_h_final_kernel, dim3((cols + FOLD_CH - 1) / FOLD_CH), dim3(256), 0, stream, (const float*)partial, (int)slices, cols, out, accumulate);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}

// Many slices (one per image and 64-pixel tile: thousands) are first folded in groups -- a workgroup per (64 columns, group of slices), four slices in flight per
// column, coalesced rows -- so that the final kernel's per-column chains stay a few loads long.  Fixed order throughout: deterministic.
static __global__ void __launch_bounds__(256) partials_group_kernel(const float* __restrict__ in, const long slices, const int cols, const int group, float* __restrict__ out)
{
	__shared__ float red[4][64];
	const int cl = threadIdx.x & 63, ph = threadIdx.x >> 6;
	const int c = blockIdx.x * 64 + cl;
	const long s0 = (long)blockIdx.y * group, s1 = s0 + group < slices ? s0 + group : slices;
	float a = 0.f, b = 0.f;
	if (c < cols) {
		long i = s0 + ph;
		for (; i + 4 < s1; i += 8) { a += in[i * cols + c]; b += in[(i + 4) * cols + c]; }
		if (i < s1) a += in[i * cols + c];
	}
	red[ph][cl] = a + b;
	__syncthreads();
	if (ph == 0 && c < cols) out[(long)blockIdx.y * cols + c] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}
int colsum_partials_f16(const float* partial, long slices, const int cols, void* out, const int accumulate, ccv_nnc_stream_context_t* ctx)
{
	if (cols <= 0) return CCV_NNC_EXEC_SUCCESS;
	if (slices > 0x7fffffffL) return CCV_NNC_EXEC_INVALID;
	hipStream_t stream = stream_of(ctx);
	if (slices > 512) { // (the grouped sums go to a second area right behind the partials)
		const int group = (int)((slices + 255) / 256 < 16 ? 16 : (slices + 255) / 256);
		const long groups = (slices + group - 1) / group;
		float* const folded = (float*)partial + (size_t)slices * cols;
		hipLaunchKernelGGL(partials_group_kernel, dim3((cols + 63) / 64, (unsigned)groups), dim3(256), 0, stream, partial, slices, cols, group, folded);
		HIP_ENFORCE(hipGetLastError());
		partial = folded;
		slices = groups;
	}
	hipLaunchKernelGGL(colsum_h_final_kernel, dim3((cols + FOLD_CH - 1) / FOLD_CH), dim3(256), 0, stream, partial, (int)slices, cols, (half_t*)out, accumulate);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}

// out[c] (+)= sum over (o, i) of x[(o * C + c) * inner + i] for halves (bias gradient of an NCHW convolution): one workgroup per
// plane, fp32 partials in the workspace, folded per channel in a fixed order.
static __global__ void __launch_bounds__(256) plane_sum_h_kernel(const half_t* __restrict__ x, const long planes, const long inner, float* __restrict__ partial)
{ // a WAVE per plane (8 halves per 16-byte load when the plane allows): a workgroup per plane left 7 x 7 planes with 49 busy threads
	typedef half_t h8 __attribute__((ext_vector_type(8)));
	const int lane = threadIdx.x & 63;
	const long nw = (long)gridDim.x * 4;
	for (long pl = (long)blockIdx.x * 4 + (threadIdx.x >> 6); pl < planes; pl += nw) {
		const half_t* const p = x + pl * inner;
		float s = 0.f;
		if ((inner & 7) == 0 && (((uintptr_t)p) & 15) == 0) {
			for (long i = lane; i < (inner >> 3); i += 64) {
				const h8 v = ((const h8*)p)[i];
#pragma unroll
				for (int e = 0; e < 8; e++) s += (float)v[e];
			}
		} else
			for (long i = lane; i < inner; i += 64) s += (float)p[i];
		for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
		if (lane == 0) partial[pl] = s;
	}
}
int chan_sum_planes_f16(const void* x, long outer, int C, long inner, void* out, int accumulate, ccv_nnc_stream_context_t* ctx)
{
	if (C <= 0 || outer <= 0) return CCV_NNC_EXEC_SUCCESS;
	float* partial = (float*)workspace_of(ctx, sizeof(float) * (size_t)outer * C);
	if (!partial) return CCV_NNC_EXEC_OOM;
	hipStream_t stream = stream_of(ctx);
	const long planes = outer * C, want = (planes + 3) / 4;
	hipLaunchKernelGGL(plane_sum_h_kernel, dim3((unsigned)(want < 0x7fffffffL ? want : 0x7fffffffL)), dim3(256), 0, stream, (const half_t*)x, planes, inner, partial);
	HIP_ENFORCE(hipGetLastError());
	hipLaunchKernelGGL(colsum_h_final_kernel, dim3((C + FOLD_CH - 1) / FOLD_CH), dim3(256), 0, stream, (const float*)partial, (int)outer, C, (half_t*)out, accumulate);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}

bool any_half_tensor(ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size)
{
	for (int i = 0; i < input_size; i++)
		if (inputs[i] && CCV_GET_DATA_TYPE(inputs[i]->info.datatype) == CCV_16F) return true;
	for (int i = 0; i < output_size; i++)
		if (outputs[i] && CCV_GET_DATA_TYPE(outputs[i]->info.datatype) == CCV_16F) return true;
	return false;
}

namespace {

// elements from the tensor's first element to one past its last (views: by their strides)
size_t tensor_span(const ccv_nnc_tensor_t* t)
{
	if (!CCV_IS_TENSOR_VIEW(t)) return tensor_count(t->info);
	int st[CCV_NNC_MAX_DIM_ALLOC];
	tensor_strides(t, st);
	const int nd = tensor_nd(t->info.dim);
	if (nd == 0) return 0;
	size_t last = 0;
	for (int i = 0; i < nd; i++) {
		if (t->info.dim[i] <= 0) return 0;
		last += (size_t)(t->info.dim[i] - 1) * (size_t)st[i];
	}
	return last + 1;
}

struct staged_t {
	void* half;   // the tensor's own (half-precision) memory
	size_t span;  // elements
	float* image; // its fp32 image in the arena
	bool is_output, is_input, load;
};

constexpr int MAX_STAGED = 64;

} // namespace

// The reference host's graph runner does not look at what a command returns (lib/nnc/ccv_nnc_graph_run.c calls ccv_nnc_cmd_exec and
// moves on): a refused command would leave its outputs unwritten in silence.  Say so on stderr, a few times per command id.
void warn_refused(const uint32_t cmd, const int ret)
{
	static uint32_t seen[32];
	static int counts[32], nseen = 0;
	if (ret == CCV_NNC_EXEC_SUCCESS) return;
	int i;
	for (i = 0; i < nseen; i++) if (seen[i] == cmd) break;
	if (i == nseen) { if (nseen == 32) return; seen[nseen] = cmd; counts[nseen++] = 0; }
	if (counts[i]++ < 3) fprintf(stderr, "[nnc-mi355x] command 0x%08x refused with %d (INVALID -1 / NO_KERNEL -2 / OOM -3): its outputs were NOT written\n", cmd, ret);
}

// Rows whose kernels read and write their LARGE tensors in half precision themselves (loads / stores of halves, fp32 arithmetic:
// cmd_ew.cpp, cmd_norm.cpp, cmd_pool.cpp).  Bit i of `in` / `out` = that input / output stays in its own memory when every tensor
// named by the masks is a dense CCV_16F tensor; the row's small tensors (batch-norm statistics, ...) still get fp32 images.
static long g_half_staged = 0, g_half_native = 0; // nnc_mi355x_debug_half_counts (test hook; not synchronised: counts, not control)
// NNC_MI355X_HALF_STATS=1: one line per command at unload -- which rows of a run went through fp32 images of their half tensors (and how many tensors)
static struct half_stats_t {
	struct { uint32_t cmd; long calls, tensors; } row[64];
	int n = 0, on = -1;
	~half_stats_t()
	{
		if (on != 1) return;
		for (int i = 0; i < n; i++) fprintf(stderr, "[nnc_mi355x] half tensors staged as fp32 images: %-48s %6ld calls, %6ld tensors\n", command_row_name(row[i].cmd), row[i].calls, row[i].tensors);
	}
} g_half_stats;
static void half_stats_note(const uint32_t cmd, const int tensors)
{
	if (g_half_stats.on < 0) { const char* e = getenv("NNC_MI355X_HALF_STATS"); g_half_stats.on = (e && *e == '1') ? 1 : 0; }
	if (g_half_stats.on != 1 || !tensors) return;
	for (int i = 0; i < g_half_stats.n; i++) if (g_half_stats.row[i].cmd == cmd) { g_half_stats.row[i].calls++; g_half_stats.row[i].tensors += tensors; return; }
	if (g_half_stats.n < 64) { g_half_stats.row[g_half_stats.n].cmd = cmd; g_half_stats.row[g_half_stats.n].calls = 1; g_half_stats.row[g_half_stats.n].tensors = tensors; g_half_stats.n++; }
}
struct native_half_t { uint32_t cmd; unsigned in, out; };
static const native_half_t g_native_half[] = {
	{ CCV_NNC_RELU_FORWARD, 1u << 0, 1u << 0 },
	{ CCV_NNC_RELU_BACKWARD, (1u << 0) | (1u << 2), 1u << 0 },            // g, (a unused), b -> h
	{ CCV_NNC_EWSUM_FORWARD, ~0u, 1u << 0 },
	{ CCV_NNC_EWSUM_BACKWARD, 1u << 0, ~0u },
	{ CCV_NNC_BATCH_NORM_FORWARD, 1u << 0, 1u << 0 },                     // x -> y
	{ CCV_NNC_BATCH_NORM_BACKWARD, (1u << 0) | (1u << 5), 1u << 0 },      // g, x -> h
	{ CCV_NNC_MAX_POOL_FORWARD, 1u << 0, 1u << 0 },
	{ CCV_NNC_MAX_POOL_BACKWARD, (1u << 0) | (1u << 1) | (1u << 2), 1u << 0 }, // g, x, y -> h
	{ CCV_NNC_AVERAGE_POOL_FORWARD, 1u << 0, 1u << 0 },
	{ CCV_NNC_AVERAGE_POOL_BACKWARD, 1u << 0, 1u << 0 },
	{ CCV_NNC_SGD_FORWARD, (1u << 0) | (1u << 1) | (1u << 2), (1u << 0) | (1u << 1) }, // g, a, m -> b, n
	{ CCV_NNC_SOFTMAX_CROSSENTROPY_FORWARD, 1u << 0, 1u << 1 },           // logits -> softmax (labels and the loss: fp32 images, a value per row)
	{ CCV_NNC_SOFTMAX_CROSSENTROPY_BACKWARD, 1u << 5, 1u << 0 },          // softmax -> h
};
static const native_half_t* native_half_row(const uint32_t cmd, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size)
{
	if (flags & CCV_NNC_ACCUMULATE_OUTPUT) return 0;
	for (const native_half_t& r : g_native_half) {
		if (r.cmd != cmd) continue;
		int seen = 0;
		for (int i = 0; i < input_size && i < 32; i++)
			if (((r.in >> i) & 1) && inputs[i]) { if (CCV_GET_DATA_TYPE(inputs[i]->info.datatype) != CCV_16F || CCV_IS_TENSOR_VIEW(inputs[i])) return 0; seen++; }
		for (int i = 0; i < output_size && i < 32; i++)
			if (((r.out >> i) & 1) && outputs[i]) { if (CCV_GET_DATA_TYPE(outputs[i]->info.datatype) != CCV_16F || CCV_IS_TENSOR_VIEW(outputs[i])) return 0; seen++; }
		return seen ? &r : 0;
	}
	return 0;
}

int half_staged_exec(const nnc_exec_f inner, const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const ctx)
{
	MarkerScope marker(cmd.cmd); // every row registered through NNC_HALF_STAGED passes here: the per-command roctx range
	if (!any_half_tensor(inputs, input_size, outputs, output_size)) {
		const int r = inner(cmd, hint, flags, inputs, input_size, outputs, output_size, ctx);
		warn_refused(cmd.cmd, r);
		return r;
	}
	if (input_size + output_size > MAX_STAGED) return CCV_NNC_EXEC_INVALID;
	staged_t st[MAX_STAGED];
	int nst = 0;
	// shadow tensor structs: a view keeps its stride block, so the larger struct is copied for every tensor
	ccv_nnc_tensor_view_t shadow[MAX_STAGED];
	ccv_nnc_tensor_t* in_s[MAX_STAGED];
	ccv_nnc_tensor_t* out_s[MAX_STAGED];
	int which[MAX_STAGED]; // staged entry of shadow i, -1 = the tensor itself is passed on
	size_t total = 0;
	auto visit = [&](ccv_nnc_tensor_t* t, const bool is_output, const int slot) {
		which[slot] = -1;
		if (!t || CCV_GET_DATA_TYPE(t->info.datatype) != CCV_16F) return;
		const size_t span = tensor_span(t);
		int e = -1;
		for (int i = 0; i < nst; i++)
			if (st[i].half == (void*)t->data.u8) { e = i; break; } // the same memory through two tensor structs (in-place commands): one image
		if (e < 0) {
			e = nst++;
			st[e].half = t->data.u8; st[e].span = span; st[e].image = 0; st[e].is_output = st[e].is_input = st[e].load = false;
		} else if (span > st[e].span) st[e].span = span;
		if (is_output) {
			st[e].is_output = true;
			// the old value under accumulation must be in the image (a view's gaps need not: they are never written back)
			if (flags & CCV_NNC_ACCUMULATE_OUTPUT) st[e].load = true;
		} else { st[e].is_input = true; st[e].load = true; }
		which[slot] = e;
	};
	// opaque tensors keep their own memory: the dropout mask is a byte buffer the host merely SIZES through a tensor of the data's
	// type (ccv_nnc_dropout.c:21-45) -- output 1 of the forward command, input 4 of the backward one
	// The LSTM's reserved space likewise (output 3 forward, input 12 backward): ccv_nnc_lstm.c:55-61 sizes it as `bytes` ELEMENTS of the data's type, so a CCV_16F
	// one spans at least the bytes fp32 planes need and the kernels keep their tape (gates, tanh(c), cell states, dropout scales) in fp32 inside it -- nothing is
	// rounded to half between the forward and the backward command, no cell state can overflow the half range, and the space is not converted twice per call
	const int opaque_in = cmd.cmd == CCV_NNC_DROPOUT_BACKWARD ? 4 : (cmd.cmd == CCV_NNC_LSTM_BACKWARD ? 12 : -1), opaque_out = cmd.cmd == CCV_NNC_DROPOUT_FORWARD ? 1 : (cmd.cmd == CCV_NNC_LSTM_FORWARD ? 3 : -1);
	const native_half_t* const native = native_half_row(cmd.cmd, flags, inputs, input_size, outputs, output_size);
	for (int i = 0; i < input_size; i++) { if (i == opaque_in || (native && i < 32 && ((native->in >> i) & 1))) which[i] = -1; else visit(inputs[i], false, i); }
	for (int i = 0; i < output_size; i++) { if (i == opaque_out || (native && i < 32 && ((native->out >> i) & 1))) which[input_size + i] = -1; else visit(outputs[i], true, input_size + i); }
	for (int i = 0; i < nst; i++) total += (st[i].span * sizeof(float) + 255) & ~(size_t)255;
	g_half_staged += nst;
	half_stats_note(cmd.cmd, nst);
	if (native)
		for (int i = 0; i < input_size + output_size; i++) {
			ccv_nnc_tensor_t* const t = i < input_size ? inputs[i] : outputs[i - input_size];
			const int k = i < input_size ? i : i - input_size;
			if (t && k < 32 && (((i < input_size ? native->in : native->out) >> k) & 1)) g_half_native++;
		}
	char* arena = (char*)nnc_staging_of(ctx, total);
	if (total && !arena) return CCV_NNC_EXEC_OOM;
	size_t off = 0;
	for (int i = 0; i < nst; i++) {
		st[i].image = (float*)(arena + off);
		off += (st[i].span * sizeof(float) + 255) & ~(size_t)255;
		if (st[i].load) half_to_float(st[i].half, st[i].image, st[i].span, ctx);
	}
	auto make_shadow = [&](ccv_nnc_tensor_t* t, const int slot) -> ccv_nnc_tensor_t* {
		if (!t || which[slot] < 0) return t;
		if (CCV_IS_TENSOR_VIEW(t)) memcpy(&shadow[slot], t, sizeof(ccv_nnc_tensor_view_t));
		else { memset(&shadow[slot], 0, sizeof(ccv_nnc_tensor_view_t)); memcpy(&shadow[slot], t, sizeof(ccv_nnc_tensor_t)); }
		ccv_nnc_tensor_t* s = (ccv_nnc_tensor_t*)&shadow[slot];
		s->info.datatype = CCV_32F;
		s->data.f32 = st[which[slot]].image;
		s->dataof = 0;
		s->data_size = 0;
		s->alias_ref = 0;
		return s;
	};
	for (int i = 0; i < input_size; i++) in_s[i] = make_shadow(inputs[i], i);
	for (int i = 0; i < output_size; i++) out_s[i] = make_shadow(outputs[i], input_size + i);
	// With staged tensors `inner` must run on the spot: its shadow tensors point into the arena and its outputs are converted back right below.  The look-ahead
	// (peephole.cpp) would record a batch norm / convolution here; a recorded command launched by a flush has no caller to report a failure to, and this
	// function went on to convert the never-written image down to half and return SUCCESS.  So: no recording underneath.
	if (nst) deferred_suppress(1);
	const int ret = inner(cmd, hint, flags, in_s, input_size, out_s, output_size, ctx);
	if (nst) deferred_suppress(-1);
	warn_refused(cmd.cmd, ret);
	if (ret != CCV_NNC_EXEC_SUCCESS) return ret;
	// per OUTPUT tensor, not per image: two views of one parent share an image and each writes its own elements
	for (int i = 0; i < output_size; i++) {
		const int e = which[input_size + i];
		if (e < 0 || !outputs[i]) continue;
		bool done = false;
		for (int j = 0; j < i && !done; j++) done = which[input_size + j] == e && outputs[j] && outputs[j]->data.u8 == outputs[i]->data.u8 && !CCV_IS_TENSOR_VIEW(outputs[j]) && !CCV_IS_TENSOR_VIEW(outputs[i]);
		if (done) continue;
		if (CCV_IS_TENSOR_VIEW(outputs[i]) && !tensor_contiguous(outputs[i])) float_to_half_view(st[e].image, st[e].half, outputs[i], ctx);
		else float_to_half(st[e].image, st[e].half, tensor_count(outputs[i]->info), ctx);
	}
	return CCV_NNC_EXEC_SUCCESS;
}

} // namespace nnc

extern "C" void nnc_mi355x_debug_half_counts(long* staged, long* native)
{
	if (staged) *staged = nnc::g_half_staged;
	if (native) *native = nnc::g_half_native;
}
