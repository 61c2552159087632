// CCV_NNC_BATCH_NORM_FORWARD / BACKWARD on gfx950.  HBM-bound: forward (training) = two reads + one write of x
// (algorithmic 2|x|: the second read of x is the price of the reference's mean -> centred-variance order, which parity
// requires); backward = 2 reads of (x, g) + 1 write of h.
// Oracle semantics: lib/nnc/cmd/norm/ccv_nnc_batch_norm_cpu_ref.c:16-297 (forward: train :44-232, test :233-) and :300-470
// (backward); I/O contract lib/nnc/cmd/norm/ccv_nnc_norm.c:5-82 (running mean / var are updated IN PLACE).
// Replaces norm/gpu/ccv_nnc_batch_norm_gpu_cudnn.cu:13-100.
//
// x is viewed as [outer][C][inner] around the one axis the statistics keep (NHWC: inner = 1, outer = N*H*W; NCHW:
// outer = N, inner = H*W).  Per-channel reductions run in two deterministic stages (slice partials in the stream
// workspace, fixed-order fold), wavefront-coalesced in whichever of C / inner is contiguous.
#include "common.h"
#include "isa.h"

using namespace nnc;

namespace {

struct chan_view_t { long outer; int C; long inner; };

// which reduction: value contributed by element (x, g) of channel c
struct RSum { __device__ float operator()(float x, float, int) const { return x; } };
struct RCenteredSq { const float* mean; __device__ float operator()(float x, float, int c) const { const float w = x - mean[c]; return w * w; } };
struct RXhatG { const float* mean; const float* inv_std; __device__ float operator()(float x, float g, int c) const { return (x - mean[c]) * inv_std[c] * g; } };

constexpr int RC_COLS = 64, RC_PHASES = 4;
// inner == 1: rows of C contiguous channels.  grid (ceil(C/64), slices); lanes = 64 consecutive channels.
static unsigned plane_grid(const long planes);
template <class F, bool USE_G, class T>
__global__ void __launch_bounds__(256) chan_reduce_rows_kernel(F f, const T* x, const T* g, const long rows, const int C, const long rows_per_slice, float* partial)
{
	__shared__ float red[RC_PHASES][RC_COLS];
	const int lane = threadIdx.x & 63, phase = threadIdx.x >> 6;
	const int c = blockIdx.x * RC_COLS + lane;
	const long r0 = (long)blockIdx.y * rows_per_slice;
	long r1 = r0 + rows_per_slice;
	if (r1 > rows) r1 = rows;
	float s = 0.f;
	if (c < C)
		for (long r = r0 + phase; r < r1; r += RC_PHASES) s += f((float)x[r * C + c], USE_G ? (float)g[r * C + c] : 0.f, c);
	red[phase][lane] = s;
	__syncthreads();
	if (phase == 0 && c < C) partial[(long)blockIdx.y * C + c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}
// inner > 1: planes of `inner` contiguous elements, ONE WAVE PER PLANE (16-byte lanes when the plane allows), partial[o][c].
typedef _Float16 half_t;
template <class T> struct pack16 { typedef T type __attribute__((ext_vector_type(16 / sizeof(T)))); }; // one 16-byte access: 4 floats / 8 halves
template <class F, bool USE_G, class T>
__global__ void __launch_bounds__(256) chan_reduce_planes_kernel(F f, const T* x, const T* g, const int C, const long inner, const long planes, float* partial)
{
	constexpr int W = 16 / sizeof(T);
	typedef typename pack16<T>::type V;
	const int lane = threadIdx.x & 63;
	const long nw = (long)gridDim.x * 4;
	for (long pl = (long)blockIdx.x * 4 + (threadIdx.x >> 6); pl < planes; pl += nw) {
		const int c = (int)(pl % C);
		const T* const xp = x + pl * inner;
		const T* const gp = USE_G ? g + pl * inner : x;
		float s = 0.f;
		if ((inner % W) == 0 && ((((uintptr_t)xp) | ((uintptr_t)gp)) & 15) == 0) {
			const long nv = inner / W;
			for (long i = lane; i < nv; i += 64) {
				const V xv = ((const V*)xp)[i];
				V gv = xv;
				if (USE_G) gv = ((const V*)gp)[i];
#pragma unroll
				for (int e = 0; e < W; e++) s += f((float)xv[e], USE_G ? (float)gv[e] : 0.f, c);
			}
		} else
			for (long i = lane; i < inner; i += 64) s += f((float)xp[i], USE_G ? (float)gp[i] : 0.f, c);
		for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
		if (lane == 0) partial[pl] = s;
	}
}
// Folding per-slice (per-plane) partials into per-channel sums, fixed order.  A workgroup is 16 channels x 16 phases: thread
// (phase = t >> 4, channel = t & 15) adds the slices phase, phase + 16, ... with four independent running sums (four loads in
// flight; one thread per channel walking every slice with one dependent add chain was 66 us per call, 64 channels x 4 phases
// still 32 - 68 us on the DawnNet / ResNet-50 steps -- as long as the sweeps over the tensors these folds finish); the 16
// phases meet in LDS.  FOLD_CH channels per workgroup also means 4x the workgroups of the 64-channel form.
// out0[c] (+)= sum_i p0[i][c]  and, when p1 is given, out1[c] (+)= sum_i p1[i][c]  (blockIdx.y picks the array)
__global__ void __launch_bounds__(256) chan_fold_kernel(const float* p0, const float* p1, const long slices, const int C, float* out0, float* out1, const int accumulate)
{
	__shared__ float red[FOLD_PH][FOLD_CH];
	const int ch = threadIdx.x & (FOLD_CH - 1), phase = threadIdx.x / FOLD_CH;
	const int c = blockIdx.x * FOLD_CH + ch;
	const float* const p = blockIdx.y ? p1 : p0;
	float* const out = blockIdx.y ? out1 : out0;
	red[phase][ch] = c < C ? fold_slices(p, slices, C, c, phase) : 0.f;
	__syncthreads();
	if (phase == 0 && c < C) {
		const float v = fold_phases(red, ch);
		out[c] = accumulate ? out[c] + v : v;
	}
}

template <class F, bool USE_G, class T = float>
static int chan_reduce(F f, const T* x, const T* g, const chan_view_t& v, float* out, ccv_nnc_stream_context_t* ctx, const int accumulate = 0)
{
	hipStream_t stream = stream_of(ctx);
	long slices;
	float* partial;
	if (v.inner == 1) {
		const int col_tiles = (v.C + RC_COLS - 1) / RC_COLS;
		slices = ((long)device_cu_count() * 4 + col_tiles - 1) / col_tiles;
		const long max_slices = (v.outer + 63) / 64;
		if (slices > max_slices) slices = max_slices;
		if (slices < 1) slices = 1;
		const long rps = (v.outer + slices - 1) / slices;
		slices = v.outer > 0 ? (v.outer + rps - 1) / rps : 1;
		partial = (float*)workspace_of(ctx, sizeof(float) * (size_t)slices * v.C);
		if (!partial) return CCV_NNC_EXEC_OOM;
		hipLaunchKernelGGL(HIP_KERNEL_NAME(chan_reduce_rows_kernel<F, USE_G, T>), dim3(col_tiles, (unsigned)slices), dim3(256), 0, stream, f, x, g, v.outer, v.C, rps > 0 ? rps : 1, partial);
	} else {
		slices = v.outer;
		partial = (float*)workspace_of(ctx, sizeof(float) * (size_t)slices * v.C);
		if (!partial) return CCV_NNC_EXEC_OOM;
		const long planes = v.outer * v.C;
		hipLaunchKernelGGL(HIP_KERNEL_NAME(chan_reduce_planes_kernel<F, USE_G, T>), dim3(plane_grid(planes)), dim3(256), 0, stream, f, x, g, v.C, v.inner, planes, partial);
	}
	HIP_ENFORCE(hipGetLastError());
	hipLaunchKernelGGL(chan_fold_kernel, dim3((v.C + FOLD_CH - 1) / FOLD_CH), dim3(256), 0, stream, (const float*)partial, (const float*)0, slices, v.C, out, (float*)0, accumulate);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}

// ---- NCHW-style layouts (inner > 1): ONE WAVE PER PLANE -----------------------------------------------------------------------
// The ResNet trainer's tensors (N x C x H x W): a channel's elements are N planes of H * W contiguous floats.  A wave walks
// whole planes (16-byte lanes when the plane size allows), so planes of every size keep all 64 lanes busy -- a workgroup per
// plane left 7 x 7 planes with 49 of 256 threads working and launched half a million workgroups per reduction.
//   forward statistics: per plane, sum and -- in a second sweep over the SAME plane, which the first one has just pulled
//     through the cache hierarchy (12 .. 50 KB) -- the centred sum of squares about the PLANE's mean; the per-channel
//     fold combines the N planes exactly (Chan et al.: M2 = sum_o [M2_o + n (mean_o - mean)^2]).  x comes from HBM once
//     where the two-reduction form read it twice; accuracy is that of the reference's mean -> centred-variance order.
//   backward statistics: sum of g and sum of xhat * g in one sweep over (x, g).
template <int G, class T, class OP>
__device__ __forceinline__ void plane_sweep(const T* __restrict__ p, const long inner, const int lane, OP op)
{
	constexpr int W = 16 / sizeof(T);
	typedef typename pack16<T>::type V;
	if ((inner % W) == 0 && (((uintptr_t)p) & 15) == 0) {
		const V* const pv = (const V*)p;
		const long nv = inner / W;
		for (long i = lane; i < nv; i += G) {
			const V v = pv[i];
#pragma unroll
			for (int e = 0; e < W; e++) op((float)v[e], i * W + e);
		}
	} else
		for (long i = lane; i < inner; i += G) op((float)p[i], i);
}
__device__ __forceinline__ float wave_sum(float s)
{
	for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64); // butterfly: every lane ends with the same total
	return s;
}
// G lanes per plane: 64 (a wave per plane) or 16 (four planes per wave: the 14 x 14 and 7 x 7 planes of a ResNet are 784 / 196 bytes -- a whole wave per plane
// spends its time on the loop overhead and the reduction, not on its 1 - 4 loads; by layer size the one-wave form ran 5 TB/s at 56 x 56 and ~0.5 at 7 x 7)
template <int G> __device__ __forceinline__ float group_sum(float s)
{
	for (int off = G / 2; off > 0; off >>= 1) s += __shfl_xor(s, off, 64); // butterfly inside the G-lane group
	return s;
}
#define PLANE_LOOP(G) \
	const int lane = threadIdx.x & (G - 1); \
	const long nw = (long)gridDim.x * 4 * (64 / G); \
	for (long pl = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 / G) + ((threadIdx.x & 63) / G); pl < planes; pl += nw)
template <class T, int G>
__global__ void __launch_bounds__(256) bn_plane_stats_kernel(const T* __restrict__ x, const long planes, const long inner, float* __restrict__ psum, float* __restrict__ pm2)
{
	PLANE_LOOP(G) {
		const T* const p = x + pl * inner;
		float s = 0.f;
		plane_sweep<G>(p, inner, lane, [&](const float v, long) { s += v; });
		s = group_sum<G>(s);
		const float m = s / (float)inner;
		float q = 0.f;
		plane_sweep<G>(p, inner, lane, [&](const float v, long) { const float d = v - m; q += d * d; });
		q = group_sum<G>(q);
		if (lane == 0) { psum[pl] = s; pm2[pl] = q; }
	}
}
// The same statistics with the plane held in registers between the two reductions: every load of the plane is issued before the first use (NV 16-byte
// chunks per lane), and the centred second moment needs no second sweep through L2 -- the two-sweep form above ran 2.8 TB/s on ResNet-50's 56 x 56 planes.
// For planes of at most G * NV chunks (the host checks, with the alignment); the per-lane order of both sums is the two-sweep kernel's.
template <class T, int G, int NV>
__global__ void __launch_bounds__(256) bn_plane_stats_reg_kernel(const T* __restrict__ x, const long planes, const long inner, float* __restrict__ psum, float* __restrict__ pm2)
{
	constexpr int W = 16 / sizeof(T);
	typedef typename pack16<T>::type V;
	const int nv = (int)(inner / W);
	PLANE_LOOP(G) {
		const V* const pv = (const V*)(x + pl * inner);
		V v[NV];
#pragma unroll
		for (int j = 0; j < NV; j++) { const int i = lane + G * j; v[j] = pv[i < nv ? i : lane < nv ? lane : 0]; }
		float s = 0.f;
#pragma unroll
		for (int j = 0; j < NV; j++)
			if (lane + G * j < nv) {
#pragma unroll
				for (int e = 0; e < W; e++) s += (float)v[j][e];
			}
		s = group_sum<G>(s);
		const float m = s / (float)inner;
		float q = 0.f;
#pragma unroll
		for (int j = 0; j < NV; j++)
			if (lane + G * j < nv) {
#pragma unroll
				for (int e = 0; e < W; e++) { const float d = (float)v[j][e] - m; q += d * d; }
			}
		q = group_sum<G>(q);
		if (lane == 0) { psum[pl] = s; pm2[pl] = q; }
	}
}
// per channel: fold the planes (fixed order), then everything bn_mean_kernel + bn_var_kernel do.  16 channels x 16 phases per
// workgroup like chan_fold_kernel.
__global__ void __launch_bounds__(256) bn_stats_fold_kernel(const float* __restrict__ psum, const float* __restrict__ pm2, const long outer, const int C, const float inner, float* saved_mean, float* saved_inv_std, float* mean, float* var, const float* scale, const float* bias, float* nscale, float* nbias, const float inv_b, const float mom, const float eps)
{
	__shared__ float red[FOLD_PH][FOLD_CH];
	__shared__ float mu_s[FOLD_CH];
	const int ch = threadIdx.x & (FOLD_CH - 1), phase = threadIdx.x / FOLD_CH;
	const int c = blockIdx.x * FOLD_CH + ch;
	red[phase][ch] = c < C ? fold_slices(psum, outer, C, c, phase) : 0.f;
	__syncthreads();
	if (phase == 0) mu_s[ch] = inv_b * fold_phases(red, ch);
	__syncthreads();
	const float mu = mu_s[ch], inv_inner = 1.f / inner;
	float q0 = 0.f, q1 = 0.f;
	if (c < C) {
		long o = phase;
		for (; o + FOLD_PH < outer; o += 2 * FOLD_PH) {
			const float d0 = psum[o * C + c] * inv_inner - mu, d1 = psum[(o + FOLD_PH) * C + c] * inv_inner - mu;
			q0 += pm2[o * C + c] + inner * d0 * d0;
			q1 += pm2[(o + FOLD_PH) * C + c] + inner * d1 * d1;
		}
		for (; o < outer; o += FOLD_PH) { const float d = psum[o * C + c] * inv_inner - mu; q0 += pm2[o * C + c] + inner * d * d; }
	}
	red[phase][ch] = q0 + q1;
	__syncthreads();
	if (phase != 0 || c >= C) return;
	const float v = inv_b * fold_phases(red, ch);
	saved_mean[c] = mu;
	mean[c] = mom * mean[c] + (1.f - mom) * mu;
	var[c] = mom * var[c] + (1.f - mom) * v;
	const float is = 1.f / sqrtf(v + eps);
	saved_inv_std[c] = is;
	const float w = is * scale[c];
	nscale[c] = w;
	nbias[c] = bias[c] - mu * w;
}
template <class T, int G>
__global__ void __launch_bounds__(256) bn_plane_back_stats_kernel(const T* __restrict__ x, const T* __restrict__ g, const long planes, const int C, const long inner, const float* __restrict__ mean, const float* __restrict__ inv_std, float* __restrict__ pg, float* __restrict__ pxg)
{
	constexpr int W = 16 / sizeof(T);
	typedef typename pack16<T>::type V;
	PLANE_LOOP(G) {
		const int c = (int)(pl % C);
		const float mu = mean[c], is = inv_std[c];
		const T* const gp = g + pl * inner;
		const T* const xp = x + pl * inner;
		float sg = 0.f, sx = 0.f;
		if ((inner % W) == 0 && ((((uintptr_t)gp) | ((uintptr_t)xp)) & 15) == 0) {
			const long nv = inner / W;
			for (long i = lane; i < nv; i += G) {
				const V gv = ((const V*)gp)[i], xv = ((const V*)xp)[i];
#pragma unroll
				for (int e = 0; e < W; e++) { sg += (float)gv[e]; sx += ((float)xv[e] - mu) * is * (float)gv[e]; }
			}
		} else
			for (long i = lane; i < inner; i += G) { const float gv = (float)gp[i]; sg += gv; sx += ((float)xp[i] - mu) * is * gv; }
		sg = group_sum<G>(sg); sx = group_sum<G>(sx);
		if (lane == 0) { pg[pl] = sg; pxg[pl] = sx; }
	}
}
// lanes per plane of the batch-norm plane kernels: 16 (four planes per wave) for planes of at most 1 KB
template <class T> static int plane_lanes(const long inner) { return tune(TUNE_BN_SMALL_PLANES) && inner * (long)sizeof(T) <= 1024 ? 16 : 64; }
static unsigned plane_grid(const long planes)
{
	// one plane per wave over the WHOLE tensor, no grid-stride cap (round 3): the same lesson as the element-wise maps (section 3.2 of DESIGN.md -- a few
	// thousand workgroups striding a multi-GB tensor keep DRAM pages from all over it in flight, a front of workgroups walking it in order does not; the
	// capped form ran the batch-norm passes at ~3.4 TB/s).  TUNE_GRID_WG_PER_CU > 0 restores a cap.
	const long want = (planes + 3) / 4, per_cu = tune(TUNE_GRID_WG_PER_CU);
	const long cap = per_cu > 0 ? (long)device_cu_count() * per_cu : 0x7fffffffL;
	return (unsigned)(want < cap ? (want > 0 ? want : 1) : cap);
}

// ---- per-channel finishing steps (C elements each) ---------------------------------------------------------------------
// after the sum pass: saved_mean = sum / B; running mean = m * mean + (1 - m) * saved_mean      (batch_norm_cpu_ref.c:67-72)
__global__ void bn_mean_kernel(float* saved_mean, float* mean, const int C, const float inv_b, const float m)
{
	const int c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= C) return;
	const float mu = inv_b * saved_mean[c];
	saved_mean[c] = mu;
	mean[c] = m * mean[c] + (1.f - m) * mu;
}
// after the centred-square pass: var_b = sum / B (biased); running var; inv_std = 1 / sqrt(var_b + eps); affine y = x * ns + nb  (:107-118,:163-232)
__global__ void bn_var_kernel(float* saved_inv_std, float* var, const float* saved_mean, const float* scale, const float* bias, float* nscale, float* nbias, const int C, const float inv_b, const float m, const float eps)
{
	const int c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= C) return;
	const float v = inv_b * saved_inv_std[c];
	var[c] = m * var[c] + (1.f - m) * v;
	const float is = 1.f / sqrtf(v + eps);
	saved_inv_std[c] = is;
	const float w = is * scale[c];
	nscale[c] = w;
	nbias[c] = bias[c] - saved_mean[c] * w;
}
__global__ void bn_test_affine_kernel(const float* mean, const float* var, const float* scale, const float* bias, float* nscale, float* nbias, const int C, const float eps)
{ // :233-297.  Reference quirk kept for parity: test mode divides by (sqrt(var) + eps), eps OUTSIDE the root (:277)
	const int c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= C) return;
	const float w = scale[c] / (sqrtf(var[c]) + eps);
	nscale[c] = w;
	nbias[c] = bias[c] - mean[c] * w;
}
// y = x * nscale[c] + nbias[c]   (any layout: one element per lane; the plane kernels below are the NCHW fast path)
template <class T>
__global__ void __launch_bounds__(256) bn_apply_kernel(const T* x, T* y, const float* nscale, const float* nbias, const size_t n, const int C, const long inner, const int relu)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
		const int c = (int)((inner == 1 ? i : i / inner) % C);
		const float r = (float)x[i] * nscale[c] + nbias[c];
		y[i] = (T)(relu && !(r > 0.f) ? 0.f : r);
	}
}
// h = (scale * inv_std / B) * (B * g - dbias - xhat * dscale)      (:440-470)
template <class T>
__global__ void __launch_bounds__(256) bn_back_kernel(const T* x, const T* g, T* h, const float* scale, const float* mean, const float* inv_std, const float* dscale, const float* dbias, const size_t n, const int C, const long inner, const float B)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
		const int c = (int)((inner == 1 ? i : i / inner) % C);
		const float is = inv_std[c];
		const float xhat = ((float)x[i] - mean[c]) * is;
		h[i] = (T)((1.f / B * scale[c] * is) * (B * (float)g[i] - dbias[c] - xhat * dscale[c]));
	}
}
// The same two maps, a wave per plane (inner > 1): the channel is one modulo per plane instead of a 64-bit division per element,
// 16-byte accesses when the plane allows.  h = a * g + b * x + k with per-channel a, b, k.
template <class T, int G>
__global__ void __launch_bounds__(256) bn_apply_planes_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ nscale, const float* __restrict__ nbias, const long planes, const int C, const long inner, const int relu)
{
	constexpr int W = 16 / sizeof(T);
	typedef typename pack16<T>::type V;
	PLANE_LOOP(G) {
		const int c = (int)(pl % C);
		const float w = nscale[c], b = nbias[c];
		const T* const xp = x + pl * inner;
		T* const yp = y + pl * inner;
		if ((inner % W) == 0 && ((((uintptr_t)xp) | ((uintptr_t)yp)) & 15) == 0) {
			const long nv = inner / W;
			for (long i = lane; i < nv; i += G) {
				const V v = ((const V*)xp)[i];
				V r;
#pragma unroll
				for (int e = 0; e < W; e++) { const float t = (float)v[e] * w + b; r[e] = (T)(relu && !(t > 0.f) ? 0.f : t); }
				((V*)yp)[i] = r;
			}
		} else
			for (long i = lane; i < inner; i += G) { const float t = (float)xp[i] * w + b; yp[i] = (T)(relu && !(t > 0.f) ? 0.f : t); }
	}
}
template <class T, int G>
__global__ void __launch_bounds__(256) bn_back_planes_kernel(const T* __restrict__ x, const T* __restrict__ g, T* __restrict__ h, const float* __restrict__ scale, const float* __restrict__ mean, const float* __restrict__ inv_std, const float* __restrict__ dscale, const float* __restrict__ dbias, const long planes, const int C, const long inner, const float B)
{
	constexpr int W = 16 / sizeof(T);
	typedef typename pack16<T>::type V;
	PLANE_LOOP(G) {
		const int c = (int)(pl % C);
		const float is = inv_std[c], mu = mean[c], k = 1.f / B * scale[c] * is, db = dbias[c], ds = dscale[c];
		const T* const xp = x + pl * inner;
		const T* const gp = g + pl * inner;
		T* const hp = h + pl * inner;
		if ((inner % W) == 0 && ((((uintptr_t)xp) | ((uintptr_t)gp) | ((uintptr_t)hp)) & 15) == 0) {
			const long nv = inner / W;
			for (long i = lane; i < nv; i += G) {
				const V xv = ((const V*)xp)[i], gv = ((const V*)gp)[i];
				V r;
#pragma unroll
				for (int e = 0; e < W; e++) { const float xhat = ((float)xv[e] - mu) * is; r[e] = (T)(k * (B * (float)gv[e] - db - xhat * ds)); }
				((V*)hp)[i] = r;
			}
		} else
			for (long i = lane; i < inner; i += G) { const float xhat = ((float)xp[i] - mu) * is; hp[i] = (T)(k * (B * (float)gp[i] - db - xhat * ds)); }
	}
}

// ---- cluster kernels (round 4): x read ONCE -------------------------------------------------------------------------------------------------------
// Training-mode batch norm needs every element of a channel twice -- for the statistics, then to be normalised -- and the plane kernels above read x from
// HBM both times (forward 3 |x| of traffic for 2 |x| of algorithm, backward 5 for 3): at batch 256 a ResNet channel is 0.2 - 13 MB, the tensor 0.1 - 0.8 GB,
// nothing the caches hold between two kernels.  Here a CLUSTER of G workgroups owns one channel and keeps it ON CHIP between the two uses: every workgroup
// loads its contiguous share of the channel's 16-byte chunks into registers (NV per thread, all loads issued before the first use), reduces it, publishes its
// partial sums, picks up its siblings', folds them in a fixed order -- every workgroup of the cluster arrives at bit-identical statistics -- and normalises
// out of the registers.  The register files of 256 CUs hold ~100 MB: the tensor passes through in a few waves of clusters.
//   * Statistics: per workgroup the sum, then the centred second moment about the WORKGROUP's mean (from registers: exact two-pass form), combined over
//     the cluster with Chan et al.'s fold -- the plane kernels' scheme with "workgroup share" in place of "plane".
//   * The hand-over is placement- and order-independent (MI355X guide, inter-workgroup visibility): a workgroup's place in the grid is a TICKET it draws
//     from an agent-scope counter when it starts, so the workgroups holding tickets below any resident one are resident or finished -- a cluster can only
//     be waiting for siblings that are running or about to be dispatched, whatever order the dispatcher chose; the only requirement is that G workgroups
//     fit on the chip at once next to whatever else runs (G <= BN_CLUSTER_MAX_G of >= 1024 slots).  Partials travel as 8-byte {epoch, value} granules,
//     one agent-scope store each, polled by one wave with agent-scope loads + s_sleep; no fences (the data is the flag).  Polls are bounded
//     (CLUSTER_SPIN_LIMIT): a workgroup that gives up raises the timeout word and the next synchronise stops the process.
//   * The last workgroup to finish puts the ticket and done counters back to zero for the stream's next launch.
constexpr int BN_CLUSTER_NV = 24;      // 16-byte chunks a thread holds (forward; backward holds half as many of x and of g): 80 VGPRs of data + 20 of offsets, 4 workgroups per CU
constexpr int BN_CLUSTER_MAX_G = 320;  // workgroups per cluster the host will ask for
static long g_bn_cluster_launches = 0; // nnc_mi355x_debug_bn_cluster_launches(): tests assert which kernels ran
constexpr int BN_CLUSTER_LDS = (8 + 2 * BN_CLUSTER_MAX_G + 8) * (int)sizeof(float);
struct bn_cluster_geom_t {
	int C;
	long inner;       // elements per plane
	unsigned nv;      // 16-byte chunks per plane
	unsigned magic;   // floor(2^32 / nv) + 1: q / nv == __umulhi(q, magic) for every chunk index q of a channel (host checks Q * nv < 2^32)
	unsigned Q;       // chunks per channel = outer * nv
	unsigned per;     // chunks per workgroup
	unsigned G;       // workgroups per channel
	unsigned grid;    // C * G
	unsigned bytes;   // of the whole tensor (< 4 GB: every chunk is a 32-bit byte offset against a raw-buffer descriptor based at the channel's first plane)
	unsigned image_bytes; // C * inner * sizeof(T)
};
typedef unsigned int bn_u4 __attribute__((ext_vector_type(4)));
// sum over the workgroup's 256 threads, the same value in every thread (fixed order: butterfly inside a wave, then the four waves)
__device__ __forceinline__ float cluster_block_sum(float v, float* const red)
{
	v = wave_sum(v);
	if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
	__syncthreads();
	const float r = (red[0] + red[1]) + (red[2] + red[3]);
	__syncthreads();
	return r;
}
// wave 0 collects the cluster's NP partials per workgroup into part[w * NP + k]; false = gave up
template <int NP>
__device__ __forceinline__ void cluster_collect(const unsigned long long* const slots, const unsigned G, const unsigned epoch, float* const part, unsigned* const timeout_word)
{
	const int lane = threadIdx.x & 63;
	const unsigned total = G * NP;
	for (unsigned base = 0; base < total; base += 64) { // (wave-uniform trip count)
		const unsigned idx = base + lane;
		unsigned spins = 0;
		for (;;) {
			unsigned long long gr = ((unsigned long long)epoch << 32);
			if (idx < total) gr = nnc_load_granule(slots + idx);
			const int ok = (unsigned)(gr >> 32) == epoch;
			if (ok && idx < total) part[idx] = __uint_as_float((unsigned)gr);
			int pending = ok ? 0 : 1;
			for (int off = 32; off > 0; off >>= 1) pending += __shfl_xor(pending, off, 64);
			if (!pending) break;
			if (++spins > CLUSTER_SPIN_LIMIT) { if (lane == 0) nnc_store_agent(timeout_word, 0xb0000000u | NP); break; }
			NNC_SPIN_SLEEP();
		}
	}
}
// chunk q of a channel -> byte offset from the channel's first plane (image n = q / nv, chunk i = q % nv of its plane); q >= q1 -> out of the descriptor's
// range: the load returns zeros, the store is dropped (no branch around either)
template <int CB>
__device__ __forceinline__ unsigned cluster_chunk_voff(const bn_cluster_geom_t& g, const unsigned q, const unsigned q1)
{
	const unsigned n = __umulhi(q, g.magic);
	const unsigned i = q - n * g.nv;
	return (n * g.image_bytes + i * (unsigned)CB) | (q < q1 ? 0u : 0xffffffffu); // (an OR with a select: hipcc turned the plain select into a branch per chunk)
}
template <class T>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t cluster_rsrc(const T* const p, const bn_cluster_geom_t& g, const unsigned c)
{
	const unsigned first = c * (unsigned)g.inner * (unsigned)sizeof(T);
	return __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p + first), 0, g.bytes - first, 0x00020000);
}
__device__ __forceinline__ void cluster_finish(cluster_sync_t* const sync, const unsigned grid)
{
	if (threadIdx.x == 0 && nnc_fetch_add_agent(&sync->done, 1) == grid - 1) { nnc_store_agent(&sync->ticket, 0); nnc_store_agent(&sync->done, 0); }
}
// One chunk: CB = 16, 8 or 4 bytes as they arrived (kept as dwords: as a vector of halves hipcc unpacks every chunk into one register per element on
// arrival).  CB is the largest power of two that divides a plane's byte size and the tensors' alignment: 16 for ResNet's 56 x 56 .. 14 x 14 fp32 planes, 8 for
// its 14 x 14 half planes (392 bytes), 4 for the 7 x 7 fp32 planes (196 bytes).
typedef unsigned int bn_u2 __attribute__((ext_vector_type(2)));
template <class T, int CB>
struct bn_chunk_t {
	static constexpr int E = CB / (int)sizeof(T); // elements
	unsigned d[CB / 4];
	__device__ __forceinline__ void load(const __amdgpu_buffer_rsrc_t rs, const unsigned voff)
	{
		if constexpr (CB == 16) { const bn_u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0); d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3]; }
		else if constexpr (CB == 8) { const bn_u2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, 0, 0); d[0] = v[0]; d[1] = v[1]; }
		else d[0] = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, 0, 0);
	}
	__device__ __forceinline__ void store(const __amdgpu_buffer_rsrc_t rs, const unsigned voff) const
	{
		if constexpr (CB == 16) __builtin_amdgcn_raw_buffer_store_b128(bn_u4{ d[0], d[1], d[2], d[3] }, rs, voff, 0, 0);
		else if constexpr (CB == 8) __builtin_amdgcn_raw_buffer_store_b64(bn_u2{ d[0], d[1] }, rs, voff, 0, 0);
		else __builtin_amdgcn_raw_buffer_store_b32(d[0], rs, voff, 0, 0);
	}
	// an opaque "new" value: otherwise the halves converted for one pass stay converted -- a register per element -- for the next
	__device__ __forceinline__ void pin()
	{
#pragma unroll
		for (int k = 0; k < CB / 4; k++) NNC_PIN_V(d[k]);
	}
	__device__ __forceinline__ float get(const int e) const
	{
		if constexpr (sizeof(T) == 4) return __uint_as_float(d[e]);
		else return (float)__builtin_bit_cast(half_t, (unsigned short)(d[e >> 1] >> (16 * (e & 1))));
	}
	__device__ __forceinline__ void set(const int e, const float v)
	{
		if constexpr (sizeof(T) == 4) d[e] = __float_as_uint(v);
		else {
			const unsigned h = (unsigned)__builtin_bit_cast(unsigned short, (half_t)v);
			d[e >> 1] = (e & 1) ? (d[e >> 1] & 0xffffu) | (h << 16) : (d[e >> 1] & 0xffff0000u) | h;
		}
	}
};

template <class T, int CB, int NV>
__global__ void __launch_bounds__(256, 4) bn_cluster_forw_kernel(const T* __restrict__ x, T* __restrict__ y, const bn_cluster_geom_t g, cluster_sync_t* const sync, const unsigned epoch, unsigned* const timeout_word, const float* __restrict__ scale, const float* __restrict__ bias, float* mean, float* var, float* __restrict__ saved_mean, float* __restrict__ saved_inv_std, const float inv_b, const float mom, const float eps, const int relu)
{
	typedef bn_chunk_t<T, CB> chunk_t;
	constexpr int W = chunk_t::E;
	HIP_DYNAMIC_SHARED(float, lds)
	float* const red = lds;                 // [4] + ticket
	float* const part = lds + 8;            // [G][2]
	float* const stat = lds + 8 + 2 * BN_CLUSTER_MAX_G;
	const int t = threadIdx.x;
	if (t == 0) ((unsigned*)red)[4] = nnc_fetch_add_agent(&sync->ticket, 1);
	__syncthreads();
	const unsigned ticket = ((const unsigned*)red)[4];
	const unsigned c = ticket / g.G, w = ticket - c * g.G;
	const unsigned q0 = w * g.per, q1 = q0 + g.per < g.Q ? q0 + g.per : g.Q;
	const __amdgpu_buffer_rsrc_t rx = cluster_rsrc(x, g, c), ry = cluster_rsrc(y, g, c);
	chunk_t raw[NV];
#pragma unroll
	for (int j = 0; j < NV; j++) raw[j].load(rx, cluster_chunk_voff<CB>(g, q0 + t + 256 * j, q1));
	// (chunks past the share's end were loaded as zeros: they add nothing to the sum, and the second moment skips them by a select, not a branch --
	// per-chunk lane masks would cost 2 SGPRs each; the sched_barriers keep hipcc from converting / centring every chunk at once, which spills)
	const int nj = q1 > q0 + t ? (int)((q1 - q0 - t + 255) >> 8) : 0; // chunks this thread holds
	float s = 0.f;
#pragma unroll
	for (int j = 0; j < NV; j++) {
#pragma unroll
		for (int e = 0; e < W; e++) s += raw[j].get(e);
		__builtin_amdgcn_sched_barrier(0);
	}
	const float cnt = (float)(q1 - q0) * (float)W;
	const float s_w = cluster_block_sum(s, red);
	const float m_w = s_w / cnt;
	float m2 = 0.f;
#pragma unroll
	for (int j = 0; j < NV; j++) {
		const bool on = j < nj;
		raw[j].pin();
#pragma unroll
		for (int e = 0; e < W; e++) { const float d = raw[j].get(e) - m_w; m2 += on ? d * d : 0.f; }
		__builtin_amdgcn_sched_barrier(0);
	}
	const float m2_w = cluster_block_sum(m2, red);
	float mu = s_w * inv_b, M2 = m2_w;
	if (g.G > 1) {
		unsigned long long* const slots = (unsigned long long*)(sync + 1) + (size_t)c * g.G * 2;
		if (t == 0) { nnc_store_granule(slots + w * 2, epoch, s_w); nnc_store_granule(slots + w * 2 + 1, epoch, m2_w); }
		if (t < 64) {
			cluster_collect<2>(slots, g.G, epoch, part, timeout_word);
			// the cluster's mean, then Chan's fold of the shares' second moments about it -- lane-strided partial sums, then the butterfly: every workgroup of
			// the cluster runs the same instructions on the same numbers
			float a = 0.f;
			for (unsigned k = t; k < g.G; k += 64) a += part[2 * k];
			a = wave_sum(a);
			const float mean_all = a * inv_b;
			float b2 = 0.f;
			for (unsigned k = t; k < g.G; k += 64) {
				const unsigned k0 = k * g.per, k1 = k0 + g.per < g.Q ? k0 + g.per : g.Q;
				const float n_k = (float)(k1 - k0) * (float)W;
				const float d = part[2 * k] / n_k - mean_all;
				b2 += part[2 * k + 1] + n_k * d * d;
			}
			b2 = wave_sum(b2);
			if (t == 0) { stat[0] = mean_all; stat[1] = b2; }
		}
		__syncthreads();
		mu = stat[0]; M2 = stat[1];
	}
	const float vb = M2 * inv_b;
	const float is = 1.f / sqrtf(vb + eps);
	const float ws = is * scale[c], bs = bias[c] - mu * ws;
	if (w == 0 && t == 0) { // batch_norm_cpu_ref.c:67-72, :107-118
		saved_mean[c] = mu;
		saved_inv_std[c] = is;
		mean[c] = mom * mean[c] + (1.f - mom) * mu;
		var[c] = mom * var[c] + (1.f - mom) * vb;
	}
	unsigned ts = t; // the stores' offsets are computed again (a few integer operations per chunk) instead of living in registers since the loads
	NNC_PIN_V(ts);
#pragma unroll
	for (int j = 0; j < NV; j++) {
		chunk_t r;
		raw[j].pin();
#pragma unroll
		for (int k = 0; k < CB / 4; k++) r.d[k] = 0;
#pragma unroll
		for (int e = 0; e < W; e++) { const float o = raw[j].get(e) * ws + bs; r.set(e, relu && !(o > 0.f) ? 0.f : o); }
		r.store(ry, cluster_chunk_voff<CB>(g, q0 + ts + 256 * j, q1));
		__builtin_amdgcn_sched_barrier(0);
	}
	cluster_finish(sync, g.grid);
}

// backward: sum of g and of xhat * g over the channel (fused, one pass over the registers), then h = (scale * inv_std / B) * (B * g - dbias - xhat * dscale)
template <class T, int CB, int NV>
__global__ void __launch_bounds__(256, 4) bn_cluster_back_kernel(const T* __restrict__ x, const T* __restrict__ gr, T* __restrict__ h, const bn_cluster_geom_t g, cluster_sync_t* const sync, const unsigned epoch, unsigned* const timeout_word, const float* __restrict__ scale, const float* __restrict__ mean, const float* __restrict__ inv_std, float* __restrict__ dscale, float* __restrict__ dbias, const float B)
{
	typedef bn_chunk_t<T, CB> chunk_t;
	constexpr int W = chunk_t::E;
	HIP_DYNAMIC_SHARED(float, lds)
	float* const red = lds;
	float* const part = lds + 8;
	float* const stat = lds + 8 + 2 * BN_CLUSTER_MAX_G;
	const int t = threadIdx.x;
	if (t == 0) ((unsigned*)red)[4] = nnc_fetch_add_agent(&sync->ticket, 1);
	__syncthreads();
	const unsigned ticket = ((const unsigned*)red)[4];
	const unsigned c = ticket / g.G, w = ticket - c * g.G;
	const unsigned q0 = w * g.per, q1 = q0 + g.per < g.Q ? q0 + g.per : g.Q;
	const __amdgpu_buffer_rsrc_t rx = cluster_rsrc(x, g, c), rg = cluster_rsrc(gr, g, c), rh = cluster_rsrc(h, g, c);
	chunk_t xraw[NV], graw[NV];
#pragma unroll
	for (int j = 0; j < NV; j++) {
		const unsigned voff = cluster_chunk_voff<CB>(g, q0 + t + 256 * j, q1);
		xraw[j].load(rx, voff);
		graw[j].load(rg, voff);
	}
	const float mu = mean[c], is = inv_std[c];
	float sg = 0.f, sx = 0.f; // (chunks past the share's end are zeros in g: they add nothing to either sum)
#pragma unroll
	for (int j = 0; j < NV; j++) {
		xraw[j].pin(); graw[j].pin();
#pragma unroll
		for (int e = 0; e < W; e++) { const float gg = graw[j].get(e); sg += gg; sx += (xraw[j].get(e) - mu) * is * gg; }
		__builtin_amdgcn_sched_barrier(0);
	}
	const float sg_w = cluster_block_sum(sg, red);
	const float sx_w = cluster_block_sum(sx, red);
	float db = sg_w, ds = sx_w;
	if (g.G > 1) {
		unsigned long long* const slots = (unsigned long long*)(sync + 1) + (size_t)c * g.G * 2;
		if (t == 0) { nnc_store_granule(slots + w * 2, epoch, sg_w); nnc_store_granule(slots + w * 2 + 1, epoch, sx_w); }
		if (t < 64) {
			cluster_collect<2>(slots, g.G, epoch, part, timeout_word);
			float a = 0.f, b2 = 0.f;
			for (unsigned k = t; k < g.G; k += 64) { a += part[2 * k]; b2 += part[2 * k + 1]; }
			a = wave_sum(a); b2 = wave_sum(b2);
			if (t == 0) { stat[0] = a; stat[1] = b2; }
		}
		__syncthreads();
		db = stat[0]; ds = stat[1];
	}
	if (w == 0 && t == 0) { dbias[c] = db; dscale[c] = ds; }
	const float k = 1.f / B * scale[c] * is;
	unsigned ts = t;
	NNC_PIN_V(ts);
#pragma unroll
	for (int j = 0; j < NV; j++) {
		chunk_t r;
		xraw[j].pin(); graw[j].pin();
#pragma unroll
		for (int kk = 0; kk < CB / 4; kk++) r.d[kk] = 0;
#pragma unroll
		for (int e = 0; e < W; e++) { const float xhat = (xraw[j].get(e) - mu) * is; r.set(e, k * (B * graw[j].get(e) - db - xhat * ds)); }
		r.store(rh, cluster_chunk_voff<CB>(g, q0 + ts + 256 * j, q1));
		__builtin_amdgcn_sched_barrier(0);
	}
	cluster_finish(sync, g.grid);
}

// Can the cluster kernels take this tensor, and with which geometry?  CB: bytes per chunk (16 / 8 / 4: the largest that divides the plane's bytes and the
// pointers); chunks_per_thread: what a thread may hold at CB = 16 (twice / four times as many at 8 / 4: the same registers).
template <class T>
static int bn_cluster_plan(const chan_view_t& v, const int chunks_per_thread, const void* p0, const void* p1, const void* p2, bn_cluster_geom_t* g)
{
	const long mode = tune(TUNE_BN_CLUSTER);
	if (mode <= 0 || v.inner <= 1 || v.C < 1 || v.outer < 1) return 0;
	const uintptr_t bits = ((uintptr_t)p0) | ((uintptr_t)p1) | ((uintptr_t)p2) | (uintptr_t)(v.inner * (long)sizeof(T));
	const int CB = (bits & 15) == 0 ? 16 : (bits & 7) == 0 ? 8 : (bits & 3) == 0 ? 4 : 0;
	if (!CB) return 0;
	const long nv = v.inner * (long)sizeof(T) / CB, Q = v.outer * nv;
	const unsigned long long bytes = (unsigned long long)v.outer * v.C * v.inner * sizeof(T);
	if (bytes > 0xfffffff0ull) return 0; // 32-bit byte offsets
	if (nv > 0xfffff || Q > 0x7fffffffL || (unsigned long long)Q * (unsigned long long)nv >= (1ull << 32)) return 0;
	long cap = 256L * chunks_per_thread * (16 / CB); // chunks one workgroup holds
	if (mode > 1 && mode < cap) cap = mode; // (tests: several workgroups per channel on small tensors)
	// A tensor whose channels fit a few full shares gives a launch of a few hundred workgroups, each a chain of dependent steps -- ticket, a deep queue of loads,
	// four barriers, the hand-over, the stores -- that nothing overlaps: 256 x 14 x 14 halves at batch 256 took 30 us for 51 MB (1.7 TB/s,
	// profiles/r04_v2_bn_bench.txt).  Smaller shares, more workgroups: the chains shorten and the chip's slots fill (TUNE_BN_CLUSTER_SLOTS workgroups, about).
	const long slots = tune(TUNE_BN_CLUSTER_SLOTS);
	if (slots > 0 && mode == 1) {
		const long share = (Q * (long)v.C + slots - 1) / slots; // chunks per workgroup that make `slots` workgroups
		const long least = 256L * 2;
		if (share < cap) cap = share > least ? share : least;
	}
	long G = (Q + cap - 1) / cap;
	if (G > BN_CLUSTER_MAX_G) return 0;
	const long per = (Q + G - 1) / G;
	G = (Q + per - 1) / per; // no empty workgroup
	const long grid = (long)v.C * G;
	if (grid > 0x7fffffffL || (size_t)grid * 16 > CLUSTER_SYNC_BYTES - 256) return 0;
	g->C = v.C; g->inner = v.inner; g->nv = (unsigned)nv; g->magic = (unsigned)((1ull << 32) / (unsigned long long)nv) + 1u;
	g->Q = (unsigned)Q; g->per = (unsigned)per; g->G = (unsigned)G; g->grid = (unsigned)grid;
	g->bytes = (unsigned)bytes; g->image_bytes = (unsigned)((unsigned long long)v.C * v.inner * sizeof(T));
	return CB;
}

// Derive the [outer][C][inner] view of x from the statistics tensor.
//   * statistics with x's rank (or right-aligned against it, lib/nnc/cmd/norm/ccv_nnc_batch_norm_cpu_ref.c:28-33): every axis of
//     extent 1 is reduced, exactly one axis is kept;
//   * a ONE-dimensional statistics tensor of C elements against an image tensor -- what ccv_cnnp_batch_norm creates
//     (lib/nnc/ccv_cnnp_model_addons.c:951-957: dim[0] = ccv_nnc_tensor_get_c(params)) -- is per CHANNEL, and the channel axis is
//     the one x's FORMAT names (NCHW: axis 1 of 4 / 0 of 3, NHWC: the last axis, CHWN: axis 0), as the backend being replaced reads
//     it through its format-aware tensor descriptors (gpu/ccv_nnc_batch_norm_gpu_cudnn.cu:13-100).  Right-aligning such a tensor
//     against NCHW data would pit C against W: found when the reference's ResNet trainer graph (NCHW) ran here -- most of its
//     batch norms were refused (and the host does not look at a command's return code), the ones with W == C normalised over
//     the wrong axis.
static bool chan_view(const ccv_nnc_tensor_t* x, const ccv_nnc_tensor_t* stat, chan_view_t* v)
{
	const int nd = tensor_nd(x->info.dim), snd = tensor_nd(stat->info.dim);
	if (nd < 1 || nd > 4 || snd > nd || !tensor_contiguous(x) || !tensor_contiguous(stat)) return false;
	int kept = -1;
	if (snd == 1 && nd >= 2 && stat->info.dim[0] > 1) {
		if (x->info.format == CCV_TENSOR_FORMAT_NCHW) kept = nd == 4 ? 1 : (nd == 3 ? 0 : nd - 1);
		else if (x->info.format == CCV_TENSOR_FORMAT_CHWN) kept = 0;
		else kept = nd - 1;
		if (stat->info.dim[0] != x->info.dim[kept]) return false;
	} else
		for (int k = 0; k < nd; k++) {
			const int sd = k - (nd - snd) >= 0 ? stat->info.dim[k - (nd - snd)] : 1;
			if (sd == 1) continue;
			if (sd != x->info.dim[k] || kept >= 0) return false;
			kept = k;
		}
	if (kept < 0) { // every axis reduced: one "channel"
		v->outer = (long)tensor_count(x->info); v->C = 1; v->inner = 1;
		return true;
	}
	v->outer = 1; v->inner = 1;
	for (int k = 0; k < kept; k++) v->outer *= x->info.dim[k];
	for (int k = kept + 1; k < nd; k++) v->inner *= x->info.dim[k];
	v->C = x->info.dim[kept];
	return true;
}

// T: the element type of x and y (float, or _Float16 when half_stage.cpp hands the trainer's CCV_16F activations through as they
// are); the statistics, scale and bias are fp32 here either way.
template <class T>
static int bnorm_forw_t(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	if (input_size != 5 || output_size < 1) return CCV_NNC_EXEC_INVALID;
	for (int i = 0; i < 5; i++) if (!inputs[i] || (i > 0 && CCV_GET_DATA_TYPE(inputs[i]->info.datatype) != CCV_32F) || !tensor_contiguous(inputs[i])) return CCV_NNC_EXEC_INVALID;
	const ccv_nnc_tensor_t* x = inputs[0];
	ccv_nnc_tensor_t* y = outputs[0];
	if (!y || !tensor_contiguous(y) || tensor_count(y->info) != tensor_count(x->info) || y->info.datatype != x->info.datatype) return CCV_NNC_EXEC_INVALID;
	const T* const xp = (const T*)x->data.u8;
	T* const yp = (T*)y->data.u8;
	chan_view_t v;
	if (!chan_view(x, inputs[1], &v)) return CCV_NNC_EXEC_INVALID;
	for (int i = 1; i < 5; i++) if ((int)tensor_count(inputs[i]->info) != v.C) return CCV_NNC_EXEC_INVALID;
	const size_t n = tensor_count(x->info);
	if (n == 0) return CCV_NNC_EXEC_SUCCESS;
	const int relu = cmd.algorithm > 0 && (cmd.algorithm & NNC_MI355X_BNORM_ALGO_FUSE_RELU) ? 1 : 0; // opt-in: y = max(0, .) as it is written (include/nnc_mi355x.h)
	hipStream_t stream = stream_of(stream_context);
	// (bench.py roofline leg, config 4: the whole command between two events; algorithmic bytes = x read once + y written once)
	ProfScope prof(cmd.info.bnorm.is_test ? "bnorm_fwd_test|nnc::bn_apply_kernel" : "bnorm_fwd|nnc::chan_reduce + bn_apply_kernel", 0, 2.0 * sizeof(T) * (double)n, (int)(n / v.C), v.C, 1, 1, 1, stream);
	const float* scale = inputs[1]->data.f32;
	const float* bias = inputs[2]->data.f32;
	float* mean = inputs[3]->data.f32;
	float* var = inputs[4]->data.f32;
	const int cb = (v.C + 255) / 256;
	// per-channel affine lives in front of the reduction partials in the workspace
	WorkspaceScope ws(stream_context, sizeof(float) * 2 * (size_t)v.C, sizeof(float) * (size_t)v.C * (size_t)(v.inner == 1 ? (long)device_cu_count() * 4 + 64 : 2 * v.outer));
	float* nscale = (float*)ws.prefix();
	if (!nscale) return CCV_NNC_EXEC_OOM;
	float* nbias = nscale + v.C;
	int ret;
	if (!cmd.info.bnorm.is_test) {
		if (output_size != 5 || !outputs[3] || !outputs[4]) return CCV_NNC_EXEC_INVALID;
		// running statistics are updated in place (ccv_nnc_norm.c:19-26)
		if (outputs[1] && outputs[1]->data.f32 != mean) return CCV_NNC_EXEC_INVALID;
		if (outputs[2] && outputs[2]->data.f32 != var) return CCV_NNC_EXEC_INVALID;
		float* saved_mean = outputs[3]->data.f32;
		float* saved_inv_std = outputs[4]->data.f32;
		if ((int)tensor_count(outputs[3]->info) != v.C || (int)tensor_count(outputs[4]->info) != v.C) return CCV_NNC_EXEC_INVALID;
		const float inv_b = 1.f / (float)(n / v.C);
		bn_cluster_geom_t cg;
		if (const int CB = bn_cluster_plan<T>(v, BN_CLUSTER_NV, xp, yp, 0, &cg)) { // a cluster of workgroups per channel: x read once (round 4)
			unsigned epoch = 0;
			unsigned* timeout_word = 0;
			cluster_sync_t* const sync = (cluster_sync_t*)cluster_sync_of(stream_context, (size_t)cg.grid * 16, &epoch, &timeout_word);
			if (sync) {
				ClusterTurn turn(stream_context); // spinning launches are one after the other per device, whatever streams they come from (common.h)
#define BN_CL_FWD(CBV) do { const auto kernel = bn_cluster_forw_kernel<T, CBV, BN_CLUSTER_NV * (16 / CBV)>; /* (a name without commas for the launch macros) */ \
					NNC_LAUNCH_CONCURRENT(kernel, dim3(cg.grid), dim3(256), BN_CLUSTER_LDS, stream, xp, yp, cg, sync, epoch, timeout_word, scale, bias, mean, var, saved_mean, saved_inv_std, inv_b, cmd.info.bnorm.momentum, cmd.info.bnorm.epsilon, relu); } while (0)
				if (CB == 16) BN_CL_FWD(16); else if (CB == 8) BN_CL_FWD(8); else BN_CL_FWD(4);
#undef BN_CL_FWD
				HIP_ENFORCE(hipGetLastError());
				++g_bn_cluster_launches;
				return CCV_NNC_EXEC_SUCCESS;
			}
		}
		if (v.inner > 1) { // planes: one sweep from HBM for both statistics (see bn_plane_stats_kernel)
			const long planes = v.outer * v.C;
			float* const psum = (float*)workspace_of(stream_context, sizeof(float) * 2 * (size_t)planes);
			if (!psum) return CCV_NNC_EXEC_OOM;
			{
				constexpr int W = 16 / (int)sizeof(T);
				const long nvec = v.inner / W;
				const bool vec = v.inner % W == 0 && (((uintptr_t)xp) & 15) == 0;
				const int G = plane_lanes<T>(v.inner);
#define BN_STATS(KERNEL, GRID) hipLaunchKernelGGL(HIP_KERNEL_NAME(KERNEL), dim3(plane_grid(GRID)), dim3(256), 0, stream, xp, planes, v.inner, psum, psum + planes)
				if (G == 16) { if (vec && nvec <= 64) BN_STATS((bn_plane_stats_reg_kernel<T, 16, 4>), (planes + 3) / 4); else BN_STATS((bn_plane_stats_kernel<T, 16>), (planes + 3) / 4); }
				else if (vec && nvec <= 256) BN_STATS((bn_plane_stats_reg_kernel<T, 64, 4>), planes);
				else if (vec && nvec <= 1024) BN_STATS((bn_plane_stats_reg_kernel<T, 64, 16>), planes);
				else BN_STATS((bn_plane_stats_kernel<T, 64>), planes);
#undef BN_STATS
			}
			HIP_ENFORCE(hipGetLastError());
			hipLaunchKernelGGL(bn_stats_fold_kernel, dim3((v.C + FOLD_CH - 1) / FOLD_CH), dim3(256), 0, stream, (const float*)psum, (const float*)(psum + planes), v.outer, v.C, (float)v.inner, saved_mean, saved_inv_std, mean, var, scale, bias, nscale, nbias, inv_b, cmd.info.bnorm.momentum, cmd.info.bnorm.epsilon);
			HIP_ENFORCE(hipGetLastError());
		} else {
		if ((ret = chan_reduce<RSum, false, T>(RSum(), xp, (const T*)0, v, saved_mean, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
		hipLaunchKernelGGL(bn_mean_kernel, dim3(cb), dim3(256), 0, stream, saved_mean, mean, v.C, inv_b, cmd.info.bnorm.momentum);
		RCenteredSq f; f.mean = saved_mean;
		if ((ret = chan_reduce<RCenteredSq, false, T>(f, xp, (const T*)0, v, saved_inv_std, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
		hipLaunchKernelGGL(bn_var_kernel, dim3(cb), dim3(256), 0, stream, saved_inv_std, var, (const float*)saved_mean, scale, bias, nscale, nbias, v.C, inv_b, cmd.info.bnorm.momentum, cmd.info.bnorm.epsilon);
		}
	} else
		hipLaunchKernelGGL(bn_test_affine_kernel, dim3(cb), dim3(256), 0, stream, (const float*)mean, (const float*)var, scale, bias, nscale, nbias, v.C, cmd.info.bnorm.epsilon);
	HIP_ENFORCE(hipGetLastError());
	if (v.inner > 1)
		{ if (plane_lanes<T>(v.inner) == 16) hipLaunchKernelGGL(HIP_KERNEL_NAME(bn_apply_planes_kernel<T, 16>), dim3(plane_grid((v.outer * v.C + 3) / 4)), dim3(256), 0, stream, xp, yp, (const float*)nscale, (const float*)nbias, v.outer * v.C, v.C, v.inner, relu); else hipLaunchKernelGGL(HIP_KERNEL_NAME(bn_apply_planes_kernel<T, 64>), dim3(plane_grid(v.outer * v.C)), dim3(256), 0, stream, xp, yp, (const float*)nscale, (const float*)nbias, v.outer * v.C, v.C, v.inner, relu); }
	else
		hipLaunchKernelGGL(HIP_KERNEL_NAME(bn_apply_kernel<T>), dim3(grid_for(n, 256)), dim3(256), 0, stream, xp, yp, (const float*)nscale, (const float*)nbias, n, v.C, v.inner, relu);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}

static int bnorm_forw_entry(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context);
static int _bnorm_forw(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	// recorded when its like has run before: the in-place RELU_FORWARD of the reference's conv - bn - relu blocks folds into the apply pass (peephole.cpp)
	uint64_t sig;
	if (const int e = deferred_take_error(stream_context)) return e; // a recorded command failed when a flush launched it (peephole.cpp)
	if (deferred_try(_bnorm_forw, DEFER_BNORM_FORWARD, cmd, hint, flags, inputs, input_size, outputs, output_size, stream_context, &sig)) return CCV_NNC_EXEC_SUCCESS;
	const int r = bnorm_forw_entry(cmd, hint, flags, inputs, input_size, outputs, output_size, stream_context);
	if (r == CCV_NNC_EXEC_SUCCESS) deferred_mark_good(sig);
	return r;
}
static int bnorm_forw_entry(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	if (input_size >= 1 && inputs[0] && CCV_GET_DATA_TYPE(inputs[0]->info.datatype) == CCV_16F) return bnorm_forw_t<half_t>(cmd, hint, flags, inputs, input_size, outputs, output_size, stream_context);
	return bnorm_forw_t<float>(cmd, hint, flags, inputs, input_size, outputs, output_size, stream_context);
}

template <class T>
static int bnorm_back_t(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	// inputs 0 (g), 5 (x), 6 (scale), 13 (saved_mean), 14 (saved_inv_std) of 15; outputs (h, dscale, dbias)   (ccv_nnc_norm.c:28-36)
	if (input_size != 15 || output_size < 3) return CCV_NNC_EXEC_INVALID;
	const ccv_nnc_tensor_t* g = inputs[0];
	const ccv_nnc_tensor_t* x = inputs[5];
	const ccv_nnc_tensor_t* scale = inputs[6];
	const ccv_nnc_tensor_t* saved_mean = inputs[13];
	const ccv_nnc_tensor_t* saved_inv_std = inputs[14];
	ccv_nnc_tensor_t* h = outputs[0];
	ccv_nnc_tensor_t* dscale = outputs[1];
	ccv_nnc_tensor_t* dbias = outputs[2];
	if (!g || !x || !scale || !saved_mean || !saved_inv_std || !h || !dscale || !dbias) return CCV_NNC_EXEC_INVALID;
	if (!tensor_contiguous(g) || !tensor_contiguous(h) || g->info.datatype != x->info.datatype || h->info.datatype != x->info.datatype) return CCV_NNC_EXEC_INVALID;
	if (CCV_GET_DATA_TYPE(scale->info.datatype) != CCV_32F || CCV_GET_DATA_TYPE(saved_mean->info.datatype) != CCV_32F || CCV_GET_DATA_TYPE(saved_inv_std->info.datatype) != CCV_32F || CCV_GET_DATA_TYPE(dscale->info.datatype) != CCV_32F || CCV_GET_DATA_TYPE(dbias->info.datatype) != CCV_32F) return CCV_NNC_EXEC_INVALID;
	const T* const xp = (const T*)x->data.u8;
	const T* const gp = (const T*)g->data.u8;
	T* const hp = (T*)h->data.u8;
	chan_view_t v;
	if (!chan_view(x, scale, &v)) return CCV_NNC_EXEC_INVALID;
	const size_t n = tensor_count(x->info);
	if (tensor_count(g->info) != n || tensor_count(h->info) != n) return CCV_NNC_EXEC_INVALID;
	if ((int)tensor_count(saved_mean->info) != v.C || (int)tensor_count(saved_inv_std->info) != v.C || (int)tensor_count(dscale->info) != v.C || (int)tensor_count(dbias->info) != v.C) return CCV_NNC_EXEC_INVALID;
	if (n == 0) return CCV_NNC_EXEC_SUCCESS;
	int ret;
	ProfScope prof("bnorm_bwd|nnc::chan_reduce + bn_back_kernel", 0, 3.0 * sizeof(T) * (double)n, (int)(n / v.C), v.C, 1, 1, 1, stream_of(stream_context)); // g, x read; h written
	bn_cluster_geom_t cg;
	if (const int CB = bn_cluster_plan<T>(v, BN_CLUSTER_NV / 2, xp, gp, hp, &cg)) { // a cluster of workgroups per channel: x and g read once (round 4)
		unsigned epoch = 0;
		unsigned* timeout_word = 0;
		cluster_sync_t* const sync = (cluster_sync_t*)cluster_sync_of(stream_context, (size_t)cg.grid * 16, &epoch, &timeout_word);
		if (sync) {
			ClusterTurn turn(stream_context);
#define BN_CL_BWD(CBV) do { const auto kernel = bn_cluster_back_kernel<T, CBV, (BN_CLUSTER_NV / 2) * (16 / CBV)>; \
				NNC_LAUNCH_CONCURRENT(kernel, dim3(cg.grid), dim3(256), BN_CLUSTER_LDS, stream_of(stream_context), xp, gp, hp, cg, sync, epoch, timeout_word, (const float*)scale->data.f32, (const float*)saved_mean->data.f32, (const float*)saved_inv_std->data.f32, dscale->data.f32, dbias->data.f32, (float)(n / v.C)); } while (0)
			if (CB == 16) BN_CL_BWD(16); else if (CB == 8) BN_CL_BWD(8); else BN_CL_BWD(4);
#undef BN_CL_BWD
			HIP_ENFORCE(hipGetLastError());
			++g_bn_cluster_launches;
			return CCV_NNC_EXEC_SUCCESS;
		}
	}
	if (v.inner > 1) { // planes: both sums in one sweep over (x, g)
		const long planes = v.outer * v.C;
		float* const pg = (float*)workspace_of(stream_context, sizeof(float) * 2 * (size_t)planes);
		if (!pg) return CCV_NNC_EXEC_OOM;
		hipStream_t st = stream_of(stream_context);
		if (plane_lanes<T>(v.inner) == 16) hipLaunchKernelGGL(HIP_KERNEL_NAME(bn_plane_back_stats_kernel<T, 16>), dim3(plane_grid((planes + 3) / 4)), dim3(256), 0, st, xp, gp, planes, v.C, v.inner, (const float*)saved_mean->data.f32, (const float*)saved_inv_std->data.f32, pg, pg + planes);
		else hipLaunchKernelGGL(HIP_KERNEL_NAME(bn_plane_back_stats_kernel<T, 64>), dim3(plane_grid(planes)), dim3(256), 0, st, xp, gp, planes, v.C, v.inner, (const float*)saved_mean->data.f32, (const float*)saved_inv_std->data.f32, pg, pg + planes);
		HIP_ENFORCE(hipGetLastError());
		hipLaunchKernelGGL(chan_fold_kernel, dim3((v.C + FOLD_CH - 1) / FOLD_CH, 2), dim3(256), 0, st, (const float*)pg, (const float*)(pg + planes), v.outer, v.C, dbias->data.f32, dscale->data.f32, 0);
		HIP_ENFORCE(hipGetLastError());
	} else {
	if ((ret = chan_reduce<RSum, false, T>(RSum(), gp, (const T*)0, v, dbias->data.f32, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
	RXhatG f; f.mean = saved_mean->data.f32; f.inv_std = saved_inv_std->data.f32;
	if ((ret = chan_reduce<RXhatG, true, T>(f, xp, gp, v, dscale->data.f32, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
	}
	if (v.inner > 1)
		{ if (plane_lanes<T>(v.inner) == 16) hipLaunchKernelGGL(HIP_KERNEL_NAME(bn_back_planes_kernel<T, 16>), dim3(plane_grid((v.outer * v.C + 3) / 4)), dim3(256), 0, stream_of(stream_context), xp, gp, hp, (const float*)scale->data.f32, (const float*)saved_mean->data.f32, (const float*)saved_inv_std->data.f32, (const float*)dscale->data.f32, (const float*)dbias->data.f32, v.outer * v.C, v.C, v.inner, (float)(n / v.C)); else hipLaunchKernelGGL(HIP_KERNEL_NAME(bn_back_planes_kernel<T, 64>), dim3(plane_grid(v.outer * v.C)), dim3(256), 0, stream_of(stream_context), xp, gp, hp, (const float*)scale->data.f32, (const float*)saved_mean->data.f32, (const float*)saved_inv_std->data.f32, (const float*)dscale->data.f32, (const float*)dbias->data.f32, v.outer * v.C, v.C, v.inner, (float)(n / v.C)); }
	else
	hipLaunchKernelGGL(HIP_KERNEL_NAME(bn_back_kernel<T>), dim3(grid_for(n, 256)), dim3(256), 0, stream_of(stream_context), xp, gp, hp, (const float*)scale->data.f32, (const float*)saved_mean->data.f32, (const float*)saved_inv_std->data.f32, (const float*)dscale->data.f32, (const float*)dbias->data.f32, n, v.C, v.inner, (float)(n / v.C));
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}
static int _bnorm_back(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	const int r = (input_size > 5 && inputs[5] && CCV_GET_DATA_TYPE(inputs[5]->info.datatype) == CCV_16F) ? bnorm_back_t<half_t>(cmd, hint, flags, inputs, input_size, outputs, output_size, stream_context)
		: bnorm_back_t<float>(cmd, hint, flags, inputs, input_size, outputs, output_size, stream_context);
	// the scale / bias gradients just enqueued (cmd_comm.cpp "Overlap": deployment (b)'s all-reduce of each starts behind its own writer)
	if (r == CCV_NNC_EXEC_SUCCESS && g_comm_overlap_on.load(std::memory_order_relaxed))
	{
		if (flags & CCV_NNC_ACCUMULATE_OUTPUT) { for (int i = 1; i < output_size && i < 3; i++) comm_gradient_touched(outputs[i]); }
		else if (output_size > 1) comm_gradients_written(outputs + 1, output_size > 3 ? 2 : output_size - 1, stream_context);
	}
	return r;
}

} // namespace

extern "C" long nnc_mi355x_debug_bn_cluster_launches(void) { return g_bn_cluster_launches; }

// out[c] (+)= sum over (o, i) of x[(o * C + c) * inner + i]: the bias gradient of a convolution on NCHW tensors (cmd_conv.cpp)
int nnc::chan_sum_planes(const float* x, long outer, int C, long inner, float* out, int accumulate, ccv_nnc_stream_context_t* ctx)
{
	chan_view_t v;
	v.outer = outer; v.C = C; v.inner = inner;
	return chan_reduce<RSum, false, float>(RSum(), x, (const float*)0, v, out, ctx, accumulate);
}

#define NNC_REG(CMD, BACKEND, FORMATS, DATATYPES, MEMORY, EXEC) \
	extern "C" void _register_command_##CMD##_backend_##BACKEND(ccv_nnc_cmd_backend_registry_t* const registry) \
	{ registry->tensor_formats = (FORMATS); registry->tensor_datatypes = (DATATYPES); registry->tensor_memory = (MEMORY); registry->algorithms = 1; registry->exec = EXEC; NNC_HALF_STAGED(registry, EXEC); }
#define ALL_FORMATS (CCV_TENSOR_FORMAT_NCHW | CCV_TENSOR_FORMAT_NHWC | CCV_TENSOR_FORMAT_CHWN)

NNC_REG(CCV_NNC_BATCH_NORM_FORWARD, CCV_NNC_BACKEND_GPU_CUDNN, ALL_FORMATS, CCV_32F, CCV_TENSOR_GPU_MEMORY, _bnorm_forw)
NNC_REG(CCV_NNC_BATCH_NORM_BACKWARD, CCV_NNC_BACKEND_GPU_CUDNN, ALL_FORMATS, CCV_32F, CCV_TENSOR_GPU_MEMORY, _bnorm_back)
