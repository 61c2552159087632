// SCALED_DOT_PRODUCT_ATTENTION forward / backward (the NLP trainers' row).
// Oracle semantics: lib/nnc/cmd/scaled_dot_product_attention/ccv_nnc_scaled_dot_product_attention_cpu_ref.c:16-497;
// replaces .../gpu/ccv_nnc_scaled_dot_product_attention_flash_attn.cu:18-470.
//   q [B, R, Hq, D], k [B, C, Hk, D], v [B, C, Hk, Dv] (3-d tensors: one head), Hq a multiple of Hk (grouped-query heads);
//   o[b, x, h, :] = sum_y softmax_y(scale * q[b, x, h, :] . k[b, y, h / (Hq / Hk), :] + mask[b, h, x, y]) v[b, y, ., :]
//   causal: row x sees the keys y < x - R + C + 1 (the bottom-right aligned triangle, cpu_ref.c:143); a row that sees none gives 0;
//   optional "unify heads" projection: d = o (as [B * R, Hq * Dv]) w^T + bias (forward only, like the backend being replaced).
// This is a FIRST kernel set, fp32 arithmetic on the VALU (half tensors through half_stage.cpp's fp32 images), written for
// correctness and determinism, streaming K / V blocks through LDS with the running-maximum softmax so no [R, C] score matrix
// exists in memory; it is not yet an MFMA kernel.  Work split:
//   forward   a workgroup per (16 query rows, head, batch): thread (row = t / 16, lane = t % 16) owns the scores of keys
//             lane, lane + 16, ... of a key block, then the output columns lane, lane + 16, ... of its row
//   backward  the forward pass again into scratch (output + log-sum-exp per row), delta[x] = g[x] . o[x];
//             dq: the forward's split, recomputing p = exp(s - lse) per key block;
//             dk, dv: a workgroup per (16 keys, KEY head, batch) walks every query block of every query head that shares the
//             key head and accumulates its 16 rows of dk / dv in registers -- no atomics, one fixed summation order.
#include "common.h"
#include <math.h>

using namespace nnc;

namespace {

#define EXEC_ARGS const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context

struct sdpa_geom_t {
	int B, R, C, Hq, Hk, D, Dv, ratio;
	long q_sb, q_sr, q_sh, k_sb, k_sc, k_sh, v_sb, v_sc, v_sh, o_sb, o_sr, o_sh;
	long m_sb, m_sh, m_sr; // additive mask [B or 1][Hq or 1][R][C], 0 strides where broadcast
	float scale;
	int causal;
};
constexpr int BR = 16; // query rows (forward, dq) / keys (dk, dv) per workgroup

__device__ __forceinline__ float group16_max(float v) { for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64)); return v; }
__device__ __forceinline__ float group16_sum(float v) { for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64); return v; }
// keys visible to query row x
__device__ __forceinline__ int visible_keys(const sdpa_geom_t& g, const int x) { if (!g.causal) return g.C; const int e = x - g.R + g.C + 1; return e < 0 ? 0 : (e > g.C ? g.C : e); }

// ---- forward --------------------------------------------------------------------------------------------------------------------
// DMAX: upper bound of D and Dv (LDS layout); BC: keys per block
template <int DMAX, int BC>
__global__ void __launch_bounds__(256) sdpa_forw_kernel(const sdpa_geom_t g, const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, const float* __restrict__ mask, float* __restrict__ o, float* __restrict__ lse)
{
	__shared__ float Qs[BR][DMAX];
	__shared__ float Ks[BC][DMAX + 1];
	__shared__ float Vs[BC][DMAX];
	__shared__ float Ps[BR][BC];
	constexpr int NJ = BC / 16, ND = DMAX / 16;
	const int t = threadIdx.x, r = t >> 4, lane = t & 15;
	const int x0 = blockIdx.x * BR, h = blockIdx.y, b = blockIdx.z, hk = h / g.ratio;
	const int x = x0 + r;
	for (int i = t; i < BR * g.D; i += 256) { const int rr = i / g.D, d = i - rr * g.D; Qs[rr][d] = x0 + rr < g.R ? q[b * g.q_sb + (long)(x0 + rr) * g.q_sr + h * g.q_sh + d] : 0.f; }
	float acc[ND];
#pragma unroll
	for (int i = 0; i < ND; i++) acc[i] = 0.f;
	float m_run = -INFINITY, l_run = 0.f;
	const int vis = x < g.R ? visible_keys(g, x) : 0;
	// the furthest key any row of this block sees
	int vis_max = 0;
	for (int rr = 0; rr < BR; rr++) if (x0 + rr < g.R) { const int e = visible_keys(g, x0 + rr); vis_max = e > vis_max ? e : vis_max; }
	for (int y0 = 0; y0 < vis_max; y0 += BC) {
		__syncthreads();
		for (int i = t; i < BC * g.D; i += 256) { const int j = i / g.D, d = i - j * g.D; Ks[j][d] = y0 + j < g.C ? k[b * g.k_sb + (long)(y0 + j) * g.k_sc + hk * g.k_sh + d] : 0.f; }
		for (int i = t; i < BC * g.Dv; i += 256) { const int j = i / g.Dv, d = i - j * g.Dv; Vs[j][d] = y0 + j < g.C ? v[b * g.v_sb + (long)(y0 + j) * g.v_sc + hk * g.v_sh + d] : 0.f; }
		__syncthreads();
		float s[NJ];
#pragma unroll
		for (int i = 0; i < NJ; i++) s[i] = 0.f;
		for (int d = 0; d < g.D; d++) {
			const float qv = Qs[r][d];
#pragma unroll
			for (int i = 0; i < NJ; i++) s[i] += qv * Ks[lane + 16 * i][d];
		}
		float bm = -INFINITY;
#pragma unroll
		for (int i = 0; i < NJ; i++) {
			const int y = y0 + lane + 16 * i;
			if (y < vis) {
				s[i] = g.scale * s[i] + (mask ? mask[b * g.m_sb + h * g.m_sh + (long)x * g.m_sr + y] : 0.f);
				bm = fmaxf(bm, s[i]);
			} else s[i] = -INFINITY;
		}
		bm = group16_max(bm);
		const float m_new = fmaxf(m_run, bm);
		float ps = 0.f;
#pragma unroll
		for (int i = 0; i < NJ; i++) {
			const float p = s[i] == -INFINITY ? 0.f : expf(s[i] - m_new);
			Ps[r][lane + 16 * i] = p;
			ps += p;
		}
		ps = group16_sum(ps);
		const float alpha = m_run == -INFINITY ? 0.f : expf(m_run - m_new);
		l_run = l_run * alpha + ps;
		m_run = m_new;
		__syncthreads();
#pragma unroll
		for (int i = 0; i < ND; i++) acc[i] *= alpha;
		for (int j = 0; j < BC; j++) {
			const float p = Ps[r][j];
#pragma unroll
			for (int i = 0; i < ND; i++) acc[i] += p * Vs[j][lane + 16 * i];
		}
	}
	if (x < g.R) {
		const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
#pragma unroll
		for (int i = 0; i < ND; i++) { const int d = lane + 16 * i; if (d < g.Dv) o[b * g.o_sb + (long)x * g.o_sr + h * g.o_sh + d] = acc[i] * inv; }
		if (lse && lane == 0) lse[((long)b * g.Hq + h) * g.R + x] = l_run > 0.f ? m_run + logf(l_run) : -INFINITY;
	}
}

// delta[b][h][x] = sum_d g[b, x, h, d] * o[b, x, h, d]   (o: the scratch copy, dense [B][R][Hq][Dv])
__global__ void __launch_bounds__(256) sdpa_delta_kernel(const sdpa_geom_t g, const float* __restrict__ gr, const long g_sb, const long g_sr, const long g_sh, const float* __restrict__ o, float* __restrict__ delta)
{
	const long n = (long)g.B * g.Hq * g.R;
	for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
		const int x = (int)(i % g.R), h = (int)((i / g.R) % g.Hq), b = (int)(i / ((long)g.R * g.Hq));
		const float* const gp = gr + b * g_sb + (long)x * g_sr + h * g_sh;
		const float* const op = o + (((long)b * g.R + x) * g.Hq + h) * g.Dv;
		float s = 0.f;
		for (int d = 0; d < g.Dv; d++) s += gp[d] * op[d];
		delta[i] = s;
	}
}

// ---- dq -------------------------------------------------------------------------------------------------------------------------
template <int DMAX, int BC>
__global__ void __launch_bounds__(256) sdpa_dq_kernel(const sdpa_geom_t g, const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, const float* __restrict__ mask, const float* __restrict__ gr, const long g_sb, const long g_sr, const long g_sh, const float* __restrict__ lse, const float* __restrict__ delta, float* __restrict__ dq, const long dq_sb, const long dq_sr, const long dq_sh)
{
	__shared__ float Qs[BR][DMAX];
	__shared__ float Gs[BR][DMAX];
	__shared__ float Ks[BC][DMAX + 1];
	__shared__ float Vs[BC][DMAX + 1];
	__shared__ float Ps[BR][BC];
	constexpr int NJ = BC / 16, ND = DMAX / 16;
	const int t = threadIdx.x, r = t >> 4, lane = t & 15;
	const int x0 = blockIdx.x * BR, h = blockIdx.y, b = blockIdx.z, hk = h / g.ratio;
	const int x = x0 + r;
	for (int i = t; i < BR * g.D; i += 256) { const int rr = i / g.D, d = i - rr * g.D; Qs[rr][d] = x0 + rr < g.R ? q[b * g.q_sb + (long)(x0 + rr) * g.q_sr + h * g.q_sh + d] : 0.f; }
	for (int i = t; i < BR * g.Dv; i += 256) { const int rr = i / g.Dv, d = i - rr * g.Dv; Gs[rr][d] = x0 + rr < g.R ? gr[b * g_sb + (long)(x0 + rr) * g_sr + h * g_sh + d] : 0.f; }
	float acc[ND];
#pragma unroll
	for (int i = 0; i < ND; i++) acc[i] = 0.f;
	const int vis = x < g.R ? visible_keys(g, x) : 0;
	const float my_lse = x < g.R ? lse[((long)b * g.Hq + h) * g.R + x] : 0.f, my_delta = x < g.R ? delta[((long)b * g.Hq + h) * g.R + x] : 0.f;
	int vis_max = 0;
	for (int rr = 0; rr < BR; rr++) if (x0 + rr < g.R) { const int e = visible_keys(g, x0 + rr); vis_max = e > vis_max ? e : vis_max; }
	for (int y0 = 0; y0 < vis_max; y0 += BC) {
		__syncthreads();
		for (int i = t; i < BC * g.D; i += 256) { const int j = i / g.D, d = i - j * g.D; Ks[j][d] = y0 + j < g.C ? k[b * g.k_sb + (long)(y0 + j) * g.k_sc + hk * g.k_sh + d] : 0.f; }
		for (int i = t; i < BC * g.Dv; i += 256) { const int j = i / g.Dv, d = i - j * g.Dv; Vs[j][d] = y0 + j < g.C ? v[b * g.v_sb + (long)(y0 + j) * g.v_sc + hk * g.v_sh + d] : 0.f; }
		__syncthreads();
		float s[NJ], dp[NJ];
#pragma unroll
		for (int i = 0; i < NJ; i++) { s[i] = 0.f; dp[i] = 0.f; }
		for (int d = 0; d < g.D; d++) {
			const float qv = Qs[r][d];
#pragma unroll
			for (int i = 0; i < NJ; i++) s[i] += qv * Ks[lane + 16 * i][d];
		}
		for (int d = 0; d < g.Dv; d++) {
			const float gv = Gs[r][d];
#pragma unroll
			for (int i = 0; i < NJ; i++) dp[i] += gv * Vs[lane + 16 * i][d];
		}
#pragma unroll
		for (int i = 0; i < NJ; i++) {
			const int y = y0 + lane + 16 * i;
			float ds = 0.f;
			if (y < vis) {
				const float sc = g.scale * s[i] + (mask ? mask[b * g.m_sb + h * g.m_sh + (long)x * g.m_sr + y] : 0.f);
				const float p = expf(sc - my_lse);
				ds = p * (dp[i] - my_delta);
			}
			Ps[r][lane + 16 * i] = ds;
		}
		__syncthreads();
		for (int j = 0; j < BC; j++) {
			const float ds = Ps[r][j];
#pragma unroll
			for (int i = 0; i < ND; i++) acc[i] += ds * Ks[j][lane + 16 * i];
		}
	}
	if (x < g.R)
#pragma unroll
		for (int i = 0; i < ND; i++) { const int d = lane + 16 * i; if (d < g.D) dq[b * dq_sb + (long)x * dq_sr + h * dq_sh + d] = g.scale * acc[i]; }
}

// ---- dk, dv ---------------------------------------------------------------------------------------------------------------------
template <int DMAX>
__global__ void __launch_bounds__(256) sdpa_dkv_kernel(const sdpa_geom_t g, const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, const float* __restrict__ mask, const float* __restrict__ gr, const long g_sb, const long g_sr, const long g_sh, const float* __restrict__ lse, const float* __restrict__ delta, float* __restrict__ dk, const long dk_sb, const long dk_sc, const long dk_sh, float* __restrict__ dv, const long dv_sb, const long dv_sc, const long dv_sh)
{
	__shared__ float Ks[BR][DMAX];
	__shared__ float Vs[BR][DMAX];
	__shared__ float Qs[BR][DMAX + 1];
	__shared__ float Gs[BR][DMAX + 1];
	__shared__ float Pt[BR][BR];
	__shared__ float St[BR][BR];
	__shared__ float Ls[BR], Ds[BR];
	constexpr int ND = DMAX / 16;
	const int t = threadIdx.x, yy = t >> 4, lane = t & 15;
	const int y0 = blockIdx.x * BR, hk = blockIdx.y, b = blockIdx.z;
	const int y = y0 + yy;
	for (int i = t; i < BR * g.D; i += 256) { const int j = i / g.D, d = i - j * g.D; Ks[j][d] = y0 + j < g.C ? k[b * g.k_sb + (long)(y0 + j) * g.k_sc + hk * g.k_sh + d] : 0.f; }
	for (int i = t; i < BR * g.Dv; i += 256) { const int j = i / g.Dv, d = i - j * g.Dv; Vs[j][d] = y0 + j < g.C ? v[b * g.v_sb + (long)(y0 + j) * g.v_sc + hk * g.v_sh + d] : 0.f; }
	float ak[ND], av[ND];
#pragma unroll
	for (int i = 0; i < ND; i++) { ak[i] = 0.f; av[i] = 0.f; }
	// first query row that can see key y0 (causal): x - R + C + 1 > y0  <=>  x > y0 + R - C - 1
	int xs = 0;
	if (g.causal) { xs = y0 + g.R - g.C; if (xs < 0) xs = 0; xs = xs / BR * BR; }
	for (int h = hk * g.ratio; h < (hk + 1) * g.ratio; h++)
		for (int x0 = xs; x0 < g.R; x0 += BR) {
			__syncthreads();
			for (int i = t; i < BR * g.D; i += 256) { const int rr = i / g.D, d = i - rr * g.D; Qs[rr][d] = x0 + rr < g.R ? q[b * g.q_sb + (long)(x0 + rr) * g.q_sr + h * g.q_sh + d] : 0.f; }
			for (int i = t; i < BR * g.Dv; i += 256) { const int rr = i / g.Dv, d = i - rr * g.Dv; Gs[rr][d] = x0 + rr < g.R ? gr[b * g_sb + (long)(x0 + rr) * g_sr + h * g_sh + d] : 0.f; }
			if (t < BR) { const int x = x0 + t; Ls[t] = x < g.R ? lse[((long)b * g.Hq + h) * g.R + x] : 0.f; Ds[t] = x < g.R ? delta[((long)b * g.Hq + h) * g.R + x] : 0.f; }
			__syncthreads();
			// thread (key yy, query lane): one score
			{
				const int x = x0 + lane;
				float s = 0.f, dp = 0.f;
				for (int d = 0; d < g.D; d++) s += Ks[yy][d] * Qs[lane][d];
				for (int d = 0; d < g.Dv; d++) dp += Vs[yy][d] * Gs[lane][d];
				float p = 0.f, ds = 0.f;
				if (x < g.R && y < g.C && y < visible_keys(g, x)) {
					const float sc = g.scale * s + (mask ? mask[b * g.m_sb + h * g.m_sh + (long)x * g.m_sr + y] : 0.f);
					p = expf(sc - Ls[lane]);
					ds = p * (dp - Ds[lane]);
				}
				Pt[yy][lane] = p;
				St[yy][lane] = ds;
			}
			__syncthreads();
			for (int xx = 0; xx < BR; xx++) {
				const float p = Pt[yy][xx], ds = St[yy][xx];
#pragma unroll
				for (int i = 0; i < ND; i++) { av[i] += p * Gs[xx][lane + 16 * i]; ak[i] += ds * Qs[xx][lane + 16 * i]; }
			}
		}
	if (y < g.C) {
#pragma unroll
		for (int i = 0; i < ND; i++) {
			const int d = lane + 16 * i;
			if (dk && d < g.D) dk[b * dk_sb + (long)y * dk_sc + hk * dk_sh + d] = g.scale * ak[i];
			if (dv && d < g.Dv) dv[b * dv_sb + (long)y * dv_sc + hk * dv_sh + d] = av[i];
		}
	}
}

// ---- host -----------------------------------------------------------------------------------------------------------------------
struct bhd_t { int b, n, h, d; long sb, sn, sh; };
static bool bhd(const ccv_nnc_tensor_t* t, bhd_t* o)
{ // [B, N, H, D] or [B, N, D]; D contiguous
	const int nd = tensor_nd(t->info.dim);
	if ((nd != 3 && nd != 4) || CCV_GET_DATA_TYPE(t->info.datatype) != CCV_32F) return false;
	int st[CCV_NNC_MAX_DIM_ALLOC];
	tensor_strides(t, st);
	if (st[nd - 1] != 1) return false;
	o->b = t->info.dim[0]; o->n = t->info.dim[1]; o->sb = st[0]; o->sn = st[1];
	if (nd == 4) { o->h = t->info.dim[2]; o->d = t->info.dim[3]; o->sh = st[2]; }
	else { o->h = 1; o->d = t->info.dim[2]; o->sh = 0; }
	return true;
}
static bool sdpa_geometry(const ccv_nnc_cmd_t& cmd, const ccv_nnc_tensor_t* q, const ccv_nnc_tensor_t* k, const ccv_nnc_tensor_t* v, const ccv_nnc_tensor_t* mask, sdpa_geom_t* g, bhd_t* qi, bhd_t* ki, bhd_t* vi)
{
	if (!bhd(q, qi) || !bhd(k, ki) || !bhd(v, vi)) return false;
	if (tensor_nd(q->info.dim) != tensor_nd(k->info.dim) || tensor_nd(k->info.dim) != tensor_nd(v->info.dim)) return false;
	if (qi->b != ki->b || ki->b != vi->b || qi->d != ki->d || ki->n != vi->n || ki->h != vi->h || qi->h < ki->h || qi->h % ki->h) return false;
	g->B = qi->b; g->R = qi->n; g->C = ki->n; g->Hq = qi->h; g->Hk = ki->h; g->D = qi->d; g->Dv = vi->d; g->ratio = qi->h / ki->h;
	g->q_sb = qi->sb; g->q_sr = qi->sn; g->q_sh = qi->sh; g->k_sb = ki->sb; g->k_sc = ki->sn; g->k_sh = ki->sh; g->v_sb = vi->sb; g->v_sc = vi->sn; g->v_sh = vi->sh;
	g->m_sb = g->m_sh = g->m_sr = 0;
	if (mask) { // [B or 1][Hq or 1][R][C], or 3-d [B or 1][R][C]
		const int nd = tensor_nd(mask->info.dim);
		if ((nd != 3 && nd != 4) || CCV_GET_DATA_TYPE(mask->info.datatype) != CCV_32F) return false;
		int st[CCV_NNC_MAX_DIM_ALLOC];
		tensor_strides(mask, st);
		if (st[nd - 1] != 1 || mask->info.dim[nd - 1] != g->C || mask->info.dim[nd - 2] != g->R) return false;
		g->m_sr = st[nd - 2];
		const int mb = mask->info.dim[0], mh = nd == 4 ? mask->info.dim[1] : 1;
		if ((mb != 1 && mb != g->B) || (mh != 1 && mh != g->Hq)) return false;
		g->m_sb = mb == 1 ? 0 : st[0];
		g->m_sh = (nd == 4 && mh != 1) ? st[1] : 0;
	}
	g->scale = cmd.info.scaled_dot_product_attention.scale;
	g->causal = cmd.info.scaled_dot_product_attention.is_causal;
	return g->D >= 1 && g->Dv >= 1 && g->D <= 256 && g->Dv <= 256;
}
static int sdpa_forward_launch(const sdpa_geom_t& g, const float* q, const float* k, const float* v, const float* mask, float* o, float* lse, hipStream_t stream)
{
	const dim3 grid((g.R + BR - 1) / BR, g.Hq, g.B);
	if (!g.R || !g.Hq || !g.B) return CCV_NNC_EXEC_SUCCESS;
	const int dm = g.D > g.Dv ? g.D : g.Dv;
	if (dm <= 64) hipLaunchKernelGGL(HIP_KERNEL_NAME(sdpa_forw_kernel<64, 64>), grid, dim3(256), 0, stream, g, q, k, v, mask, o, lse);
	else if (dm <= 128) hipLaunchKernelGGL(HIP_KERNEL_NAME(sdpa_forw_kernel<128, 32>), grid, dim3(256), 0, stream, g, q, k, v, mask, o, lse);
	else hipLaunchKernelGGL(HIP_KERNEL_NAME(sdpa_forw_kernel<256, 16>), grid, dim3(256), 0, stream, g, q, k, v, mask, o, lse);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}

static int _sdpa_forw(EXEC_ARGS)
{ // inputs (q, k, v, [mask], [w], [bias]); outputs (o, [lse], [the attention output before the projection])
	if (input_size < 3 || output_size < 1 || !inputs[0] || !inputs[1] || !inputs[2]) return CCV_NNC_EXEC_INVALID;
	const ccv_nnc_tensor_t* mask = input_size > 3 ? inputs[3] : 0;
	const ccv_nnc_tensor_t* w = input_size > 4 ? inputs[4] : 0;
	const ccv_nnc_tensor_t* bias = input_size > 5 ? inputs[5] : 0;
	if (bias && !w) return CCV_NNC_EXEC_INVALID;
	ccv_nnc_tensor_t* const c = w ? (output_size > 2 ? outputs[2] : 0) : outputs[0];
	ccv_nnc_tensor_t* const lse_t = output_size > 1 ? outputs[1] : 0;
	if (!c) return CCV_NNC_EXEC_INVALID;
	sdpa_geom_t g;
	bhd_t qi, ki, vi, ci;
	if (!sdpa_geometry(cmd, inputs[0], inputs[1], inputs[2], mask, &g, &qi, &ki, &vi) || !bhd(c, &ci)) return CCV_NNC_EXEC_INVALID;
	if (ci.b != g.B || ci.n != g.R || ci.h != g.Hq || ci.d != g.Dv) return CCV_NNC_EXEC_INVALID;
	g.o_sb = ci.sb; g.o_sr = ci.sn; g.o_sh = ci.sh;
	float* lse = 0;
	if (lse_t) {
		if (CCV_GET_DATA_TYPE(lse_t->info.datatype) != CCV_32F || !tensor_contiguous(lse_t) || tensor_count(lse_t->info) != (size_t)g.B * g.Hq * g.R) return CCV_NNC_EXEC_INVALID;
		lse = lse_t->data.f32;
	}
	int ret = sdpa_forward_launch(g, inputs[0]->data.f32, inputs[1]->data.f32, inputs[2]->data.f32, mask ? mask->data.f32 : 0, c->data.f32, lse, stream_of(stream_context));
	if (ret != CCV_NNC_EXEC_SUCCESS || !w) return ret;
	// unify heads: d[B * R, E] = c[B * R, E] w[E, E]^T (+ bias), E = Hq * Dv  (cpu_ref.c:185-252)
	ccv_nnc_tensor_t* const d = outputs[0];
	const int E = g.Hq * g.Dv;
	if (!d || !tensor_contiguous(c) || !tensor_contiguous(d) || !tensor_contiguous(w) || tensor_nd(w->info.dim) != 2 || w->info.dim[0] != E || w->info.dim[1] != E) return CCV_NNC_EXEC_INVALID;
	if (tensor_count(d->info) != (size_t)g.B * g.R * E || (bias && tensor_count(bias->info) != (size_t)E)) return CCV_NNC_EXEC_INVALID;
	ccv_nnc_tensor_t a2 = *c, d2 = *d;
	memset(a2.info.dim, 0, sizeof(a2.info.dim)); memset(d2.info.dim, 0, sizeof(d2.info.dim));
	a2.info.dim[0] = g.B * g.R; a2.info.dim[1] = E; d2.info.dim[0] = g.B * g.R; d2.info.dim[1] = E;
	a2.type &= ~CCV_TENSOR_VIEW; d2.type &= ~CCV_TENSOR_VIEW;
	ccv_nnc_cmd_t gemm;
	memset(&gemm, 0, sizeof(gemm));
	gemm.cmd = CCV_NNC_GEMM_FORWARD; gemm.backend = CCV_NNC_NO_BACKEND; gemm.algorithm = -1;
	gemm.info.blas.a[0] = 1; gemm.info.blas.a[1] = 1;
	gemm.info.blas.transpose_b[0] = 0; gemm.info.blas.transpose_b[1] = 1;
	ccv_nnc_tensor_t* gin[3] = { &a2, (ccv_nnc_tensor_t*)w, (ccv_nnc_tensor_t*)bias };
	ccv_nnc_tensor_t* gout[1] = { &d2 };
	ccv_nnc_hint_t no_hint;
	memset(&no_hint, 0, sizeof(no_hint));
	return nnc_mi355x_cmd_exec(gemm, no_hint, 0, gin, bias ? 3 : 2, gout, 1, stream_context);
}

static int _sdpa_back(EXEC_ARGS)
{ // inputs (g, ., ., q, k, v, [mask], [w], [bias], [y], [lse], [qkv]); outputs (dq, dk, dv, ...)
	if (input_size < 6 || output_size < 3 || !inputs[0] || !inputs[3] || !inputs[4] || !inputs[5]) return CCV_NNC_EXEC_INVALID;
	const ccv_nnc_tensor_t* mask = input_size > 6 ? inputs[6] : 0;
	if (input_size > 7 && inputs[7]) return CCV_NNC_EXEC_INVALID; // the head-unifying projection has no backward here (nor in the backend replaced, flash_attn.cu:246-248)
	ccv_nnc_tensor_t* const dq = outputs[0];
	ccv_nnc_tensor_t* const dk = outputs[1];
	ccv_nnc_tensor_t* const dv = outputs[2];
	sdpa_geom_t g;
	bhd_t qi, ki, vi, gi, dqi, dki, dvi;
	if (!sdpa_geometry(cmd, inputs[3], inputs[4], inputs[5], mask, &g, &qi, &ki, &vi) || !bhd(inputs[0], &gi)) return CCV_NNC_EXEC_INVALID;
	if (gi.b != g.B || gi.n != g.R || gi.h != g.Hq || gi.d != g.Dv) return CCV_NNC_EXEC_INVALID;
	if (dq && (!bhd(dq, &dqi) || dqi.b != g.B || dqi.n != g.R || dqi.h != g.Hq || dqi.d != g.D)) return CCV_NNC_EXEC_INVALID;
	if (dk && (!bhd(dk, &dki) || dki.b != g.B || dki.n != g.C || dki.h != g.Hk || dki.d != g.D)) return CCV_NNC_EXEC_INVALID;
	if (dv && (!bhd(dv, &dvi) || dvi.b != g.B || dvi.n != g.C || dvi.h != g.Hk || dvi.d != g.Dv)) return CCV_NNC_EXEC_INVALID;
	if (!g.B || !g.R || !g.C) return CCV_NNC_EXEC_SUCCESS;
	const size_t rows = (size_t)g.B * g.Hq * g.R;
	const size_t o_bytes = (sizeof(float) * rows * g.Dv + 255) & ~(size_t)255, r_bytes = (sizeof(float) * rows + 255) & ~(size_t)255;
	char* const ws = (char*)workspace_of(stream_context, o_bytes + 2 * r_bytes);
	if (!ws) return CCV_NNC_EXEC_OOM;
	float* const o = (float*)ws; float* const lse = (float*)(ws + o_bytes); float* const delta = (float*)(ws + o_bytes + r_bytes);
	hipStream_t stream = stream_of(stream_context);
	sdpa_geom_t gf = g;
	gf.o_sb = (long)g.R * g.Hq * g.Dv; gf.o_sr = (long)g.Hq * g.Dv; gf.o_sh = g.Dv;
	const float* const qp = inputs[3]->data.f32; const float* const kp = inputs[4]->data.f32; const float* const vp = inputs[5]->data.f32;
	const float* const mp = mask ? mask->data.f32 : 0; const float* const gp = inputs[0]->data.f32;
	int ret = sdpa_forward_launch(gf, qp, kp, vp, mp, o, lse, stream);
	if (ret != CCV_NNC_EXEC_SUCCESS) return ret;
	hipLaunchKernelGGL(sdpa_delta_kernel, dim3(grid_for(rows, 256)), dim3(256), 0, stream, g, gp, gi.sb, gi.sn, gi.sh, (const float*)o, delta);
	HIP_ENFORCE(hipGetLastError());
	const int dm = g.D > g.Dv ? g.D : g.Dv;
	if (dq) {
		const dim3 grid((g.R + BR - 1) / BR, g.Hq, g.B);
		if (dm <= 64) hipLaunchKernelGGL(HIP_KERNEL_NAME(sdpa_dq_kernel<64, 64>), grid, dim3(256), 0, stream, g, qp, kp, vp, mp, gp, gi.sb, gi.sn, gi.sh, (const float*)lse, (const float*)delta, dq->data.f32, dqi.sb, dqi.sn, dqi.sh);
		else if (dm <= 128) hipLaunchKernelGGL(HIP_KERNEL_NAME(sdpa_dq_kernel<128, 32>), grid, dim3(256), 0, stream, g, qp, kp, vp, mp, gp, gi.sb, gi.sn, gi.sh, (const float*)lse, (const float*)delta, dq->data.f32, dqi.sb, dqi.sn, dqi.sh);
		else hipLaunchKernelGGL(HIP_KERNEL_NAME(sdpa_dq_kernel<256, 16>), grid, dim3(256), 0, stream, g, qp, kp, vp, mp, gp, gi.sb, gi.sn, gi.sh, (const float*)lse, (const float*)delta, dq->data.f32, dqi.sb, dqi.sn, dqi.sh);
		HIP_ENFORCE(hipGetLastError());
	}
	if (dk || dv) {
		const dim3 grid((g.C + BR - 1) / BR, g.Hk, g.B);
		float* const dkp = dk ? dk->data.f32 : 0; float* const dvp = dv ? dv->data.f32 : 0;
		const long ksb = dk ? dki.sb : 0, ksn = dk ? dki.sn : 0, ksh = dk ? dki.sh : 0, vsb = dv ? dvi.sb : 0, vsn = dv ? dvi.sn : 0, vsh = dv ? dvi.sh : 0;
		if (dm <= 64) hipLaunchKernelGGL(HIP_KERNEL_NAME(sdpa_dkv_kernel<64>), grid, dim3(256), 0, stream, g, qp, kp, vp, mp, gp, gi.sb, gi.sn, gi.sh, (const float*)lse, (const float*)delta, dkp, ksb, ksn, ksh, dvp, vsb, vsn, vsh);
		else if (dm <= 128) hipLaunchKernelGGL(HIP_KERNEL_NAME(sdpa_dkv_kernel<128>), grid, dim3(256), 0, stream, g, qp, kp, vp, mp, gp, gi.sb, gi.sn, gi.sh, (const float*)lse, (const float*)delta, dkp, ksb, ksn, ksh, dvp, vsb, vsn, vsh);
		else hipLaunchKernelGGL(HIP_KERNEL_NAME(sdpa_dkv_kernel<256>), grid, dim3(256), 0, stream, g, qp, kp, vp, mp, gp, gi.sb, gi.sn, gi.sh, (const float*)lse, (const float*)delta, dkp, ksb, ksn, ksh, dvp, vsb, vsn, vsh);
		HIP_ENFORCE(hipGetLastError());
	}
	return CCV_NNC_EXEC_SUCCESS;
}

} // namespace

#define NNC_REG(CMD, BACKEND, EXEC) \
	extern "C" void _register_command_##CMD##_backend_##BACKEND(ccv_nnc_cmd_backend_registry_t* const registry) \
	{ registry->tensor_formats = CCV_TENSOR_FORMAT_NCHW | CCV_TENSOR_FORMAT_NHWC; registry->tensor_datatypes = CCV_32F; registry->tensor_memory = CCV_TENSOR_GPU_MEMORY; registry->algorithms = 1; registry->exec = EXEC; NNC_HALF_STAGED(registry, EXEC); }

NNC_REG(CCV_NNC_SCALED_DOT_PRODUCT_ATTENTION_FORWARD, CCV_NNC_BACKEND_GPU_REF, _sdpa_forw)
NNC_REG(CCV_NNC_SCALED_DOT_PRODUCT_ATTENTION_BACKWARD, CCV_NNC_BACKEND_GPU_REF, _sdpa_back)
