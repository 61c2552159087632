// SCALED_DOT_PRODUCT_ATTENTION forward / backward (the NLP trainers' row).
// Oracle semantics: lib/nnc/cmd/scaled_dot_product_attention/ccv_nnc_scaled_dot_product_attention_cpu_ref.c:16-497;
// replaces .../gpu/ccv_nnc_scaled_dot_product_attention_flash_attn.cu:18-470.
//   q [B, R, Hq, D], k [B, C, Hk, D], v [B, C, Hk, Dv] (3-d tensors: one head), Hq a multiple of Hk (grouped-query heads);
//   o[b, x, h, :] = sum_y softmax_y(scale * q[b, x, h, :] . k[b, y, h / (Hq / Hk), :] + mask[b, h, x, y]) v[b, y, ., :]
//   causal: row x sees the keys y < x - R + C + 1 (the bottom-right aligned triangle, cpu_ref.c:143); a row that sees none gives 0;
//   optional "unify heads" projection: d = o (as [B * R, Hq * Dv]) w^T + bias (forward only, like the backend being replaced).
// The VALU kernel set (the fallback for shapes the matrix-core kernels further down do not take; half tensors go through half_stage.cpp's fp32 images), written for
// correctness and determinism, streaming K / V blocks through LDS with the running-maximum softmax so no [R, C] score matrix
// exists in memory.  Work split:
//   forward   a workgroup per (16 query rows, head, batch): thread (row = t / 16, lane = t % 16) owns the scores of keys
//             lane, lane + 16, ... of a key block, then the output columns lane, lane + 16, ... of its row
//   backward  the forward pass again into scratch (output + log-sum-exp per row), delta[x] = g[x] . o[x];
//             dq: the forward's split, recomputing p = exp(s - lse) per key block;
//             dk, dv: a workgroup per (16 keys, KEY head, batch) walks every query block of every query head that shares the
//             key head and accumulates its 16 rows of dk / dv in registers -- no atomics, one fixed summation order.
#include "common.h"
#include "isa.h" // half_t, halfx8, nnc_mfma_f16
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

// ---- forward on the matrix cores (round 3) -----------------------------------------------------------------------------------------
// fp32 in, fp32 accumulate on v_mfma_f32_32x32x2_f32 (the exact fmaf chain of mfma_gemm.h), the running-maximum softmax kept.  A wave owns 32 query rows, a
// workgroup (4 waves) 128; key blocks of 32 stream through LDS, shared by the four waves.  Both products are computed TRANSPOSED so that the softmax never
// crosses lanes except for one half-wave swap:
//   S^T [32 keys x 32 rows] = K Q^T      A = K tile out of LDS (lane: key l & 31, its half's D / 2 values as 16-byte reads), B = Q^T held in registers for the
//                                        whole kernel (lane: row l & 31, D / 2 values); MFMA i contracts d = i (lanes 0-31) and d = D / 2 + i (lanes 32-63)
//   D layout: lane (row = l & 31, half = l >> 5) holds the keys ky(r) = (r & 3) + 8 (r >> 2) + 4 half of ITS row: maximum and sum are 16 registers + one swap
//   O^T [Dv x 32 rows] += V^T P^T        MFMA j contracts the keys ky(j) (lanes 0-31) and ky(j) + 4 (lanes 32-63): its B operand IS register j of P^T -- the
//                                        probabilities never leave the registers they were computed in; A = V[key][dv = l & 31] out of LDS
// Conditions (sdpa_forward_launch): D % 8 == 0, Dv % 32 == 0, both <= 128, 16-byte aligned rows; everything else takes the kernel above.
template <int DH, int TV> // DH = D / 2 rounded up to the instantiation (32 or 64), TV = Dv / 32 (1 .. 4)
__global__ void __launch_bounds__(256) sdpa_forw_mfma_kernel(const sdpa_geom_t g, const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, const float* __restrict__ mask, float* __restrict__ o, float* __restrict__ lse)
{
	constexpr int DP = 2 * DH, KP = DP + 4, VP = 32 * TV + 4; // LDS row pitches (floats)
	__shared__ __attribute__((aligned(16))) float Ks[32 * KP];
	__shared__ __attribute__((aligned(16))) float Vs[32 * VP];
	typedef float floatx16 __attribute__((ext_vector_type(16)));
	const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
	const int li = lane & 31, lh = lane >> 5;
	const int h = blockIdx.y, b = blockIdx.z, hk = h / g.ratio;
	const int xw = blockIdx.x * 128 + wave * 32; // the wave's first query row
	const int x = xw + li;
	const int dh = g.D >> 1; // this half's share of d: [lh * dh, lh * dh + dh)
	float qreg[DH];
#pragma unroll
	for (int i = 0; i < DH; i += 4) {
		float4 qv = make_float4(0.f, 0.f, 0.f, 0.f);
		if (x < g.R && i < dh) qv = *(const float4*)(q + b * g.q_sb + (long)x * g.q_sr + h * g.q_sh + lh * dh + i);
		qreg[i] = qv.x; qreg[i + 1] = qv.y; qreg[i + 2] = qv.z; qreg[i + 3] = qv.w;
	}
	floatx16 acc[TV];
#pragma unroll
	for (int tv = 0; tv < TV; tv++)
#pragma unroll
		for (int r = 0; r < 16; r++) acc[tv][r] = 0.f;
	float m_run = -INFINITY, l_run = 0.f;
	const int vis = x < g.R ? visible_keys(g, x) : 0;
	int vis_max = 0; // the furthest key any row of this WORKGROUP sees (the key loop is shared: barriers inside)
	{
		const int x_last = blockIdx.x * 128 + 127 < g.R ? blockIdx.x * 128 + 127 : g.R - 1;
		vis_max = visible_keys(g, x_last); // visible_keys is monotone in x
	}
	const float* const mrow = mask ? mask + b * g.m_sb + h * g.m_sh + (long)(x < g.R ? x : 0) * g.m_sr : 0;
	for (int y0 = 0; y0 < vis_max; y0 += 32) {
		__syncthreads();
		// the K and V tiles: 16-byte chunks, rows beyond C read as zeros
		for (int c = t; c < 32 * (g.D >> 2); c += 256) {
			const int j = c / (g.D >> 2), d = (c - j * (g.D >> 2)) << 2;
			*(float4*)(Ks + j * KP + d) = y0 + j < g.C ? *(const float4*)(k + b * g.k_sb + (long)(y0 + j) * g.k_sc + hk * g.k_sh + d) : make_float4(0.f, 0.f, 0.f, 0.f);
		}
		for (int c = t; c < 32 * (g.Dv >> 2); c += 256) {
			const int j = c / (g.Dv >> 2), d = (c - j * (g.Dv >> 2)) << 2;
			*(float4*)(Vs + j * VP + d) = y0 + j < g.C ? *(const float4*)(v + b * g.v_sb + (long)(y0 + j) * g.v_sc + hk * g.v_sh + d) : make_float4(0.f, 0.f, 0.f, 0.f);
		}
		__syncthreads();
		floatx16 s;
#pragma unroll
		for (int r = 0; r < 16; r++) s[r] = 0.f;
		const float* const krow = Ks + li * KP + lh * dh;
#pragma unroll
		for (int i = 0; i < DH; i += 4) {
			if (i < dh) {
				const float4 kv = *(const float4*)(krow + i);
				s = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.x, qreg[i], s, 0, 0, 0);
				s = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.y, qreg[i + 1], s, 0, 0, 0);
				s = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.z, qreg[i + 2], s, 0, 0, 0);
				s = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.w, qreg[i + 3], s, 0, 0, 0);
			}
		}
		// scores of this lane's row: register r <-> key y0 + (r & 3) + 8 (r >> 2) + 4 lh
		float bm = -INFINITY;
#pragma unroll
		for (int r = 0; r < 16; r++) {
			const int y = y0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
			if (y < vis) {
				s[r] = g.scale * s[r] + (mrow ? mrow[y] : 0.f);
				bm = fmaxf(bm, s[r]);
			} else s[r] = -INFINITY;
		}
		bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
		const float m_new = fmaxf(m_run, bm);
		float ps = 0.f;
#pragma unroll
		for (int r = 0; r < 16; r++) {
			s[r] = s[r] == -INFINITY ? 0.f : expf(s[r] - m_new);
			ps += s[r];
		}
		ps += __shfl_xor(ps, 32, 64);
		const float alpha = m_run == -INFINITY ? 0.f : expf(m_run - m_new);
		l_run = l_run * alpha + ps;
		m_run = m_new;
#pragma unroll
		for (int tv = 0; tv < TV; tv++)
#pragma unroll
			for (int r = 0; r < 16; r++) acc[tv][r] *= alpha;
#pragma unroll
		for (int j = 0; j < 16; j++) {
			const float* const vrow = Vs + ((j & 3) + 8 * (j >> 2) + 4 * lh) * VP + li;
#pragma unroll
			for (int tv = 0; tv < TV; tv++)
				if (tv * 32 < g.Dv) acc[tv] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[32 * tv], s[j], acc[tv], 0, 0, 0);
		}
	}
	if (x < g.R) {
		const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
		float* const orow = o + b * g.o_sb + (long)x * g.o_sr + h * g.o_sh;
#pragma unroll
		for (int tv = 0; tv < TV; tv++)
			if (tv * 32 < g.Dv) {
#pragma unroll
				for (int r4 = 0; r4 < 4; r4++) { // registers 4 r4 .. 4 r4 + 3 are dv = 32 tv + 8 r4 + 4 lh + 0..3
					const int d = 32 * tv + 8 * r4 + 4 * lh;
					orow[d] = acc[tv][4 * r4] * inv; orow[d + 1] = acc[tv][4 * r4 + 1] * inv; orow[d + 2] = acc[tv][4 * r4 + 2] * inv; orow[d + 3] = acc[tv][4 * r4 + 3] * inv;
				}
			}
		if (lse && lh == 0) lse[((long)b * g.Hq + h) * g.R + x] = l_run > 0.f ? m_run + logf(l_run) : -INFINITY;
	}
}

// ---- forward in half precision on the f16 matrix cores (round 4) ---------------------------------------------------------------------
// CCV_16F q / k / v / o (the reference's flash_attn rows, scaled_dot_product_attention/gpu/..._flash_attn.cu): the fp32 kernel's decomposition on
// v_mfma_f32_32x32x16_f16 -- 16 reduction terms per instruction instead of 2, fp32 accumulation, the running-maximum softmax in fp32:
//   S^T [32 keys x 32 rows] = K Q^T      A = 8 halves of the lane's key row out of LDS (one ds_read_b128 per 16 of d), B = 8 halves of the lane's query row, held in
//                                        registers for the whole kernel; lane (l & 31, half l >> 5) stands for d = 16 s + 8 half + 0..7 in both operands
//   O^T [Dv x 32 rows] += V^T P^T        two sub-steps of 16 keys; the lane holds the probabilities of the keys ky(r) = (r & 3) + 8 (r >> 2) + 4 half of its row, so
//                                        sub-step t takes registers 8 t .. 8 t + 7 -- rounded to half, still in place -- as B, and A = the same eight keys of column
//                                        dv = l & 31 out of a TRANSPOSED V tile in LDS ([dv][key]: two 8-byte reads: keys 16 t + 4 half + 0..3 and + 8)
// Conditions: D % 16 == 0, Dv % 32 == 0, both <= 128, 16-byte aligned rows; an additive mask is a CCV_16F tensor too (other shapes, and the head projection, go through
// fp32 images and the fp32 kernels).  The log-sum-exp output, if asked for, is fp32.
template <int DS, int TV> // DS = D / 16 (1 .. 8), TV = Dv / 32 (1 .. 4)
__global__ void __launch_bounds__(256) sdpa_forw_f16_kernel(const sdpa_geom_t g, const half_t* __restrict__ q, const half_t* __restrict__ k, const half_t* __restrict__ v, const half_t* __restrict__ mask, half_t* __restrict__ o, float* __restrict__ lse)
{
	constexpr int KP = 16 * DS + 8, VTP = 40; // halves per row: K tile [32 keys][D + 8], V^T tile [Dv][32 keys + 8]
	__shared__ __attribute__((aligned(16))) half_t Ks[32 * KP];
	__shared__ __attribute__((aligned(16))) half_t Vt[32 * TV * VTP];
	typedef float floatx16 __attribute__((ext_vector_type(16)));
	const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
	const int li = lane & 31, lh = lane >> 5;
	const int h = blockIdx.y, b = blockIdx.z, hk = h / g.ratio;
	const int x = blockIdx.x * 128 + wave * 32 + li;
	halfx8 qf[DS];
#pragma unroll
	for (int s = 0; s < DS; s++) {
		qf[s] = halfx8{ 0, 0, 0, 0, 0, 0, 0, 0 };
		if (x < g.R) qf[s] = *(const halfx8*)(q + b * g.q_sb + (long)x * g.q_sr + h * g.q_sh + 16 * s + 8 * lh);
	}
	floatx16 acc[TV];
#pragma unroll
	for (int tv = 0; tv < TV; tv++)
#pragma unroll
		for (int r = 0; r < 16; r++) acc[tv][r] = 0.f;
	float m_run = -INFINITY, l_run = 0.f;
	const int vis = x < g.R ? visible_keys(g, x) : 0;
	int vis_max = 0;
	{
		const int x_last = blockIdx.x * 128 + 127 < g.R ? blockIdx.x * 128 + 127 : g.R - 1;
		vis_max = visible_keys(g, x_last);
	}
	const half_t* const mrow = mask ? mask + b * g.m_sb + h * g.m_sh + (long)(x < g.R ? x : 0) * g.m_sr : 0;
	for (int y0 = 0; y0 < vis_max; y0 += 32) {
		__syncthreads();
		for (int c = t; c < 32 * 2 * DS; c += 256) { // K: 16-byte chunks as they lie
			const int j = c / (2 * DS), d = (c - j * (2 * DS)) << 3;
			*(halfx8*)(Ks + j * KP + d) = y0 + j < g.C ? *(const halfx8*)(k + b * g.k_sb + (long)(y0 + j) * g.k_sc + hk * g.k_sh + d) : halfx8{ 0, 0, 0, 0, 0, 0, 0, 0 };
		}
		for (int c = t; c < 32 * 4 * TV; c += 256) { // V: a 16-byte chunk of key j lands transposed, one half per dv row
			const int j = c & 31, d = (c >> 5) << 3;
			const halfx8 u = y0 + j < g.C ? *(const halfx8*)(v + b * g.v_sb + (long)(y0 + j) * g.v_sc + hk * g.v_sh + d) : halfx8{ 0, 0, 0, 0, 0, 0, 0, 0 };
#pragma unroll
			for (int e = 0; e < 8; e++) Vt[(d + e) * VTP + j] = u[e];
		}
		__syncthreads();
		floatx16 s;
#pragma unroll
		for (int r = 0; r < 16; r++) s[r] = 0.f;
#pragma unroll
		for (int i = 0; i < DS; i++) s = nnc_mfma_f16(*(const halfx8*)(Ks + li * KP + 16 * i + 8 * lh), qf[i], s);
		// scores of this lane's row: register r <-> key y0 + (r & 3) + 8 (r >> 2) + 4 lh
		float bm = -INFINITY;
#pragma unroll
		for (int r = 0; r < 16; r++) {
			const int y = y0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
			if (y < vis) { s[r] = g.scale * s[r] + (mrow ? (float)mrow[y] : 0.f); bm = fmaxf(bm, s[r]); }
			else s[r] = -INFINITY;
		}
		bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
		const float m_new = fmaxf(m_run, bm);
		float ps = 0.f;
#pragma unroll
		for (int r = 0; r < 16; r++) {
			s[r] = s[r] == -INFINITY ? 0.f : expf(s[r] - m_new);
			ps += s[r];
		}
		ps += __shfl_xor(ps, 32, 64);
		const float alpha = m_run == -INFINITY ? 0.f : expf(m_run - m_new);
		l_run = l_run * alpha + ps;
		m_run = m_new;
#pragma unroll
		for (int tv = 0; tv < TV; tv++)
#pragma unroll
			for (int r = 0; r < 16; r++) acc[tv][r] *= alpha;
#pragma unroll
		for (int tt = 0; tt < 2; tt++) {
			const halfx8 pf = halfx8{ (half_t)s[8 * tt], (half_t)s[8 * tt + 1], (half_t)s[8 * tt + 2], (half_t)s[8 * tt + 3], (half_t)s[8 * tt + 4], (half_t)s[8 * tt + 5], (half_t)s[8 * tt + 6], (half_t)s[8 * tt + 7] };
#pragma unroll
			for (int tv = 0; tv < TV; tv++) {
				const half_t* const vrow = Vt + (32 * tv + li) * VTP + 16 * tt + 4 * lh;
				const halfx4 lo = *(const halfx4*)vrow, hi = *(const halfx4*)(vrow + 8);
				acc[tv] = nnc_mfma_f16(halfx8{ lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3] }, pf, acc[tv]);
			}
		}
	}
	if (x < g.R) {
		const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
		half_t* const orow = o + b * g.o_sb + (long)x * g.o_sr + h * g.o_sh;
#pragma unroll
		for (int tv = 0; tv < TV; tv++)
#pragma unroll
			for (int r4 = 0; r4 < 4; r4++) // registers 4 r4 .. 4 r4 + 3 are dv = 32 tv + 8 r4 + 4 lh + 0..3
				*(halfx4*)(orow + 32 * tv + 8 * r4 + 4 * lh) = halfx4{ (half_t)(acc[tv][4 * r4] * inv), (half_t)(acc[tv][4 * r4 + 1] * inv), (half_t)(acc[tv][4 * r4 + 2] * inv), (half_t)(acc[tv][4 * r4 + 3] * inv) };
		if (lse && lh == 0) lse[((long)b * g.Hq + h) * g.R + x] = l_run > 0.f ? m_run + logf(l_run) : -INFINITY;
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

// ---- backward on the matrix cores (round 4) --------------------------------------------------------------------------------------
// The forward kernel's decomposition, twice more.  dq: a wave owns 32 query rows (their q and g rows in registers for the whole kernel), key blocks of 32 stream
// through LDS; per block  S^T = K Q^T  and  dP^T = V G^T  (A = the K / V tile, lane: key l & 31, its half's values as 16-byte reads; B = the row's registers), so
// lane (row, half) holds p and dp of the keys ky(r) of ITS row:  ds = p (dp - delta)  stays in the registers it was computed in and is the B operand of
// dQ^T [D x 32 rows] += K^T dS^T  (A = K[key ky(j)][d = l & 31] out of the same LDS tile).  p = exp(scale s + mask - lse) from the forward pass's log-sum-exp: no
// running maximum here.  Conditions: D % 32 == 0 (whole output tiles), Dv % 8 == 0, both <= 128, 16-byte rows.
template <int DH, int GH, int TD> // DH / GH = D / 2, Dv / 2 rounded up to 32 or 64; TD = D / 32
__global__ void __launch_bounds__(256) sdpa_dq_mfma_kernel(const sdpa_geom_t g, const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, const float* __restrict__ mask, const float* __restrict__ gr, const long g_sb, const long g_sr, const long g_sh, const float* __restrict__ lse, const float* __restrict__ delta, float* __restrict__ dq, const long dq_sb, const long dq_sr, const long dq_sh)
{
	constexpr int KP = 2 * DH + 4, VP = 2 * GH + 4;
	__shared__ __attribute__((aligned(16))) float Ks[32 * KP];
	__shared__ __attribute__((aligned(16))) float Vs[32 * VP];
	typedef float floatx16 __attribute__((ext_vector_type(16)));
	const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
	const int li = lane & 31, lh = lane >> 5;
	const int h = blockIdx.y, b = blockIdx.z, hk = h / g.ratio;
	const int x = blockIdx.x * 128 + wave * 32 + li;
	const int dh = g.D >> 1, gh = g.Dv >> 1;
	float qreg[DH], greg[GH];
#pragma unroll
	for (int i = 0; i < DH; i += 4) {
		float4 u = make_float4(0.f, 0.f, 0.f, 0.f);
		if (x < g.R && i < dh) u = *(const float4*)(q + b * g.q_sb + (long)x * g.q_sr + h * g.q_sh + lh * dh + i);
		qreg[i] = u.x; qreg[i + 1] = u.y; qreg[i + 2] = u.z; qreg[i + 3] = u.w;
	}
#pragma unroll
	for (int i = 0; i < GH; i += 4) {
		float4 u = make_float4(0.f, 0.f, 0.f, 0.f);
		if (x < g.R && i < gh) u = *(const float4*)(gr + b * g_sb + (long)x * g_sr + h * g_sh + lh * gh + i);
		greg[i] = u.x; greg[i + 1] = u.y; greg[i + 2] = u.z; greg[i + 3] = u.w;
	}
	floatx16 acc[TD];
#pragma unroll
	for (int td = 0; td < TD; td++)
#pragma unroll
		for (int r = 0; r < 16; r++) acc[td][r] = 0.f;
	const int vis = x < g.R ? visible_keys(g, x) : 0;
	const float my_lse = x < g.R ? lse[((long)b * g.Hq + h) * g.R + x] : 0.f, my_delta = x < g.R ? delta[((long)b * g.Hq + h) * g.R + x] : 0.f;
	int vis_max = 0;
	{
		const int x_last = blockIdx.x * 128 + 127 < g.R ? blockIdx.x * 128 + 127 : g.R - 1;
		vis_max = visible_keys(g, x_last);
	}
	const float* const mrow = mask ? mask + b * g.m_sb + h * g.m_sh + (long)(x < g.R ? x : 0) * g.m_sr : 0;
	for (int y0 = 0; y0 < vis_max; y0 += 32) {
		__syncthreads();
		for (int c = t; c < 32 * (g.D >> 2); c += 256) {
			const int j = c / (g.D >> 2), d = (c - j * (g.D >> 2)) << 2;
			*(float4*)(Ks + j * KP + d) = y0 + j < g.C ? *(const float4*)(k + b * g.k_sb + (long)(y0 + j) * g.k_sc + hk * g.k_sh + d) : make_float4(0.f, 0.f, 0.f, 0.f);
		}
		for (int c = t; c < 32 * (g.Dv >> 2); c += 256) {
			const int j = c / (g.Dv >> 2), d = (c - j * (g.Dv >> 2)) << 2;
			*(float4*)(Vs + j * VP + d) = y0 + j < g.C ? *(const float4*)(v + b * g.v_sb + (long)(y0 + j) * g.v_sc + hk * g.v_sh + d) : make_float4(0.f, 0.f, 0.f, 0.f);
		}
		__syncthreads();
		floatx16 s, dp;
#pragma unroll
		for (int r = 0; r < 16; r++) { s[r] = 0.f; dp[r] = 0.f; }
		const float* const krow = Ks + li * KP + lh * dh;
#pragma unroll
		for (int i = 0; i < DH; i += 4)
			if (i < dh) {
				const float4 u = *(const float4*)(krow + i);
				s = __builtin_amdgcn_mfma_f32_32x32x2f32(u.x, qreg[i], s, 0, 0, 0);
				s = __builtin_amdgcn_mfma_f32_32x32x2f32(u.y, qreg[i + 1], s, 0, 0, 0);
				s = __builtin_amdgcn_mfma_f32_32x32x2f32(u.z, qreg[i + 2], s, 0, 0, 0);
				s = __builtin_amdgcn_mfma_f32_32x32x2f32(u.w, qreg[i + 3], s, 0, 0, 0);
			}
		const float* const vrow = Vs + li * VP + lh * gh;
#pragma unroll
		for (int i = 0; i < GH; i += 4)
			if (i < gh) {
				const float4 u = *(const float4*)(vrow + i);
				dp = __builtin_amdgcn_mfma_f32_32x32x2f32(u.x, greg[i], dp, 0, 0, 0);
				dp = __builtin_amdgcn_mfma_f32_32x32x2f32(u.y, greg[i + 1], dp, 0, 0, 0);
				dp = __builtin_amdgcn_mfma_f32_32x32x2f32(u.z, greg[i + 2], dp, 0, 0, 0);
				dp = __builtin_amdgcn_mfma_f32_32x32x2f32(u.w, greg[i + 3], dp, 0, 0, 0);
			}
		// this lane's row: register r <-> key y0 + (r & 3) + 8 (r >> 2) + 4 lh;  ds = p (dp - delta)
#pragma unroll
		for (int r = 0; r < 16; r++) {
			const int y = y0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
			float ds = 0.f;
			if (y < vis) {
				const float sc = g.scale * s[r] + (mrow ? mrow[y] : 0.f);
				ds = expf(sc - my_lse) * (dp[r] - my_delta);
			}
			s[r] = ds;
		}
#pragma unroll
		for (int j = 0; j < 16; j++) {
			const float* const kcol = Ks + ((j & 3) + 8 * (j >> 2) + 4 * lh) * KP + li;
#pragma unroll
			for (int td = 0; td < TD; td++)
				if (td * 32 < g.D) acc[td] = __builtin_amdgcn_mfma_f32_32x32x2f32(kcol[32 * td], s[j], acc[td], 0, 0, 0);
		}
	}
	if (x < g.R) {
		float* const orow = dq + b * dq_sb + (long)x * dq_sr + h * dq_sh;
#pragma unroll
		for (int td = 0; td < TD; td++)
			if (td * 32 < g.D) {
#pragma unroll
				for (int r4 = 0; r4 < 4; r4++) { // registers 4 r4 .. 4 r4 + 3 are d = 32 td + 8 r4 + 4 lh + 0..3
					const int d = 32 * td + 8 * r4 + 4 * lh;
					orow[d] = g.scale * acc[td][4 * r4]; orow[d + 1] = g.scale * acc[td][4 * r4 + 1]; orow[d + 2] = g.scale * acc[td][4 * r4 + 2]; orow[d + 3] = g.scale * acc[td][4 * r4 + 3];
				}
			}
	}
}

// dk, dv: a wave owns 32 KEYS (their k and v rows in registers), a workgroup 128; the query blocks of 32 of every query head that shares the key head stream
// through LDS (q, g, lse, delta).  S = Q K^T and dP = G V^T with A = the Q / G tile (lane: query l & 31, its half's values), B = the key's registers: lane
// (key, half) holds p and ds of the queries qx(r) of ITS key, and they are the B operands of  dV^T [Dv x 32 keys] += G^T P  and  dK^T [D x 32 keys] += Q^T dS
// (A = G / Q [query qx(j)][column l & 31] out of the same tiles).  One fixed summation order, no atomics.  Conditions: D % 32 == 0, Dv % 32 == 0, both <= 128;
// up to 64 columns each both gradients come out of one pass (four accumulator tiles + both key rows + the two score tiles in 160 registers), beyond that the
// kernel runs twice -- once for dv (no dP product), once for dk -- with half the accumulators each.
template <int TD, int TV, bool DO_K = true, bool DO_V = true> // D / 32, Dv / 32; both gradients in one pass up to 64 columns each, one per pass beyond (the caller launches twice)
__global__ void __launch_bounds__(256) sdpa_dkv_mfma_kernel(const sdpa_geom_t g, const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, const float* __restrict__ mask, const float* __restrict__ gr, const long g_sb, const long g_sr, const long g_sh, const float* __restrict__ lse, const float* __restrict__ delta, float* __restrict__ dk, const long dk_sb, const long dk_sc, const long dk_sh, float* __restrict__ dv, const long dv_sb, const long dv_sc, const long dv_sh)
{
	constexpr int DH = 16 * TD, GH = 16 * TV, QP = 2 * DH + 4, GP = 2 * GH + 4;
	__shared__ __attribute__((aligned(16))) float Qs[32 * QP];
	__shared__ __attribute__((aligned(16))) float Gs[32 * GP];
	__shared__ float Ls[32], Ds[32];
	typedef float floatx16 __attribute__((ext_vector_type(16)));
	const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
	const int li = lane & 31, lh = lane >> 5;
	const int hk = blockIdx.y, b = blockIdx.z;
	const int y0wg = blockIdx.x * 128, y = y0wg + wave * 32 + li;
	float kreg[DH], vreg[GH];
#pragma unroll
	for (int i = 0; i < DH; i += 4) {
		const float4 u = y < g.C ? *(const float4*)(k + b * g.k_sb + (long)y * g.k_sc + hk * g.k_sh + lh * DH + i) : make_float4(0.f, 0.f, 0.f, 0.f);
		kreg[i] = u.x; kreg[i + 1] = u.y; kreg[i + 2] = u.z; kreg[i + 3] = u.w;
	}
	if constexpr (DO_K) {
#pragma unroll
		for (int i = 0; i < GH; i += 4) {
			const float4 u = y < g.C ? *(const float4*)(v + b * g.v_sb + (long)y * g.v_sc + hk * g.v_sh + lh * GH + i) : make_float4(0.f, 0.f, 0.f, 0.f);
			vreg[i] = u.x; vreg[i + 1] = u.y; vreg[i + 2] = u.z; vreg[i + 3] = u.w;
		}
	}
	floatx16 ak[TD], av[TV];
#pragma unroll
	for (int i = 0; i < TD; i++)
#pragma unroll
		for (int r = 0; r < 16; r++) ak[i][r] = 0.f;
#pragma unroll
	for (int i = 0; i < TV; i++)
#pragma unroll
		for (int r = 0; r < 16; r++) av[i][r] = 0.f;
	// first query row that can see the workgroup's first key (causal): x > y0wg + R - C - 1
	int xs = 0;
	if (g.causal) { xs = y0wg + g.R - g.C; if (xs < 0) xs = 0; xs = xs / 32 * 32; }
	for (int h = hk * g.ratio; h < (hk + 1) * g.ratio; h++)
		for (int x0 = xs; x0 < g.R; x0 += 32) {
			__syncthreads();
			for (int c = t; c < 32 * (g.D >> 2); c += 256) {
				const int j = c / (g.D >> 2), d = (c - j * (g.D >> 2)) << 2;
				*(float4*)(Qs + j * QP + d) = x0 + j < g.R ? *(const float4*)(q + b * g.q_sb + (long)(x0 + j) * g.q_sr + h * g.q_sh + d) : make_float4(0.f, 0.f, 0.f, 0.f);
			}
			for (int c = t; c < 32 * (g.Dv >> 2); c += 256) {
				const int j = c / (g.Dv >> 2), d = (c - j * (g.Dv >> 2)) << 2;
				*(float4*)(Gs + j * GP + d) = x0 + j < g.R ? *(const float4*)(gr + b * g_sb + (long)(x0 + j) * g_sr + h * g_sh + d) : make_float4(0.f, 0.f, 0.f, 0.f);
			}
			if (t < 32) { const int x = x0 + t; Ls[t] = x < g.R ? lse[((long)b * g.Hq + h) * g.R + x] : 0.f; Ds[t] = x < g.R ? delta[((long)b * g.Hq + h) * g.R + x] : 0.f; }
			__syncthreads();
			floatx16 s, dp;
#pragma unroll
			for (int r = 0; r < 16; r++) { s[r] = 0.f; dp[r] = 0.f; }
			const float* const qrow = Qs + li * QP + lh * DH;
#pragma unroll
			for (int i = 0; i < DH; i += 4) {
				const float4 u = *(const float4*)(qrow + i);
				s = __builtin_amdgcn_mfma_f32_32x32x2f32(u.x, kreg[i], s, 0, 0, 0);
				s = __builtin_amdgcn_mfma_f32_32x32x2f32(u.y, kreg[i + 1], s, 0, 0, 0);
				s = __builtin_amdgcn_mfma_f32_32x32x2f32(u.z, kreg[i + 2], s, 0, 0, 0);
				s = __builtin_amdgcn_mfma_f32_32x32x2f32(u.w, kreg[i + 3], s, 0, 0, 0);
			}
			if constexpr (DO_K) {
				const float* const grow = Gs + li * GP + lh * GH;
#pragma unroll
				for (int i = 0; i < GH; i += 4) {
					const float4 u = *(const float4*)(grow + i);
					dp = __builtin_amdgcn_mfma_f32_32x32x2f32(u.x, vreg[i], dp, 0, 0, 0);
					dp = __builtin_amdgcn_mfma_f32_32x32x2f32(u.y, vreg[i + 1], dp, 0, 0, 0);
					dp = __builtin_amdgcn_mfma_f32_32x32x2f32(u.z, vreg[i + 2], dp, 0, 0, 0);
					dp = __builtin_amdgcn_mfma_f32_32x32x2f32(u.w, vreg[i + 3], dp, 0, 0, 0);
				}
			}
			// this lane's key: register r <-> query x0 + qx, qx = (r & 3) + 8 (r >> 2) + 4 lh
#pragma unroll
			for (int r = 0; r < 16; r++) {
				const int qx = (r & 3) + 8 * (r >> 2) + 4 * lh, x = x0 + qx;
				float p = 0.f, ds = 0.f;
				if (x < g.R && y < g.C && y < visible_keys(g, x)) {
					const float sc = g.scale * s[r] + (mask ? mask[b * g.m_sb + h * g.m_sh + (long)x * g.m_sr + y] : 0.f);
					p = expf(sc - Ls[qx]);
					ds = p * (dp[r] - Ds[qx]);
				}
				s[r] = p; dp[r] = ds;
			}
#pragma unroll
			for (int j = 0; j < 16; j++) {
				const int qx = (j & 3) + 8 * (j >> 2) + 4 * lh;
				const float* const gcol = Gs + qx * GP + li;
				const float* const qcol = Qs + qx * QP + li;
				if constexpr (DO_V) {
#pragma unroll
					for (int i = 0; i < TV; i++) av[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(gcol[32 * i], s[j], av[i], 0, 0, 0);
				}
				if constexpr (DO_K) {
#pragma unroll
					for (int i = 0; i < TD; i++) ak[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(qcol[32 * i], dp[j], ak[i], 0, 0, 0);
				}
			}
		}
	if (y < g.C) {
#pragma unroll
		for (int r4 = 0; r4 < 4; r4++) { // registers 4 r4 .. 4 r4 + 3 of tile i are column 32 i + 8 r4 + 4 lh + 0..3 of this lane's key
			if (DO_K && dk) {
				float* const o = dk + b * dk_sb + (long)y * dk_sc + hk * dk_sh;
#pragma unroll
				for (int i = 0; i < TD; i++)
#pragma unroll
					for (int e = 0; e < 4; e++) o[32 * i + 8 * r4 + 4 * lh + e] = g.scale * ak[i][4 * r4 + e];
			}
			if (DO_V && dv) {
				float* const o = dv + b * dv_sb + (long)y * dv_sc + hk * dv_sh;
#pragma unroll
				for (int i = 0; i < TV; i++)
#pragma unroll
					for (int e = 0; e < 4; e++) o[32 * i + 8 * r4 + 4 * lh + e] = av[i][4 * r4 + e];
			}
		}
	}
}

// ---- backward in half precision on the f16 matrix cores (round 4) --------------------------------------------------------------------
// The fp32 matrix-core kernels' decomposition on v_mfma_f32_32x32x16_f16.  Every product whose reduction runs over the streamed tile's ROWS wants that tile
// transposed in LDS (eight consecutive rows of one column as two 8-byte reads), so the streamed K (dq) / Q and G (dk, dv) tiles are staged both ways; p and ds are
// rounded to half in the registers they were computed in and are the B operands of the second products (registers 8 t .. 8 t + 7 <-> the eight rows of sub-step t).
// delta = sum_d g o from the half-precision forward pass's scratch output.
__global__ void __launch_bounds__(256) sdpa_delta_h_kernel(const sdpa_geom_t g, const half_t* __restrict__ gr, const long g_sb, const long g_sr, const long g_sh, const half_t* __restrict__ o, float* __restrict__ delta)
{
	const long n = (long)g.B * g.Hq * g.R;
	for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
		const int x = (int)(i % g.R), h = (int)((i / g.R) % g.Hq), b = (int)(i / ((long)g.R * g.Hq));
		const half_t* const gp = gr + b * g_sb + (long)x * g_sr + h * g_sh;
		const half_t* const op = o + (((long)b * g.R + x) * g.Hq + h) * g.Dv;
		float s = 0.f;
		for (int d = 0; d < g.Dv; d++) s += (float)gp[d] * (float)op[d];
		delta[i] = s;
	}
}
// eight rows of column `col` of a transposed tile ([column][40 halves]): rows 16 t + 4 lh + 0..3 and + 8 -- the rows registers 8 t .. 8 t + 7 of a 32 x 32 result stand for
__device__ __forceinline__ halfx8 sdpa_tfrag(const half_t* const tt, const int col, const int t16, const int lh)
{
	const half_t* const p = tt + col * 40 + 16 * t16 + 4 * lh;
	const halfx4 lo = *(const halfx4*)p, hi = *(const halfx4*)(p + 8);
	return halfx8{ lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3] };
}
__device__ __forceinline__ halfx8 sdpa_half8(const float (&v)[16], const int t16)
{
	return halfx8{ (half_t)v[8 * t16], (half_t)v[8 * t16 + 1], (half_t)v[8 * t16 + 2], (half_t)v[8 * t16 + 3], (half_t)v[8 * t16 + 4], (half_t)v[8 * t16 + 5], (half_t)v[8 * t16 + 6], (half_t)v[8 * t16 + 7] };
}
// 32 rows x COLS halves of a [rows][stride] tensor into a row-major tile (pitch COLS + 8) and, when tt != 0, its transpose ([column][40]); rows >= limit read as zeros
template <int COLS>
__device__ __forceinline__ void sdpa_stage_h(const half_t* const src, const long row_stride, const int first, const int limit, half_t* const rm, half_t* const tt, const int t)
{
	for (int c = t; c < 32 * (COLS / 8); c += 256) {
		const int j = c & 31, d = (c >> 5) << 3;
		const halfx8 u = first + j < limit ? *(const halfx8*)(src + (long)(first + j) * row_stride + d) : halfx8{ 0, 0, 0, 0, 0, 0, 0, 0 };
		*(halfx8*)(rm + j * (COLS + 8) + d) = u;
		if (tt) {
#pragma unroll
			for (int e = 0; e < 8; e++) tt[(d + e) * 40 + j] = u[e];
		}
	}
}
template <int DS, int GS> // D / 16 (even: D % 32 == 0), Dv / 16
__global__ void __launch_bounds__(256) sdpa_dq_f16_kernel(const sdpa_geom_t g, const half_t* __restrict__ q, const half_t* __restrict__ k, const half_t* __restrict__ v, const half_t* __restrict__ mask, const half_t* __restrict__ gr, const long g_sb, const long g_sr, const long g_sh, const float* __restrict__ lse, const float* __restrict__ delta, half_t* __restrict__ dq, const long dq_sb, const long dq_sr, const long dq_sh)
{
	constexpr int D = 16 * DS, DV = 16 * GS, TD = DS / 2;
	__shared__ __attribute__((aligned(16))) half_t Ks[32 * (D + 8)];
	__shared__ __attribute__((aligned(16))) half_t Kt[D * 40];
	__shared__ __attribute__((aligned(16))) half_t Vs[32 * (DV + 8)];
	typedef float floatx16 __attribute__((ext_vector_type(16)));
	const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
	const int li = lane & 31, lh = lane >> 5;
	const int h = blockIdx.y, b = blockIdx.z, hk = h / g.ratio;
	const int x = blockIdx.x * 128 + wave * 32 + li;
	halfx8 qf[DS], gf[GS];
#pragma unroll
	for (int i = 0; i < DS; i++) qf[i] = x < g.R ? *(const halfx8*)(q + b * g.q_sb + (long)x * g.q_sr + h * g.q_sh + 16 * i + 8 * lh) : halfx8{ 0, 0, 0, 0, 0, 0, 0, 0 };
#pragma unroll
	for (int i = 0; i < GS; i++) gf[i] = x < g.R ? *(const halfx8*)(gr + b * g_sb + (long)x * g_sr + h * g_sh + 16 * i + 8 * lh) : halfx8{ 0, 0, 0, 0, 0, 0, 0, 0 };
	floatx16 acc[TD];
#pragma unroll
	for (int td = 0; td < TD; td++)
#pragma unroll
		for (int r = 0; r < 16; r++) acc[td][r] = 0.f;
	const int vis = x < g.R ? visible_keys(g, x) : 0;
	const float my_lse = x < g.R ? lse[((long)b * g.Hq + h) * g.R + x] : 0.f, my_delta = x < g.R ? delta[((long)b * g.Hq + h) * g.R + x] : 0.f;
	int vis_max = 0;
	{
		const int x_last = blockIdx.x * 128 + 127 < g.R ? blockIdx.x * 128 + 127 : g.R - 1;
		vis_max = visible_keys(g, x_last);
	}
	const half_t* const mrow = mask ? mask + b * g.m_sb + h * g.m_sh + (long)(x < g.R ? x : 0) * g.m_sr : 0;
	for (int y0 = 0; y0 < vis_max; y0 += 32) {
		__syncthreads();
		sdpa_stage_h<D>(k + b * g.k_sb + hk * g.k_sh, g.k_sc, y0, g.C, Ks, Kt, t);
		sdpa_stage_h<DV>(v + b * g.v_sb + hk * g.v_sh, g.v_sc, y0, g.C, Vs, (half_t*)0, t);
		__syncthreads();
		floatx16 s, dp;
#pragma unroll
		for (int r = 0; r < 16; r++) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
		for (int i = 0; i < DS; i++) s = nnc_mfma_f16(*(const halfx8*)(Ks + li * (D + 8) + 16 * i + 8 * lh), qf[i], s);
#pragma unroll
		for (int i = 0; i < GS; i++) dp = nnc_mfma_f16(*(const halfx8*)(Vs + li * (DV + 8) + 16 * i + 8 * lh), gf[i], dp);
		float ds[16];
#pragma unroll
		for (int r = 0; r < 16; r++) {
			const int y = y0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
			ds[r] = y < vis ? expf(g.scale * s[r] + (mrow ? (float)mrow[y] : 0.f) - my_lse) * (dp[r] - my_delta) : 0.f;
		}
#pragma unroll
		for (int tt = 0; tt < 2; tt++) {
			const halfx8 dsf = sdpa_half8(ds, tt);
#pragma unroll
			for (int td = 0; td < TD; td++) acc[td] = nnc_mfma_f16(sdpa_tfrag(Kt, 32 * td + li, tt, lh), dsf, acc[td]);
		}
	}
	if (x < g.R) {
		half_t* const orow = dq + b * dq_sb + (long)x * dq_sr + h * dq_sh;
#pragma unroll
		for (int td = 0; td < TD; td++)
#pragma unroll
			for (int r4 = 0; r4 < 4; r4++)
				*(halfx4*)(orow + 32 * td + 8 * r4 + 4 * lh) = halfx4{ (half_t)(g.scale * acc[td][4 * r4]), (half_t)(g.scale * acc[td][4 * r4 + 1]), (half_t)(g.scale * acc[td][4 * r4 + 2]), (half_t)(g.scale * acc[td][4 * r4 + 3]) };
	}
}
template <int DS, int GS> // D / 16, Dv / 16 (both even)
__global__ void __launch_bounds__(256) sdpa_dkv_f16_kernel(const sdpa_geom_t g, const half_t* __restrict__ q, const half_t* __restrict__ k, const half_t* __restrict__ v, const half_t* __restrict__ mask, const half_t* __restrict__ gr, const long g_sb, const long g_sr, const long g_sh, const float* __restrict__ lse, const float* __restrict__ delta, half_t* __restrict__ dk, const long dk_sb, const long dk_sc, const long dk_sh, half_t* __restrict__ dv, const long dv_sb, const long dv_sc, const long dv_sh)
{
	constexpr int D = 16 * DS, DV = 16 * GS, TD = DS / 2, TV = GS / 2;
	__shared__ __attribute__((aligned(16))) half_t Qs[32 * (D + 8)];
	__shared__ __attribute__((aligned(16))) half_t Qt[D * 40];
	__shared__ __attribute__((aligned(16))) half_t Gs[32 * (DV + 8)];
	__shared__ __attribute__((aligned(16))) half_t Gt[DV * 40];
	__shared__ float Ls[32], Ds[32];
	typedef float floatx16 __attribute__((ext_vector_type(16)));
	const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
	const int li = lane & 31, lh = lane >> 5;
	const int hk = blockIdx.y, b = blockIdx.z;
	const int y0wg = blockIdx.x * 128, y = y0wg + wave * 32 + li;
	halfx8 kf[DS], vf[GS];
#pragma unroll
	for (int i = 0; i < DS; i++) kf[i] = y < g.C ? *(const halfx8*)(k + b * g.k_sb + (long)y * g.k_sc + hk * g.k_sh + 16 * i + 8 * lh) : halfx8{ 0, 0, 0, 0, 0, 0, 0, 0 };
#pragma unroll
	for (int i = 0; i < GS; i++) vf[i] = y < g.C ? *(const halfx8*)(v + b * g.v_sb + (long)y * g.v_sc + hk * g.v_sh + 16 * i + 8 * lh) : halfx8{ 0, 0, 0, 0, 0, 0, 0, 0 };
	floatx16 ak[TD], av[TV];
#pragma unroll
	for (int i = 0; i < TD; i++)
#pragma unroll
		for (int r = 0; r < 16; r++) ak[i][r] = 0.f;
#pragma unroll
	for (int i = 0; i < TV; i++)
#pragma unroll
		for (int r = 0; r < 16; r++) av[i][r] = 0.f;
	int xs = 0;
	if (g.causal) { xs = y0wg + g.R - g.C; if (xs < 0) xs = 0; xs = xs / 32 * 32; }
	for (int h = hk * g.ratio; h < (hk + 1) * g.ratio; h++)
		for (int x0 = xs; x0 < g.R; x0 += 32) {
			__syncthreads();
			sdpa_stage_h<D>(q + b * g.q_sb + h * g.q_sh, g.q_sr, x0, g.R, Qs, Qt, t);
			sdpa_stage_h<DV>(gr + b * g_sb + h * g_sh, g_sr, x0, g.R, Gs, Gt, t);
			if (t < 32) { const int x = x0 + t; Ls[t] = x < g.R ? lse[((long)b * g.Hq + h) * g.R + x] : 0.f; Ds[t] = x < g.R ? delta[((long)b * g.Hq + h) * g.R + x] : 0.f; }
			__syncthreads();
			floatx16 s, dp;
#pragma unroll
			for (int r = 0; r < 16; r++) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
			for (int i = 0; i < DS; i++) s = nnc_mfma_f16(*(const halfx8*)(Qs + li * (D + 8) + 16 * i + 8 * lh), kf[i], s);
#pragma unroll
			for (int i = 0; i < GS; i++) dp = nnc_mfma_f16(*(const halfx8*)(Gs + li * (DV + 8) + 16 * i + 8 * lh), vf[i], dp);
			float pv[16], ds[16];
#pragma unroll
			for (int r = 0; r < 16; r++) {
				const int qx = (r & 3) + 8 * (r >> 2) + 4 * lh, x = x0 + qx;
				float p = 0.f;
				if (x < g.R && y < g.C && y < visible_keys(g, x)) p = expf(g.scale * s[r] + (mask ? (float)mask[b * g.m_sb + h * g.m_sh + (long)x * g.m_sr + y] : 0.f) - Ls[qx]);
				pv[r] = p;
				ds[r] = p * (dp[r] - Ds[qx]);
			}
#pragma unroll
			for (int tt = 0; tt < 2; tt++) {
				const halfx8 pf = sdpa_half8(pv, tt), dsf = sdpa_half8(ds, tt);
#pragma unroll
				for (int i = 0; i < TV; i++) av[i] = nnc_mfma_f16(sdpa_tfrag(Gt, 32 * i + li, tt, lh), pf, av[i]);
#pragma unroll
				for (int i = 0; i < TD; i++) ak[i] = nnc_mfma_f16(sdpa_tfrag(Qt, 32 * i + li, tt, lh), dsf, ak[i]);
			}
		}
	if (y < g.C) {
#pragma unroll
		for (int r4 = 0; r4 < 4; r4++) {
			if (dk) {
				half_t* const o = dk + b * dk_sb + (long)y * dk_sc + hk * dk_sh;
#pragma unroll
				for (int i = 0; i < TD; i++) *(halfx4*)(o + 32 * i + 8 * r4 + 4 * lh) = halfx4{ (half_t)(g.scale * ak[i][4 * r4]), (half_t)(g.scale * ak[i][4 * r4 + 1]), (half_t)(g.scale * ak[i][4 * r4 + 2]), (half_t)(g.scale * ak[i][4 * r4 + 3]) };
			}
			if (dv) {
				half_t* const o = dv + b * dv_sb + (long)y * dv_sc + hk * dv_sh;
#pragma unroll
				for (int i = 0; i < TV; i++) *(halfx4*)(o + 32 * i + 8 * r4 + 4 * lh) = halfx4{ (half_t)av[i][4 * r4], (half_t)av[i][4 * r4 + 1], (half_t)av[i][4 * r4 + 2], (half_t)av[i][4 * r4 + 3] };
			}
		}
	}
}

// ---- host -----------------------------------------------------------------------------------------------------------------------
struct bhd_t { int b, n, h, d; long sb, sn, sh; };
static bool bhd(const ccv_nnc_tensor_t* t, bhd_t* o, const int datatype = CCV_32F)
{ // [B, N, H, D] or [B, N, D]; D contiguous
	const int nd = tensor_nd(t->info.dim);
	if ((nd != 3 && nd != 4) || CCV_GET_DATA_TYPE(t->info.datatype) != datatype) return false;
	int st[CCV_NNC_MAX_DIM_ALLOC];
	tensor_strides(t, st);
	if (st[nd - 1] != 1) return false;
	o->b = t->info.dim[0]; o->n = t->info.dim[1]; o->sb = st[0]; o->sn = st[1];
	if (nd == 4) { o->h = t->info.dim[2]; o->d = t->info.dim[3]; o->sh = st[2]; }
	else { o->h = 1; o->d = t->info.dim[2]; o->sh = 0; }
	return true;
}
static bool sdpa_mask_geometry(const ccv_nnc_tensor_t* mask, sdpa_geom_t* g, int datatype);
static bool sdpa_geometry(const ccv_nnc_cmd_t& cmd, const ccv_nnc_tensor_t* q, const ccv_nnc_tensor_t* k, const ccv_nnc_tensor_t* v, const ccv_nnc_tensor_t* mask, sdpa_geom_t* g, bhd_t* qi, bhd_t* ki, bhd_t* vi)
{
	if (!bhd(q, qi) || !bhd(k, ki) || !bhd(v, vi)) return false;
	if (tensor_nd(q->info.dim) != tensor_nd(k->info.dim) || tensor_nd(k->info.dim) != tensor_nd(v->info.dim)) return false;
	if (qi->b != ki->b || ki->b != vi->b || qi->d != ki->d || ki->n != vi->n || ki->h != vi->h || qi->h < ki->h || qi->h % ki->h) return false;
	g->B = qi->b; g->R = qi->n; g->C = ki->n; g->Hq = qi->h; g->Hk = ki->h; g->D = qi->d; g->Dv = vi->d; g->ratio = qi->h / ki->h;
	g->q_sb = qi->sb; g->q_sr = qi->sn; g->q_sh = qi->sh; g->k_sb = ki->sb; g->k_sc = ki->sn; g->k_sh = ki->sh; g->v_sb = vi->sb; g->v_sc = vi->sn; g->v_sh = vi->sh;
	g->m_sb = g->m_sh = g->m_sr = 0;
	if (mask && !sdpa_mask_geometry(mask, g, CCV_32F)) return false;
	g->scale = cmd.info.scaled_dot_product_attention.scale;
	g->causal = cmd.info.scaled_dot_product_attention.is_causal;
	return g->D >= 1 && g->Dv >= 1 && g->D <= 256 && g->Dv <= 256;
}
static bool sdpa_mask_geometry(const ccv_nnc_tensor_t* const mask, sdpa_geom_t* const g, const int datatype)
{
	{ // [B or 1][Hq or 1][R][C], or 3-d [B or 1][R][C]
		const int nd = tensor_nd(mask->info.dim);
		if ((nd != 3 && nd != 4) || CCV_GET_DATA_TYPE(mask->info.datatype) != datatype) return false;
		int st[CCV_NNC_MAX_DIM_ALLOC];
		tensor_strides(mask, st);
		if (st[nd - 1] != 1 || mask->info.dim[nd - 1] != g->C || mask->info.dim[nd - 2] != g->R) return false;
		g->m_sr = st[nd - 2];
		const int mb = mask->info.dim[0], mh = nd == 4 ? mask->info.dim[1] : 1;
		if ((mb != 1 && mb != g->B) || (mh != 1 && mh != g->Hq)) return false;
		g->m_sb = mb == 1 ? 0 : st[0];
		g->m_sh = (nd == 4 && mh != 1) ? st[1] : 0;
	}
	return true;
}
static int sdpa_forward_launch(const sdpa_geom_t& g, const float* q, const float* k, const float* v, const float* mask, float* o, float* lse, hipStream_t stream)
{
	if (!g.R || !g.Hq || !g.B) return CCV_NNC_EXEC_SUCCESS;
	// the matrix-core kernel: whole 16-byte chunks of q / k / v rows, the output tile a whole number of MFMA tiles
	const bool rows16 = !(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) && !((g.q_sb | g.q_sr | g.q_sh | g.k_sb | g.k_sc | g.k_sh | g.v_sb | g.v_sc | g.v_sh) & 3);
	if (tune(TUNE_SDPA_MFMA) && rows16 && g.D % 8 == 0 && g.Dv % 32 == 0 && g.D <= 128 && g.Dv <= 128) {
		const dim3 grid((g.R + 127) / 128, g.Hq, g.B);
		const int tv = g.Dv / 32;
		// (both products of every (row, key) pair; causal masks skip key blocks, so this is an upper bound there)
		ProfScope prof("sdpa_fwd|nnc::sdpa_forw_mfma_kernel", 2.0 * g.B * g.Hq * (double)g.R * g.C * (g.D + g.Dv), 0, g.R, g.C, g.D, g.B * g.Hq, 1, stream);
#define SDPA_MFMA(DH, TV) hipLaunchKernelGGL(HIP_KERNEL_NAME(sdpa_forw_mfma_kernel<DH, TV>), grid, dim3(256), 0, stream, g, q, k, v, mask, o, lse)
		if (g.D <= 64) { if (tv == 1) SDPA_MFMA(32, 1); else if (tv == 2) SDPA_MFMA(32, 2); else if (tv == 3) SDPA_MFMA(32, 3); else SDPA_MFMA(32, 4); }
		else { if (tv == 1) SDPA_MFMA(64, 1); else if (tv == 2) SDPA_MFMA(64, 2); else if (tv == 3) SDPA_MFMA(64, 3); else SDPA_MFMA(64, 4); }
#undef SDPA_MFMA
		HIP_ENFORCE(hipGetLastError());
		return CCV_NNC_EXEC_SUCCESS;
	}
	const dim3 grid((g.R + BR - 1) / BR, g.Hq, g.B);
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

static void sdpa_forw_f16_launch(const sdpa_geom_t& g, const half_t* const qp, const half_t* const kp, const half_t* const vp, const half_t* const mp, half_t* const op, float* const lse, hipStream_t stream)
{
	const dim3 grid((g.R + 127) / 128, g.Hq, g.B);
	note_kernel("sdpa_fwd_f16");
	ProfScope prof("sdpa_fwd_h|nnc::sdpa_forw_f16_kernel", 2.0 * g.B * g.Hq * (double)g.R * g.C * (g.D + g.Dv), 0, g.R, g.C, g.D, g.B * g.Hq, 1, stream);
#define SDPA_F16_TV(DS) do { switch (g.Dv / 32) { \
		case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(sdpa_forw_f16_kernel<DS, 1>), grid, dim3(256), 0, stream, g, qp, kp, vp, mp, op, lse); break; \
		case 2: hipLaunchKernelGGL(HIP_KERNEL_NAME(sdpa_forw_f16_kernel<DS, 2>), grid, dim3(256), 0, stream, g, qp, kp, vp, mp, op, lse); break; \
		case 3: hipLaunchKernelGGL(HIP_KERNEL_NAME(sdpa_forw_f16_kernel<DS, 3>), grid, dim3(256), 0, stream, g, qp, kp, vp, mp, op, lse); break; \
		default: hipLaunchKernelGGL(HIP_KERNEL_NAME(sdpa_forw_f16_kernel<DS, 4>), grid, dim3(256), 0, stream, g, qp, kp, vp, mp, op, lse); break; } } while (0)
	switch (g.D / 16) {
		case 1: SDPA_F16_TV(1); break; case 2: SDPA_F16_TV(2); break; case 3: SDPA_F16_TV(3); break; case 4: SDPA_F16_TV(4); break;
		case 5: SDPA_F16_TV(5); break; case 6: SDPA_F16_TV(6); break; case 7: SDPA_F16_TV(7); break; default: SDPA_F16_TV(8); break;
	}
#undef SDPA_F16_TV
	HIP_ENFORCE(hipGetLastError());
}
// CCV_16F tensors: the f16 kernel where its conditions hold (CCV_NNC_EXEC_NO_KERNEL = not this path: the caller goes through fp32 images)
static int sdpa_forw_half(EXEC_ARGS)
{
	if (input_size < 3 || output_size < 1 || !inputs[0] || !inputs[1] || !inputs[2] || !outputs[0]) return CCV_NNC_EXEC_NO_KERNEL;
	for (int i = 4; i < input_size; i++) if (inputs[i]) return CCV_NNC_EXEC_NO_KERNEL; // the head projection: the fp32 route
	for (int i = 2; i < output_size; i++) if (outputs[i]) return CCV_NNC_EXEC_NO_KERNEL;
	const ccv_nnc_tensor_t* const mask = input_size > 3 ? inputs[3] : 0;
	if (!tune(TUNE_SDPA_MFMA)) return CCV_NNC_EXEC_NO_KERNEL;
	const ccv_nnc_tensor_t* const q = inputs[0]; const ccv_nnc_tensor_t* const k = inputs[1]; const ccv_nnc_tensor_t* const v = inputs[2];
	ccv_nnc_tensor_t* const c = outputs[0];
	ccv_nnc_tensor_t* const lse_t = output_size > 1 ? outputs[1] : 0;
	bhd_t qi, ki, vi, ci;
	if (!bhd(q, &qi, CCV_16F) || !bhd(k, &ki, CCV_16F) || !bhd(v, &vi, CCV_16F) || !bhd(c, &ci, CCV_16F)) return CCV_NNC_EXEC_NO_KERNEL;
	if (tensor_nd(q->info.dim) != tensor_nd(k->info.dim) || tensor_nd(k->info.dim) != tensor_nd(v->info.dim)) return CCV_NNC_EXEC_NO_KERNEL;
	if (qi.b != ki.b || ki.b != vi.b || qi.d != ki.d || ki.n != vi.n || ki.h != vi.h || qi.h < ki.h || qi.h % ki.h) return CCV_NNC_EXEC_NO_KERNEL;
	sdpa_geom_t g;
	memset(&g, 0, sizeof(g));
	g.B = qi.b; g.R = qi.n; g.C = ki.n; g.Hq = qi.h; g.Hk = ki.h; g.D = qi.d; g.Dv = vi.d; g.ratio = qi.h / ki.h;
	g.q_sb = qi.sb; g.q_sr = qi.sn; g.q_sh = qi.sh; g.k_sb = ki.sb; g.k_sc = ki.sn; g.k_sh = ki.sh; g.v_sb = vi.sb; g.v_sc = vi.sn; g.v_sh = vi.sh;
	g.scale = cmd.info.scaled_dot_product_attention.scale;
	g.causal = cmd.info.scaled_dot_product_attention.is_causal;
	if (ci.b != g.B || ci.n != g.R || ci.h != g.Hq || ci.d != g.Dv) return CCV_NNC_EXEC_INVALID;
	g.o_sb = ci.sb; g.o_sr = ci.sn; g.o_sh = ci.sh;
	if (mask && !sdpa_mask_geometry(mask, &g, CCV_16F)) return CCV_NNC_EXEC_NO_KERNEL;
	if (g.D % 16 || g.Dv % 32 || g.D > 128 || g.Dv > 128 || g.D < 16) return CCV_NNC_EXEC_NO_KERNEL;
	if ((((uintptr_t)q->data.u8 | (uintptr_t)k->data.u8 | (uintptr_t)v->data.u8) & 15) || ((uintptr_t)c->data.u8 & 7)) return CCV_NNC_EXEC_NO_KERNEL;
	if ((g.q_sb | g.q_sr | g.q_sh | g.k_sb | g.k_sc | g.k_sh | g.v_sb | g.v_sc | g.v_sh) & 7) return CCV_NNC_EXEC_NO_KERNEL;
	if ((g.o_sb | g.o_sr | g.o_sh) & 3) return CCV_NNC_EXEC_NO_KERNEL;
	float* lse = 0;
	if (lse_t) {
		if (CCV_GET_DATA_TYPE(lse_t->info.datatype) != CCV_32F || !tensor_contiguous(lse_t) || tensor_count(lse_t->info) != (size_t)g.B * g.Hq * g.R) return CCV_NNC_EXEC_NO_KERNEL;
		lse = lse_t->data.f32;
	}
	MarkerScope marker(cmd.cmd);
	if (!g.R || !g.Hq || !g.B) return CCV_NNC_EXEC_SUCCESS;
	sdpa_forw_f16_launch(g, (const half_t*)q->data.u8, (const half_t*)k->data.u8, (const half_t*)v->data.u8, mask ? (const half_t*)mask->data.u8 : 0, (half_t*)c->data.u8, lse, stream_of(stream_context));
	return CCV_NNC_EXEC_SUCCESS;
}
// the registered forward entry: fp32 tensors -> the fp32 kernels; half q / k / v / o -> the f16 kernel where it applies; anything else with a half tensor -> the fp32
// kernels on fp32 images (half_stage.cpp)
static int _sdpa_forw_any(EXEC_ARGS)
{
	if (!any_half_tensor(inputs, input_size, outputs, output_size)) {
		MarkerScope marker(cmd.cmd);
		return _sdpa_forw(cmd, hint, flags, inputs, input_size, outputs, output_size, stream_context);
	}
	const int r = sdpa_forw_half(cmd, hint, flags, inputs, input_size, outputs, output_size, stream_context); // (opens its own marker range once it knows it runs)
	if (r != CCV_NNC_EXEC_NO_KERNEL) return r;
	return half_staged_exec(_sdpa_forw, cmd, hint, flags, inputs, input_size, outputs, output_size, stream_context);
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
	// the matrix-core kernels: whole 16-byte chunks of every row, whole 32-column output tiles
	const bool rows16 = !(((uintptr_t)qp | (uintptr_t)kp | (uintptr_t)vp | (uintptr_t)gp) & 15) && !((g.q_sb | g.q_sr | g.q_sh | g.k_sb | g.k_sc | g.k_sh | g.v_sb | g.v_sc | g.v_sh | gi.sb | gi.sn | gi.sh) & 3);
	const bool mfma_dq = tune(TUNE_SDPA_MFMA) && rows16 && g.D % 32 == 0 && g.Dv % 8 == 0 && g.D <= 128 && g.Dv <= 128;
	const bool mfma_dkv = tune(TUNE_SDPA_MFMA) && rows16 && g.D % 32 == 0 && g.Dv % 32 == 0 && g.D <= 128 && g.Dv <= 128;
	if (dq && mfma_dq) {
		const dim3 grid((g.R + 127) / 128, g.Hq, g.B);
		const int td = g.D / 32;
		note_kernel("sdpa_dq_mfma");
		ProfScope prof("sdpa_dq|nnc::sdpa_dq_mfma_kernel", 2.0 * g.B * g.Hq * (double)g.R * g.C * (2 * g.D + g.Dv), 0, g.R, g.C, g.D, g.B * g.Hq, 1, stream);
#define SDPA_DQ(DH, GH, TD) hipLaunchKernelGGL(HIP_KERNEL_NAME(sdpa_dq_mfma_kernel<DH, GH, TD>), grid, dim3(256), 0, stream, g, qp, kp, vp, mp, gp, gi.sb, gi.sn, gi.sh, (const float*)lse, (const float*)delta, dq->data.f32, dqi.sb, dqi.sn, dqi.sh)
		if (g.Dv <= 64) { if (td == 1) SDPA_DQ(32, 32, 1); else if (td == 2) SDPA_DQ(32, 32, 2); else if (td == 3) SDPA_DQ(64, 32, 3); else SDPA_DQ(64, 32, 4); }
		else { if (td == 1) SDPA_DQ(32, 64, 1); else if (td == 2) SDPA_DQ(32, 64, 2); else if (td == 3) SDPA_DQ(64, 64, 3); else SDPA_DQ(64, 64, 4); }
#undef SDPA_DQ
		HIP_ENFORCE(hipGetLastError());
	} else if (dq) {
		const dim3 grid((g.R + BR - 1) / BR, g.Hq, g.B);
		if (dm <= 64) hipLaunchKernelGGL(HIP_KERNEL_NAME(sdpa_dq_kernel<64, 64>), grid, dim3(256), 0, stream, g, qp, kp, vp, mp, gp, gi.sb, gi.sn, gi.sh, (const float*)lse, (const float*)delta, dq->data.f32, dqi.sb, dqi.sn, dqi.sh);
		else if (dm <= 128) hipLaunchKernelGGL(HIP_KERNEL_NAME(sdpa_dq_kernel<128, 32>), grid, dim3(256), 0, stream, g, qp, kp, vp, mp, gp, gi.sb, gi.sn, gi.sh, (const float*)lse, (const float*)delta, dq->data.f32, dqi.sb, dqi.sn, dqi.sh);
		else hipLaunchKernelGGL(HIP_KERNEL_NAME(sdpa_dq_kernel<256, 16>), grid, dim3(256), 0, stream, g, qp, kp, vp, mp, gp, gi.sb, gi.sn, gi.sh, (const float*)lse, (const float*)delta, dq->data.f32, dqi.sb, dqi.sn, dqi.sh);
		HIP_ENFORCE(hipGetLastError());
	}
	if ((dk || dv) && mfma_dkv) {
		const dim3 grid((g.C + 127) / 128, g.Hk, g.B);
		float* const dkp = dk ? dk->data.f32 : 0; float* const dvp = dv ? dv->data.f32 : 0;
		const long ksb = dk ? dki.sb : 0, ksn = dk ? dki.sn : 0, ksh = dk ? dki.sh : 0, vsb = dv ? dvi.sb : 0, vsn = dv ? dvi.sn : 0, vsh = dv ? dvi.sh : 0;
		note_kernel("sdpa_dkv_mfma");
		ProfScope prof("sdpa_dkv|nnc::sdpa_dkv_mfma_kernel", 2.0 * g.B * g.Hq * (double)g.R * g.C * (2 * g.D + 2 * g.Dv), 0, g.C, g.R, g.D, g.B * g.Hk, 1, stream);
#define SDPA_DKV(TD, TV) hipLaunchKernelGGL(HIP_KERNEL_NAME(sdpa_dkv_mfma_kernel<TD, TV>), grid, dim3(256), 0, stream, g, qp, kp, vp, mp, gp, gi.sb, gi.sn, gi.sh, (const float*)lse, (const float*)delta, dkp, ksb, ksn, ksh, dvp, vsb, vsn, vsh)
		if (g.D <= 64 && g.Dv <= 64) {
			if (g.D == 32) { if (g.Dv == 32) SDPA_DKV(1, 1); else SDPA_DKV(1, 2); }
			else { if (g.Dv == 32) SDPA_DKV(2, 1); else SDPA_DKV(2, 2); }
		} else {
			// one gradient per pass (registers): dv first, then dk
#define SDPA_DKV2(TD, TV) do { if (dvp) hipLaunchKernelGGL(HIP_KERNEL_NAME(sdpa_dkv_mfma_kernel<TD, TV, false, true>), grid, dim3(256), 0, stream, g, qp, kp, vp, mp, gp, gi.sb, gi.sn, gi.sh, (const float*)lse, (const float*)delta, dkp, ksb, ksn, ksh, dvp, vsb, vsn, vsh); \
			if (dkp) hipLaunchKernelGGL(HIP_KERNEL_NAME(sdpa_dkv_mfma_kernel<TD, TV, true, false>), grid, dim3(256), 0, stream, g, qp, kp, vp, mp, gp, gi.sb, gi.sn, gi.sh, (const float*)lse, (const float*)delta, dkp, ksb, ksn, ksh, dvp, vsb, vsn, vsh); } while (0)
#define SDPA_DKV2_TV(TD) do { switch (g.Dv / 32) { case 1: SDPA_DKV2(TD, 1); break; case 2: SDPA_DKV2(TD, 2); break; case 3: SDPA_DKV2(TD, 3); break; default: SDPA_DKV2(TD, 4); break; } } while (0)
			switch (g.D / 32) { case 1: SDPA_DKV2_TV(1); break; case 2: SDPA_DKV2_TV(2); break; case 3: SDPA_DKV2_TV(3); break; default: SDPA_DKV2_TV(4); break; }
#undef SDPA_DKV2_TV
#undef SDPA_DKV2
		}
#undef SDPA_DKV
		HIP_ENFORCE(hipGetLastError());
	} else if (dk || dv) {
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

// CCV_16F g / q / k / v -> dq / dk / dv: the f16 kernels where their conditions hold (CCV_NNC_EXEC_NO_KERNEL = not this path)
static int sdpa_back_half(EXEC_ARGS)
{
	if (input_size < 6 || output_size < 3 || !inputs[0] || !inputs[3] || !inputs[4] || !inputs[5]) return CCV_NNC_EXEC_NO_KERNEL;
	for (int i = 7; i < input_size && i < 9; i++) if (inputs[i]) return CCV_NNC_EXEC_NO_KERNEL; // the head projection: the fp32 route
	const ccv_nnc_tensor_t* const mask = input_size > 6 ? inputs[6] : 0;
	if (!tune(TUNE_SDPA_MFMA)) return CCV_NNC_EXEC_NO_KERNEL;
	ccv_nnc_tensor_t* const dq = outputs[0]; ccv_nnc_tensor_t* const dk = outputs[1]; ccv_nnc_tensor_t* const dv = outputs[2];
	const ccv_nnc_tensor_t* const q = inputs[3]; const ccv_nnc_tensor_t* const k = inputs[4]; const ccv_nnc_tensor_t* const v = inputs[5];
	bhd_t qi, ki, vi, gi, dqi, dki, dvi;
	if (!bhd(q, &qi, CCV_16F) || !bhd(k, &ki, CCV_16F) || !bhd(v, &vi, CCV_16F) || !bhd(inputs[0], &gi, CCV_16F)) return CCV_NNC_EXEC_NO_KERNEL;
	if (tensor_nd(q->info.dim) != tensor_nd(k->info.dim) || tensor_nd(k->info.dim) != tensor_nd(v->info.dim)) return CCV_NNC_EXEC_NO_KERNEL;
	if (qi.b != ki.b || ki.b != vi.b || qi.d != ki.d || ki.n != vi.n || ki.h != vi.h || qi.h < ki.h || qi.h % ki.h) return CCV_NNC_EXEC_NO_KERNEL;
	sdpa_geom_t g;
	memset(&g, 0, sizeof(g));
	g.B = qi.b; g.R = qi.n; g.C = ki.n; g.Hq = qi.h; g.Hk = ki.h; g.D = qi.d; g.Dv = vi.d; g.ratio = qi.h / ki.h;
	g.q_sb = qi.sb; g.q_sr = qi.sn; g.q_sh = qi.sh; g.k_sb = ki.sb; g.k_sc = ki.sn; g.k_sh = ki.sh; g.v_sb = vi.sb; g.v_sc = vi.sn; g.v_sh = vi.sh;
	g.scale = cmd.info.scaled_dot_product_attention.scale;
	g.causal = cmd.info.scaled_dot_product_attention.is_causal;
	if (gi.b != g.B || gi.n != g.R || gi.h != g.Hq || gi.d != g.Dv) return CCV_NNC_EXEC_INVALID;
	if (dq && (!bhd(dq, &dqi, CCV_16F) || dqi.b != g.B || dqi.n != g.R || dqi.h != g.Hq || dqi.d != g.D)) return CCV_NNC_EXEC_NO_KERNEL;
	if (dk && (!bhd(dk, &dki, CCV_16F) || dki.b != g.B || dki.n != g.C || dki.h != g.Hk || dki.d != g.D)) return CCV_NNC_EXEC_NO_KERNEL;
	if (dv && (!bhd(dv, &dvi, CCV_16F) || dvi.b != g.B || dvi.n != g.C || dvi.h != g.Hk || dvi.d != g.Dv)) return CCV_NNC_EXEC_NO_KERNEL;
	if (g.D % 32 || g.Dv % 32 || g.D > 128 || g.Dv > 128) return CCV_NNC_EXEC_NO_KERNEL;
	if (mask && !sdpa_mask_geometry(mask, &g, CCV_16F)) return CCV_NNC_EXEC_NO_KERNEL;
	const half_t* const mp = mask ? (const half_t*)mask->data.u8 : 0;
	if (((uintptr_t)q->data.u8 | (uintptr_t)k->data.u8 | (uintptr_t)v->data.u8 | (uintptr_t)inputs[0]->data.u8) & 15) return CCV_NNC_EXEC_NO_KERNEL;
	if ((g.q_sb | g.q_sr | g.q_sh | g.k_sb | g.k_sc | g.k_sh | g.v_sb | g.v_sc | g.v_sh | gi.sb | gi.sn | gi.sh) & 7) return CCV_NNC_EXEC_NO_KERNEL;
	if (dq && ((((uintptr_t)dq->data.u8) & 7) || ((dqi.sb | dqi.sn | dqi.sh) & 3))) return CCV_NNC_EXEC_NO_KERNEL;
	if (dk && ((((uintptr_t)dk->data.u8) & 7) || ((dki.sb | dki.sn | dki.sh) & 3))) return CCV_NNC_EXEC_NO_KERNEL;
	if (dv && ((((uintptr_t)dv->data.u8) & 7) || ((dvi.sb | dvi.sn | dvi.sh) & 3))) return CCV_NNC_EXEC_NO_KERNEL;
	MarkerScope marker(cmd.cmd);
	if (!g.B || !g.R || !g.C) return CCV_NNC_EXEC_SUCCESS;
	const size_t rows = (size_t)g.B * g.Hq * g.R;
	const size_t o_bytes = (sizeof(half_t) * rows * g.Dv + 255) & ~(size_t)255, r_bytes = (sizeof(float) * rows + 255) & ~(size_t)255;
	char* const ws = (char*)workspace_of(stream_context, o_bytes + 2 * r_bytes);
	if (!ws) return CCV_NNC_EXEC_OOM;
	half_t* const o = (half_t*)ws; float* const lse = (float*)(ws + o_bytes); float* const delta = (float*)(ws + o_bytes + r_bytes);
	hipStream_t stream = stream_of(stream_context);
	sdpa_geom_t gf = g;
	gf.o_sb = (long)g.R * g.Hq * g.Dv; gf.o_sr = (long)g.Hq * g.Dv; gf.o_sh = g.Dv;
	const half_t* const qp = (const half_t*)q->data.u8; const half_t* const kp = (const half_t*)k->data.u8; const half_t* const vp = (const half_t*)v->data.u8; const half_t* const gp = (const half_t*)inputs[0]->data.u8;
	sdpa_forw_f16_launch(gf, qp, kp, vp, mp, o, lse, stream);
	hipLaunchKernelGGL(sdpa_delta_h_kernel, dim3(grid_for(rows, 256)), dim3(256), 0, stream, g, gp, gi.sb, gi.sn, gi.sh, (const half_t*)o, delta);
	HIP_ENFORCE(hipGetLastError());
#define SDPA_BY_GS(KERNEL, DS, ...) do { switch (g.Dv / 32) { \
		case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(KERNEL<DS, 2>), __VA_ARGS__); break; case 2: hipLaunchKernelGGL(HIP_KERNEL_NAME(KERNEL<DS, 4>), __VA_ARGS__); break; \
		case 3: hipLaunchKernelGGL(HIP_KERNEL_NAME(KERNEL<DS, 6>), __VA_ARGS__); break; default: hipLaunchKernelGGL(HIP_KERNEL_NAME(KERNEL<DS, 8>), __VA_ARGS__); break; } } while (0)
#define SDPA_BY_DS(KERNEL, ...) do { switch (g.D / 32) { \
		case 1: SDPA_BY_GS(KERNEL, 2, __VA_ARGS__); break; case 2: SDPA_BY_GS(KERNEL, 4, __VA_ARGS__); break; \
		case 3: SDPA_BY_GS(KERNEL, 6, __VA_ARGS__); break; default: SDPA_BY_GS(KERNEL, 8, __VA_ARGS__); break; } } while (0)
	if (dq) {
		note_kernel("sdpa_dq_f16");
		ProfScope prof("sdpa_dq_h|nnc::sdpa_dq_f16_kernel", 2.0 * g.B * g.Hq * (double)g.R * g.C * (2 * g.D + g.Dv), 0, g.R, g.C, g.D, g.B * g.Hq, 1, stream);
		SDPA_BY_DS(sdpa_dq_f16_kernel, dim3((g.R + 127) / 128, g.Hq, g.B), dim3(256), 0, stream, g, qp, kp, vp, mp, gp, gi.sb, gi.sn, gi.sh, (const float*)lse, (const float*)delta, (half_t*)dq->data.u8, dqi.sb, dqi.sn, dqi.sh);
		HIP_ENFORCE(hipGetLastError());
	}
	if (dk || dv) {
		half_t* const dkp = dk ? (half_t*)dk->data.u8 : 0; half_t* const dvp = dv ? (half_t*)dv->data.u8 : 0;
		const long ksb = dk ? dki.sb : 0, ksn = dk ? dki.sn : 0, ksh = dk ? dki.sh : 0, vsb = dv ? dvi.sb : 0, vsn = dv ? dvi.sn : 0, vsh = dv ? dvi.sh : 0;
		note_kernel("sdpa_dkv_f16");
		ProfScope prof("sdpa_dkv_h|nnc::sdpa_dkv_f16_kernel", 2.0 * g.B * g.Hq * (double)g.R * g.C * (2 * g.D + 2 * g.Dv), 0, g.C, g.R, g.D, g.B * g.Hk, 1, stream);
		SDPA_BY_DS(sdpa_dkv_f16_kernel, dim3((g.C + 127) / 128, g.Hk, g.B), dim3(256), 0, stream, g, qp, kp, vp, mp, gp, gi.sb, gi.sn, gi.sh, (const float*)lse, (const float*)delta, dkp, ksb, ksn, ksh, dvp, vsb, vsn, vsh);
		HIP_ENFORCE(hipGetLastError());
	}
#undef SDPA_BY_DS
#undef SDPA_BY_GS
	return CCV_NNC_EXEC_SUCCESS;
}
static int _sdpa_back_any(EXEC_ARGS)
{
	if (!any_half_tensor(inputs, input_size, outputs, output_size)) {
		MarkerScope marker(cmd.cmd);
		return _sdpa_back(cmd, hint, flags, inputs, input_size, outputs, output_size, stream_context);
	}
	const int r = sdpa_back_half(cmd, hint, flags, inputs, input_size, outputs, output_size, stream_context);
	if (r != CCV_NNC_EXEC_NO_KERNEL) return r;
	return half_staged_exec(_sdpa_back, cmd, hint, flags, inputs, input_size, outputs, output_size, stream_context);
}

} // namespace

#define NNC_REG(CMD, BACKEND, EXEC) \
	extern "C" void _register_command_##CMD##_backend_##BACKEND(ccv_nnc_cmd_backend_registry_t* const registry) \
	{ registry->tensor_formats = CCV_TENSOR_FORMAT_NCHW | CCV_TENSOR_FORMAT_NHWC; registry->tensor_datatypes = CCV_32F; registry->tensor_memory = CCV_TENSOR_GPU_MEMORY; registry->algorithms = 1; registry->exec = EXEC; NNC_HALF_STAGED(registry, EXEC); }

extern "C" void _register_command_CCV_NNC_SCALED_DOT_PRODUCT_ATTENTION_FORWARD_backend_CCV_NNC_BACKEND_GPU_REF(ccv_nnc_cmd_backend_registry_t* const registry)
{ registry->tensor_formats = CCV_TENSOR_FORMAT_NCHW | CCV_TENSOR_FORMAT_NHWC; registry->tensor_datatypes = CCV_32F | CCV_16F; registry->tensor_memory = CCV_TENSOR_GPU_MEMORY; registry->algorithms = 1; registry->exec = _sdpa_forw_any; NNC_DEPALETTIZED(registry, _sdpa_forw_any); /* palettized head-projection weights (ccv_nnc_scaled_dot_product_attention_flash_attn.cu:123, 459) */ }
extern "C" void _register_command_CCV_NNC_SCALED_DOT_PRODUCT_ATTENTION_BACKWARD_backend_CCV_NNC_BACKEND_GPU_REF(ccv_nnc_cmd_backend_registry_t* const registry)
{ registry->tensor_formats = CCV_TENSOR_FORMAT_NCHW | CCV_TENSOR_FORMAT_NHWC; registry->tensor_datatypes = CCV_32F | CCV_16F; registry->tensor_memory = CCV_TENSOR_GPU_MEMORY; registry->algorithms = 1; registry->exec = _sdpa_back_any; NNC_DEPALETTIZED(registry, _sdpa_back_any); /* (:342, 470) */ }
