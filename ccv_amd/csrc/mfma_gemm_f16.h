// Half-precision MFMA contraction core for gfx950: C[M][N] = sum_k A(m,k) * B(k,n), CCV_16F operands, fp32 accumulation on
// v_mfma_f32_32x32x16_f16 (the double-K f16 form of gfx950: 8 halves per lane per operand, ~2.2 PFLOP/s peak -- 14x the fp32
// form), CCV_16F result.  The reference's trainers run in half precision (lib/nnc/cmd/convolution/gpu/
// ccv_nnc_conv_gpu_cudnn.cu:478-500 registers CCV_32F | CCV_16F; test/int/nnc/cifar.tests.c:473); this is that datapath for the
// contraction commands: GEMM forward / backward and convolution forward / data gradient / filter gradient as implicit GEMMs.
//
// It reuses the ADDRESS half of the fp32 core (mfma_gemm.h): the loader functors (plain matrix, im2col, dgrad weights, wgrad
// im2col) compute ELEMENT offsets and are indifferent to the element size, and TileFetch's chunk schedule (4 elements per
// chunk, BK = 32) carries over -- a chunk of halves is an 8-byte load.  What differs is everything after the load:
//   * both operands are staged in ONE LDS image shape, [rows][40 halves] (k contiguous, row stride 80 bytes): a lane's
//     fragment -- 8 consecutive k of its row -- is one 16-byte ds_read_b128, conflict-free across 16-lane groups (20-word row
//     stride).  A row-contiguous operand (4 rows x 1 k per chunk) is staged as it comes, [32 k][rows + 32 halves] (one ds_write_b64 per chunk; round 2
//     transposed it on the way in with four ds_write_b16), and its fragments are gathered by the LDS transpose read ds_read_b64_tr_b16 (tr_read4 below;
//     the layout and the bank arithmetic are mfma_gemm_f16_buf.h's).
//   * one K-step (32 deep) is two MFMAs per 32x32 tile instead of sixteen; with 14x the MFMA rate the kernel is bound by
//     operand traffic (HBM / L2 -> LDS), not by MFMA issue, so the steady state is left to hipcc's scheduler: global loads
//     of tile kt+1 are issued before the MFMAs of tile kt and written to the other LDS buffer after them.
// The lane -> k map of the instruction does not matter for correctness as long as A and B use the same one (the sum over
// k is symmetric): lane (row = l & 31, half = l >> 5) supplies k = 16 s + 8 half + 0..7 of K-sub-step s for both operands.
#pragma once
#include "mfma_gemm.h"

namespace nnc {


constexpr int GEMM16_LDK = 40; // halves per LDS row: 32 of a K-step + 8 of padding (80 bytes: 16-byte aligned rows, 20-word stride)


// Epilogues.  Direct store: c[m * ldm + n * ldn] = alpha * acc (+ bias[n]) (+ old c when accumulating), rounded to half.
struct EpiStoreH {
	half_t* c;
	long ldm, ldn;
	const half_t* bias; // bias[m * bias_ldm + n * bias_ldn]; may be null
	float alpha;
	int accumulate;
	int M, N;
	long bias_ldm, bias_ldn;
	int vec = 0; // 1: store4 allowed (mfma_gemm.h, "epilogues"): four halves of a row in one 8-byte store; 2: store8 -- EIGHT halves, 16 bytes (round 6: an 8-byte
	             // access moves 0.54 - 0.70 of the 16-byte rate, MI355X guide; the 1 x 1 convolutions that WRITE the wide tensor are bound by exactly these stores)
	static constexpr int FLUSH_UNROLL = 2;
	static constexpr bool PLANAR = false;
	__device__ __forceinline__ void operator()(int m, int n, float v) const
	{
		if (m < M && n < N) {
			const long o = (long)m * ldm + (long)n * ldn;
			v *= alpha;
			if (bias) v += (float)bias[(long)m * bias_ldm + (long)n * bias_ldn];
			if (accumulate) v += (float)c[o];
			c[o] = (half_t)v;
		}
	}
	__device__ __forceinline__ void store8(int m, int n, const float4 lo, const float4 hi) const
	{ // same arithmetic per element as store4 / operator(): bit-identical results
		if (m < M && n < N) {
			const long o = (long)m * ldm + (long)n;
			float v[8] = { lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w };
#pragma unroll
			for (int e = 0; e < 8; e++) v[e] *= alpha;
			if (bias) {
				const half_t* const b = bias + (long)m * bias_ldm + (long)n * bias_ldn;
#pragma unroll
				for (int e = 0; e < 8; e++) v[e] += (float)b[e * bias_ldn];
			}
			if (accumulate) {
				const halfx8 u = *(const halfx8*)(c + o);
#pragma unroll
				for (int e = 0; e < 8; e++) v[e] += (float)u[e];
			}
			*(halfx8*)(c + o) = halfx8{ (half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3], (half_t)v[4], (half_t)v[5], (half_t)v[6], (half_t)v[7] };
		}
	}
	__device__ __forceinline__ void store4(int m, int n, float4 v) const
	{
		if (m < M && n < N) {
			const long o = (long)m * ldm + (long)n;
			v.x *= alpha; v.y *= alpha; v.z *= alpha; v.w *= alpha;
			if (bias) {
				const half_t* const b = bias + (long)m * bias_ldm + (long)n * bias_ldn;
				v.x += (float)b[0]; v.y += (float)b[bias_ldn]; v.z += (float)b[2 * bias_ldn]; v.w += (float)b[3 * bias_ldn];
			}
			if (accumulate) { const halfx4 u = *(const halfx4*)(c + o); v.x += (float)u[0]; v.y += (float)u[1]; v.z += (float)u[2]; v.w += (float)u[3]; }
			*(halfx4*)(c + o) = halfx4{ (half_t)v.x, (half_t)v.y, (half_t)v.z, (half_t)v.w };
		}
	}
};
// Split-K partial (fp32 slabs, exactly as the fp32 core's): splitk_reduce_half_kernel finishes.
struct EpiPartialH {
	float* c;
	const half_t* bias; // unused
	long slab;
	int M, N;
	int vec = 0;
	static constexpr int FLUSH_UNROLL = 1;
	static constexpr bool PLANAR = false;
	__device__ __forceinline__ void operator()(int m, int n, float v) const
	{
		if (m < M && n < N) c[(long)m * N + n] = v;
	}
	__device__ __forceinline__ void store4(int m, int n, const float4 v) const
	{
		if (m < M && n < N) *(float4*)(c + (long)m * N + n) = v;
	}
};
// Planar store: the contraction's rows are the pixels m = (image, p) of an NHWC view, its columns the channels n, and the result goes to an NCHW tensor
// c[image][n][p] -- the convolutions of the half-precision trainers (NCHW tensors) write their output where it belongs instead of into an NHWC scratch image that a
// transpose pass then re-lays (one read + one write of the output less per forward / data-gradient command).  The kernel stages each 64-row slice of the block tile
// TRANSPOSED in LDS ([n][m], pitch 65: conflict-free both ways) and a lane writes four consecutive p of one channel plane in one 8-byte store; P % 4 == 0 keeps such a
// group inside one image.  Bias per channel n; no alpha / accumulate (the callers have none).
struct EpiStoreHT {
	half_t* c;
	const half_t* bias; // bias[n]; may be null
	int M, N, P;        // N = channels of the output tensor, P = pixels per image
	FastDiv d_p;
	static constexpr bool PLANAR = true;
	static constexpr int FLUSH_UNROLL = 2;
	int vec = 1; // 2: groups of EIGHT pixels of a plane per store (16 bytes; P % 8 == 0 and a 16-byte aligned tensor: the launcher checks)
	__device__ __forceinline__ void store8p(const int m, const int n, const float4 lo, const float4 hi) const
	{
		if (m < M && n < N) { // M % 8 == 0 (whole images of P % 8 == 0 pixels)
			const int img = d_p.div(m);
			const float b = bias ? (float)bias[n] : 0.f;
			*(halfx8*)(c + ((long)img * N + n) * P + (m - img * P)) = halfx8{ (half_t)(lo.x + b), (half_t)(lo.y + b), (half_t)(lo.z + b), (half_t)(lo.w + b), (half_t)(hi.x + b), (half_t)(hi.y + b), (half_t)(hi.z + b), (half_t)(hi.w + b) };
		}
	}
	__device__ __forceinline__ void operator()(int, int, float) const {}          // (the kernel's other two ways out are never taken for a planar epilogue:
	__device__ __forceinline__ void store4(int, int, const float4) const {}       //  they only have to compile)
	__device__ __forceinline__ void store4p(const int m, const int n, const float4 v) const
	{
		if (m < M && n < N) { // M % 4 == 0 (whole images of P % 4 == 0 pixels)
			const int img = d_p.div(m);
			const float b = bias ? (float)bias[n] : 0.f;
			*(halfx4*)(c + ((long)img * N + n) * P + (m - img * P)) = halfx4{ (half_t)(v.x + b), (half_t)(v.y + b), (half_t)(v.z + b), (half_t)(v.w + b) };
		}
	}
};
__device__ __forceinline__ long M_N_slab(const EpiStoreHT&) { return 0; }
__device__ __forceinline__ long M_N_slab(const EpiStoreH&) { return 0; }
__device__ __forceinline__ long M_N_slab(const EpiPartialH& e) { return e.slab; }

// halves per k row of a row-contiguous operand's LDS image: the tile's rows + 32 (160 / 96 halves = 80 / 48 dwords: the four k rows of a transpose-read
// block land 16 banks apart)
constexpr int gemm16_npitch(const int rows) { return rows + 32; }

// The chunks one thread stages for an operand tile, on top of TileFetch's address schedule.
template <class L, int NCH>
struct TileFetchH : TileFetch<L, NCH> {
	typedef TileFetch<L, NCH> Base;
	static_assert(L::VECTOR, "the half-precision core takes vector loaders only (4-element chunks: 8-byte loads)");
	__device__ __forceinline__ void issue(const L& l, uint2 (&r)[NCH]) const
	{
#pragma unroll
		for (int jj = 0; jj < NCH; jj++) r[jj] = *(const uint2*)((const half_t*)l.p + this->off[jj]);
	}
	__device__ __forceinline__ void store(half_t* lds, const uint2 (&r)[NCH], const int t) const
	{
#pragma unroll
		for (int jj = 0; jj < NCH; jj++) {
			const int id = t + GEMM_THREADS * jj;
			if (L::KCONTIG) *(uint2*)(lds + (id >> 3) * GEMM16_LDK + ((id & 7) << 2)) = r[jj]; // row id >> 3, k (id & 7) * 4 .. + 3
			else *(uint2*)(lds + (id / (Base::ROWS / 4)) * gemm16_npitch(Base::ROWS) + ((id % (Base::ROWS / 4)) << 2)) = r[jj]; // k = id / (ROWS / 4), rows (id % (ROWS / 4)) * 4 .. + 3, as they come
		}
	}
};

// The same with EIGHT elements per chunk (round 4): one 16-byte load, one ds_write_b128 and one offset per chunk -- half the loads, LDS writes and address
// arithmetic per K-step of the form above.  The loaders do not care how wide a chunk is (they resolve the offset of its first element); what has to hold is that
// eight consecutive k (k-contiguous operand) or rows (row-contiguous operand) lie next to each other, 16-byte aligned -- channel counts and strides in
// multiples of eight: the launcher checks (loader_vec8_ok, gemm_launch.h).  The LDS images are the ones above, so the fragment reads do not change.
template <class L, int NCH> // NCH = ROWS / 64
struct TileFetchH8 {
	static constexpr int ROWS = NCH * 64;
	typedef unsigned int u4 __attribute__((ext_vector_type(4)));
	typedef u4 reg_t;
	static_assert(L::VECTOR, "vector loaders only");
	typename L::Ctx ctx[L::KCONTIG ? NCH : 1];
	int koff[NCH];
	long off[NCH];
	__device__ __forceinline__ void init(const L& l, const int row0, const int t)
	{
#pragma unroll
		for (int jj = 0; jj < NCH; jj++) {
			const int id = t + GEMM_THREADS * jj;
			if (L::KCONTIG) { ctx[L::KCONTIG ? jj : 0] = l.make(row0 + (id >> 2)); koff[jj] = (id & 3) << 3; } // row id >> 2, k (id & 3) * 8 .. + 7
			else { if (jj == 0) ctx[0] = l.make(row0 + ((id % (ROWS / 8)) << 3)); koff[jj] = id / (ROWS / 8); } // k = id / (ROWS / 8), rows (id % (ROWS / 8)) * 8 .. + 7
		}
	}
	template <bool FIRST>
	__device__ __forceinline__ void prep(const L& l, const int kbase, const int klimit)
	{
		if (L::KCONTIG) {
			const typename L::KCtx kc = l.kctx(kbase + koff[0], klimit);
#pragma unroll
			for (int jj = 0; jj < NCH; jj++) off[jj] = l.offset(ctx[L::KCONTIG ? jj : 0], kc);
		} else {
#pragma unroll
			for (int jj = 0; jj < NCH; jj++) off[jj] = l.offset(ctx[0], l.kctx(kbase + koff[jj], klimit));
		}
	}
	__device__ __forceinline__ void issue(const L& l, u4 (&r)[NCH]) const
	{
#pragma unroll
		for (int jj = 0; jj < NCH; jj++) r[jj] = *(const u4*)((const half_t*)l.p + off[jj]);
	}
	__device__ __forceinline__ void store(half_t* lds, const u4 (&r)[NCH], const int t) const
	{
#pragma unroll
		for (int jj = 0; jj < NCH; jj++) {
			const int id = t + GEMM_THREADS * jj;
			if (L::KCONTIG) *(u4*)(lds + (id >> 2) * GEMM16_LDK + ((id & 3) << 3)) = r[jj];
			else *(u4*)(lds + (id / (ROWS / 8)) * gemm16_npitch(ROWS) + ((id % (ROWS / 8)) << 3)) = r[jj];
		}
	}
};
// the fetcher and its register type by chunk width: W 32-row tiles per wave -> 64 W rows per operand tile
template <class L, int W, int CW> struct FetchHOf { typedef TileFetchH<L, W * 2> type; typedef uint2 reg_t; static constexpr int NCH = W * 2; };
template <class L, int W> struct FetchHOf<L, W, 8> { typedef TileFetchH8<L, W> type; typedef typename TileFetchH8<L, W>::u4 reg_t; static constexpr int NCH = W; };

// The fragment of sub-step s for the 32 rows starting at `base`: k = 16 s + 8 lh + 0..7 of row base + li, out of either image
template <bool KC, int ROWS>
__device__ __forceinline__ halfx8 frag16(const half_t* const s_, const int base, const int li, const int lh, const int s)
{
	if (KC) return *(const halfx8*)(s_ + (base + li) * GEMM16_LDK + 16 * s + 8 * lh);
	constexpr int NP = gemm16_npitch(ROWS);
	const half_t* const blk = s_ + (16 * s + 8 * lh) * NP + base + 16 * (li >> 4);
	const halfx4 lo = tr_read4(blk, NP, li & 15), hi = tr_read4(blk + 4 * NP, NP, li & 15);
	return halfx8{ lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3] };
}

// grid: x = tiles (* split-K slices), XCD-swizzled exactly as the fp32 core; z = batch / conv group.
template <class LA, class LB, class EPI, int WM, int WN, int CW = 4> // CW: elements per staged chunk (4: 8-byte loads; 8: 16-byte loads, TileFetchH8)
__global__ void __launch_bounds__(GEMM_THREADS) mfma_gemm_f16_kernel(LA la, LB lb, EPI epi, const int tiles_m, const int tiles_n, const int K, const int k_per_split, const int splits, const long a_zoff, const long b_zoff, const long c_zoff, const long bias_zoff, const KOrder ko)
{
	constexpr int BM = 64 * WM, BN = 64 * WN;
	constexpr int A_HALVES = LA::KCONTIG ? BM * GEMM16_LDK : GEMM_BK * gemm16_npitch(BM), B_HALVES = LB::KCONTIG ? BN * GEMM16_LDK : GEMM_BK * gemm16_npitch(BN);
	__shared__ __attribute__((aligned(16))) half_t lds[2][A_HALVES + B_HALVES];
	const int t = threadIdx.x;
	const int lane = t & 63, wave = t >> 6;
	const int wm = wave >> 1, wn = wave & 1;
	const int li = lane & 31, lh = lane >> 5;
	const int nwg = gridDim.x;
	const int bid = blockIdx.x;
	int tile, slice = 0, zi = (int)blockIdx.z;
	if (splits < 0) {
		if (!gemm_batch_xcd_map(bid, tiles_m * tiles_n, -splits, &tile, &zi)) return;
	} else {
		const int xcd = bid & 7, idx = bid >> 3;
		if (splits > 1) {
			const int tiles = tiles_m * tiles_n;
			const int j = idx / tiles;
			tile = idx - j * tiles;
			slice = xcd + 8 * j;
		} else {
			const int q = nwg >> 3, r = nwg & 7;
			tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
		}
	}
	const int tile_m = tile / tiles_n, tile_n = tile - tile_m * tiles_n;
	(void)tiles_m;
	const int m0 = tile_m * BM, n0 = tile_n * BN;
	// the loaders' base pointers are typed float* (mfma_gemm.h) but address halves here: offsets are in ELEMENTS
	la.p = (const float*)((const half_t*)la.p + (long)zi * a_zoff); la.zoff -= (long)zi * a_zoff;
	lb.p = (const float*)((const half_t*)lb.p + (long)zi * b_zoff); lb.zoff -= (long)zi * b_zoff;
	epi.c += (long)zi * c_zoff;
	if (splits > 1) epi.c += (long)slice * M_N_slab(epi);
	const int k_begin = slice * k_per_split;
	const int k_end = (k_begin + k_per_split < K) ? k_begin + k_per_split : K;
	const int nk = (k_end - k_begin + GEMM_BK - 1) / GEMM_BK;
	auto kmap = [&](const int kb) -> int {
		if (!ko.taps) return kb;
		if (kb >= k_end) return ko.K;
		const int s = kb / GEMM_BK;
		const int cc = ko.d.div(s);
		return (s - cc * ko.taps) * ko.C + cc * GEMM_BK;
	};
	const int klim = ko.taps ? ko.K : k_end;
	typename FetchHOf<LA, WM, CW>::type fa;
	typename FetchHOf<LB, WN, CW>::type fb;
	fa.init(la, m0, t);
	fb.init(lb, n0, t);
	floatx16 acc[WM][WN];
#pragma unroll
	for (int i = 0; i < WM; i++)
#pragma unroll
		for (int j = 0; j < WN; j++)
#pragma unroll
			for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
	const int row_a = wm * (32 * WM), col_b = wn * (32 * WN);
	// (Round 6 measured a second register set here -- the loads of tile kt + 2 issued at the top of K-step kt, the buffer kernel's schedule -- on the trainers'
	// shapes: no change for forward / data gradient (100.9 -> 100.3 us), the filter gradient 7 - 13 % slower, and the NHWC-output form lost a wave per SIMD to the
	// 16 extra registers; profiles/r06_v10_*_kernel_stats.md.  The K-step is not waiting for its loads: per eight MFMAs a wave issues ~100 other instructions
	// (the gather loaders' address arithmetic, fragment reads, LDS writes, the barrier), and three waves per SIMD make that the bound.  One set, as before.)
	typename FetchHOf<LA, WM, CW>::reg_t ra[FetchHOf<LA, WM, CW>::NCH];
	typename FetchHOf<LB, WN, CW>::reg_t rb[FetchHOf<LB, WN, CW>::NCH];
	if (nk > 0) {
		const int k0 = kmap(k_begin);
		fa.template prep<true>(la, k0, klim);
		fb.template prep<true>(lb, k0, klim);
		fa.issue(la, ra);
		fb.issue(lb, rb);
		fa.store(lds[0], ra, t);
		fb.store(lds[0] + A_HALVES, rb, t);
	}
	__syncthreads();
	for (int kt = 0; kt < nk; kt++) {
		const int cur = kt & 1;
		const bool more = kt + 1 < nk;
		if (more) { // next tile: addresses, then the loads go out ahead of this tile's MFMAs
			const int k1 = kmap(k_begin + (kt + 1) * GEMM_BK);
			fa.template prep<true>(la, k1, klim);
			fb.template prep<true>(lb, k1, klim);
			fa.issue(la, ra);
			fb.issue(lb, rb);
		}
		const half_t* const sa = lds[cur];
		const half_t* const sb = lds[cur] + A_HALVES;
#pragma unroll
		for (int s = 0; s < 2; s++) {
			halfx8 fa8[WM], fb8[WN];
#pragma unroll
			for (int ti = 0; ti < WM; ti++) fa8[ti] = frag16<LA::KCONTIG, BM>(sa, row_a + 32 * ti, li, lh, s);
#pragma unroll
			for (int tj = 0; tj < WN; tj++) fb8[tj] = frag16<LB::KCONTIG, BN>(sb, col_b + 32 * tj, li, lh, s);
			if (!LA::KCONTIG || !LB::KCONTIG) { // the transpose reads are asm: hipcc does not count them (the operands tie the MFMAs below behind the wait)
				NNC_WAIT_LGKM0();
#pragma unroll
				for (int ti = 0; ti < WM; ti++) NNC_PIN_VEC(fa8[ti]);
#pragma unroll
				for (int tj = 0; tj < WN; tj++) NNC_PIN_VEC(fb8[tj]);
			}
#pragma unroll
			for (int ti = 0; ti < WM; ti++)
#pragma unroll
				for (int tj = 0; tj < WN; tj++) acc[ti][tj] = nnc_mfma_f16(fa8[ti], fb8[tj], acc[ti][tj]);
		}
		if (more) {
			fa.store(lds[cur ^ 1], ra, t);
			fb.store(lds[cur ^ 1] + A_HALVES, rb, t);
		}
		__syncthreads();
	}
	// D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
	if (epi.bias) epi.bias += (long)zi * bias_zoff;
	if constexpr (EPI::PLANAR) { // NCHW output: each 64-row slice staged transposed ([n][m], pitch 65), read back along m (EpiStoreHT above)
		constexpr int PM = 65;
		static_assert(BN * PM * 2 <= 2 * (A_HALVES + B_HALVES), "the transposed slice (fp32) fits the operand buffers");
		float* const cs = (float*)&lds[0][0];
#pragma unroll
		for (int ti = 0; ti < WM; ti++) {
			__syncthreads();
#pragma unroll
			for (int tj = 0; tj < WN; tj++)
#pragma unroll
				for (int r = 0; r < 16; r++) cs[(col_b + 32 * tj + li) * PM + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh] = acc[ti][tj][r];
			__syncthreads();
			if (epi.vec == 2) {
#pragma unroll 2
				for (int j = 0; j < BN * 8 / GEMM_THREADS; j++) {
					const int id = t + GEMM_THREADS * j;
					const int nl = id >> 3, ml = (id & 7) << 3; // 8 groups of eight staged rows per channel
					const float* const q = cs + nl * PM + ml;
					epi.store8p(m0 + (ml >> 5) * (32 * WM) + 32 * ti + (ml & 31), n0 + nl, make_float4(q[0], q[1], q[2], q[3]), make_float4(q[4], q[5], q[6], q[7]));
				}
				continue;
			}
#pragma unroll 2
			for (int j = 0; j < BN * 16 / GEMM_THREADS; j++) {
				const int id = t + GEMM_THREADS * j;
				const int nl = id >> 4, ml = (id & 15) << 2; // 16 groups of four staged rows per channel
				const float* const q = cs + nl * PM + ml;
				epi.store4p(m0 + (ml >> 5) * (32 * WM) + 32 * ti + (ml & 31), n0 + nl, make_float4(q[0], q[1], q[2], q[3]));
			}
		}
		return;
	}
	if (epi.vec) { // through LDS, one tile row of every wave per pass (mfma_gemm.h: "epilogues", epi_flush_rows)
		constexpr int PITCH = BN + 8;
		static_assert(64 * PITCH * 2 <= 2 * (A_HALVES + B_HALVES), "the staged slice (fp32) fits the operand buffers");
		float* const cs = (float*)&lds[0][0];
#pragma unroll
		for (int ti = 0; ti < WM; ti++) {
			__syncthreads();
#pragma unroll
			for (int tj = 0; tj < WN; tj++)
#pragma unroll
				for (int r = 0; r < 16; r++) cs[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * PITCH + col_b + 32 * tj + li] = acc[ti][tj][r];
			__syncthreads();
			epi_flush_rows<GEMM_THREADS, 64, BN>(cs, epi, m0, n0, t, [&](const int sr) { return (sr >> 5) * (32 * WM) + 32 * ti + (sr & 31); });
		}
		return;
	}
#pragma unroll
	for (int ti = 0; ti < WM; ti++)
#pragma unroll
		for (int tj = 0; tj < WN; tj++) {
			const int n = n0 + col_b + 32 * tj + li;
#pragma unroll
			for (int r = 0; r < 16; r++) {
				const int m = m0 + row_a + 32 * ti + (r & 3) + 8 * (r >> 2) + 4 * lh;
				epi(m, n, acc[ti][tj][r]);
			}
		}
}

// (a split-K contraction into a half-precision tensor is finished by splitk_reduce_kernel<half_t>, mfma_gemm.h)

} // namespace nnc
