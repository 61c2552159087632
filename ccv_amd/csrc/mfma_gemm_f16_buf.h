// Half-precision contraction of two PLAIN matrices with no VALU in the K loop (the half-precision sibling of BufMatLoader in mfma_gemm.h):
//   * operands arrive by 16-byte buffer loads -- eight halves of a row (k-contiguous operand) or eight rows of one k (row-contiguous operand, e.g. the
//     [C][P] activation plane of a 1x1 convolution on NCHW tensors) -- at a per-lane offset that never changes plus the K-step's wave-uniform soffset;
//   * a k-contiguous operand is staged as [rows][40 halves] (one ds_read_b128 per fragment, as in mfma_gemm_f16.h); a ROW-contiguous operand is staged
//     as it comes, [32 k][160 halves] (one ds_write_b128 per chunk instead of mfma_gemm_f16.h's eight ds_write_b16), and its fragments -- eight
//     consecutive k of one row per lane -- are gathered by the LDS transpose read of gfx950, ds_read_b64_tr_b16: each 16-lane group reads a
//     [4 k][16 rows] block (lane i supplies the address of four consecutive rows of k = i >> 2) and lane i receives the four k of row i.  Pitch 160
//     halves = 320 bytes: the four k rows of a block land 16 banks apart (80 dwords mod 64), conflict-free;
//   * rows / columns beyond the matrix: an out-of-range chunk reads zeros (buffer range check); a chunk that straddles the end of a row-contiguous
//     operand's rows reads its neighbours -- garbage that only ever meets output columns >= N, which the epilogue does not store.
// Block tile (32 TM WM) x (32 TN WN): WM x WN waves of TM x TN MFMA tiles each, BK = 32 (two sub-steps of v_mfma_f32_32x32x16_f16 per tile).  Two shapes are
// instantiated: 128 x 128 (2 x 2 waves of 2 x 2 tiles: 40 KB of LDS, three workgroups per CU) and 256 x 256 (2 x 4 waves of 4 x 2 tiles, 80 KB) -- with 64 x 64
// per wave the four waves' fragment reads alone (1 KB per MFMA) keep the LDS busy every cycle the MFMAs run; 128 x 64 per wave needs 3/4 of that.
// Both operands use the k map of mfma_gemm_f16.h: lane (row = l & 31, half = l >> 5) supplies k = 16 s + 8 half + 0..7 of sub-step s.
#pragma once
#include "mfma_gemm_f16.h"

namespace nnc {

template <bool KC, int ROWS, int NT, int BK>
struct FetchH16 {
	static constexpr int KPITCH = BK + 8; // halves per row of a k-contiguous operand's image (80 / 144 bytes: 16-byte aligned rows, 20 / 36-dword stride: ds_read_b128 conflict-free)
	static constexpr int NPITCH = gemm16_npitch(ROWS);
	static constexpr int NCH = ROWS * (BK / 8) / NT; // 16-byte chunks per thread and K-step
	static_assert(ROWS * (BK / 8) % NT == 0 && NCH >= 1, "whole chunks per thread");
	typedef unsigned int u4 __attribute__((ext_vector_type(4)));
	static constexpr int LDS_HALVES = KC ? ROWS * KPITCH : BK * NPITCH;
	__amdgpu_buffer_rsrc_t rs;
	unsigned voff[NCH];
	unsigned kscale;
	__device__ __forceinline__ void init(const BufMatLoader<KC>& l, const int row0, const int t)
	{
		const half_t* const p = (const half_t*)l.p;
		const long extent = KC ? (long)(l.R - 1) * l.ldr + l.K : (long)(l.K - 1) * l.ldk + l.R; // halves from p to the end of the matrix
		const long base = KC ? (long)row0 * l.ldr : (long)row0;
		const long left = (extent - base) * 2;
		rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p + base), 0, (unsigned)(left > 0x7fffffffL ? 0x7fffffffL : (left < 0 ? 0 : left)), 0x00020000);
		kscale = KC ? 2u : (unsigned)l.ldk * 2u;
#pragma unroll
		for (int jj = 0; jj < NCH; jj++) {
			const int id = t + NT * jj;
			if (KC) { const int r = id / (BK / 8); voff[jj] = row0 + r < l.R ? (unsigned)r * (unsigned)l.ldr * 2u + (unsigned)(id % (BK / 8)) * 16u : 0x80000000u; }
			else { const int k = id / (ROWS / 8), r = (id % (ROWS / 8)) << 3; voff[jj] = row0 + r < l.R ? (unsigned)k * (unsigned)l.ldk * 2u + (unsigned)r * 2u : 0x80000000u; }
		}
	}
	__device__ __forceinline__ void issue(u4 (&r)[NCH], const int kbase) const
	{
#pragma unroll
		for (int jj = 0; jj < NCH; jj++) r[jj] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[jj], (unsigned)kbase * kscale, 0);
	}
	__device__ __forceinline__ void store(half_t* const lds, const u4 (&r)[NCH], const int t) const
	{
#pragma unroll
		for (int jj = 0; jj < NCH; jj++) {
			const int id = t + NT * jj;
			if (KC) *(u4*)(lds + (id / (BK / 8)) * KPITCH + ((id % (BK / 8)) << 3)) = r[jj];
			else *(u4*)(lds + (id / (ROWS / 8)) * NPITCH + ((id % (ROWS / 8)) << 3)) = r[jj];
		}
	}
	// the fragment of sub-step s for the 32 rows starting at `base`: k = 16 s + 8 lh + 0..7 of row base + li
	__device__ __forceinline__ static halfx8 frag(const half_t* const s_, const int base, const int li, const int lh, const int s)
	{
		if (KC) return *(const halfx8*)(s_ + (base + li) * KPITCH + 16 * s + 8 * lh);
		const half_t* const blk = s_ + (16 * s + 8 * lh) * NPITCH + base + 16 * (li >> 4);
		const halfx4 lo = tr_read4(blk, NPITCH, li & 15), hi = tr_read4(blk + 4 * NPITCH, NPITCH, li & 15);
		return halfx8{ lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3] };
	}
};

// grid: x = tiles (* split-K slices), XCD-swizzled exactly as mfma_gemm_f16_kernel; z = batch.  K and the K-slices are whole K-steps (the host checks).
template <bool AKC, bool BKC, class EPI, int TM = 2, int TN = 2, int WM = 2, int WN = 2, int BK = 32>
__global__ void __launch_bounds__(64 * WM * WN) mfma_gemm_f16_buf_kernel(BufMatLoader<AKC> la, BufMatLoader<BKC> lb, EPI epi, const int tiles_m, const int tiles_n, const int K, const int k_per_split, const int splits, const long a_zoff, const long b_zoff, const long c_zoff, const long bias_zoff)
{
	constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, NT = 64 * WM * WN;
	typedef FetchH16<AKC, BM, NT, BK> FA;
	typedef FetchH16<BKC, BN, NT, BK> FB;
	constexpr int A_HALVES = FA::LDS_HALVES, B_HALVES = FB::LDS_HALVES;
	__shared__ __attribute__((aligned(16))) half_t lds[2][A_HALVES + B_HALVES];
	const int t = threadIdx.x;
	const int lane = t & 63, wave = t >> 6;
	const int wm = wave / WN, wn = wave % WN;
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
	la.p = (const float*)((const half_t*)la.p + (long)zi * a_zoff);
	lb.p = (const float*)((const half_t*)lb.p + (long)zi * b_zoff);
	epi.c += (long)zi * c_zoff;
	if (splits > 1) epi.c += (long)slice * M_N_slab(epi);
	const int k_begin = slice * k_per_split;
	const int k_end = (k_begin + k_per_split < K) ? k_begin + k_per_split : K;
	const int nk = (k_end - k_begin + BK - 1) / BK;
	FA fa;
	FB fb;
	fa.init(la, m0, t);
	fb.init(lb, n0, t);
	floatx16 acc[TM][TN];
#pragma unroll
	for (int i = 0; i < TM; i++)
#pragma unroll
		for (int j = 0; j < TN; j++)
#pragma unroll
			for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
	const int row_a = wm * (32 * TM), col_b = wn * (32 * TN);
	// Two register sets: the loads of tile kt + 2 go out at the top of K-step kt (into the set tile kt came through), tile kt + 1 -- loaded a whole K-step
	// ago -- is written to the other LDS buffer behind the MFMAs.  Pairs of K-steps with the set index a compile-time constant.
	typename FA::u4 ra[2][FA::NCH], rb[2][FB::NCH];
	if (nk > 0) {
		fa.issue(ra[0], k_begin);
		fb.issue(rb[0], k_begin);
		if (nk > 1) { fa.issue(ra[1], k_begin + BK); fb.issue(rb[1], k_begin + BK); }
		fa.store(lds[0], ra[0], t);
		fb.store(lds[0] + A_HALVES, rb[0], t);
	}
	__syncthreads();
	auto kstep = [&](auto sid, const int kt) {
		constexpr int S = decltype(sid)::value;
		if (kt + 2 < nk) {
			fa.issue(ra[S], k_begin + (kt + 2) * BK);
			fb.issue(rb[S], k_begin + (kt + 2) * BK);
		}
		const half_t* const sa = lds[S];
		const half_t* const sb = lds[S] + A_HALVES;
#pragma unroll
		for (int s = 0; s < BK / 16; s++) {
			halfx8 fa8[TM], fb8[TN];
#pragma unroll
			for (int ti = 0; ti < TM; ti++) fa8[ti] = FA::frag(sa, row_a + 32 * ti, li, lh, s);
#pragma unroll
			for (int tj = 0; tj < TN; tj++) fb8[tj] = FB::frag(sb, col_b + 32 * tj, li, lh, s);
			// the transpose reads are asm: hipcc does not count them (the operands tie the MFMAs below behind the wait)
			if (!AKC || !BKC) {
				NNC_WAIT_LGKM0();
#pragma unroll
				for (int ti = 0; ti < TM; ti++) NNC_PIN_VEC(fa8[ti]);
#pragma unroll
				for (int tj = 0; tj < TN; tj++) NNC_PIN_VEC(fb8[tj]);
			}
#pragma unroll
			for (int ti = 0; ti < TM; ti++)
#pragma unroll
				for (int tj = 0; tj < TN; tj++) acc[ti][tj] = nnc_mfma_f16(fa8[ti], fb8[tj], acc[ti][tj]);
		}
		if (kt + 1 < nk) {
			fa.store(lds[S ^ 1], ra[S ^ 1], t);
			fb.store(lds[S ^ 1] + A_HALVES, rb[S ^ 1], t);
		}
		__syncthreads();
	};
	{
		int kt = 0;
		for (; kt + 1 < nk; kt += 2) {
			kstep(GroupId<0>(), kt);
			kstep(GroupId<1>(), kt + 1);
		}
		if (kt < nk) kstep(GroupId<0>(), kt);
	}
	// D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
	if (epi.bias) epi.bias += (long)zi * bias_zoff;
	if (epi.vec) { // through LDS, one tile row of every wave per pass (mfma_gemm.h: "epilogues", epi_flush_rows)
		constexpr int PITCH = BN + 8;
		static_assert(32 * WM * PITCH * 2 <= 2 * (A_HALVES + B_HALVES), "the staged slice (fp32) fits the operand buffers");
		float* const cs = (float*)&lds[0][0];
#pragma unroll
		for (int ti = 0; ti < TM; ti++) {
			__syncthreads();
#pragma unroll
			for (int tj = 0; tj < TN; tj++)
#pragma unroll
				for (int r = 0; r < 16; r++) cs[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * PITCH + col_b + 32 * tj + li] = acc[ti][tj][r];
			__syncthreads();
			epi_flush_rows<NT, 32 * WM, BN>(cs, epi, m0, n0, t, [&](const int sr) { return (sr >> 5) * (32 * TM) + 32 * ti + (sr & 31); });
		}
		return;
	}
#pragma unroll
	for (int ti = 0; ti < TM; ti++)
#pragma unroll
		for (int tj = 0; tj < TN; tj++) {
			const int n = n0 + col_b + 32 * tj + li;
#pragma unroll
			for (int r = 0; r < 16; r++) {
				const int m = m0 + row_a + 32 * ti + (r & 3) + 8 * (r >> 2) + 4 * lh;
				epi(m, n, acc[ti][tj][r]);
			}
		}
}

} // namespace nnc
