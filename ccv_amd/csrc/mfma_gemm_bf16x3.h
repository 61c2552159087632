// fp32 contraction of two PLAIN matrices on the bf16 matrix pipe (round 6).  Every fp32 operand element is split EXACTLY into three bf16 values
//     x = hi + mid + lo        hi = x with its low 16 bits cleared, mid = (x - hi) likewise, lo = x - hi - mid
// (24 significand bits = 8 + 8 + 8: the two subtractions are exact and lo has at most 8 significant bits, so no rounding happens anywhere in the split),
// and ALL NINE partial products a_i * b_j are accumulated in fp32 by v_mfma_f32_32x32x16_bf16 -- each product of two 8-bit significands is exact in fp32,
// so the result differs from the fp32 matrix instruction's only by the order of the fp32 additions.  Nothing is dropped: "x6" / "x3" variants would be
// narrower arithmetic than the reference's fp32 loops.
// Why: the fp32 matrix instructions run at 1/16 of the bf16 rate AND share their datapath with the VALU (tools/phase_probe.cpp); nine bf16 instructions of
// 32 x 32 x 16 (9 x 32 = 288 cycles) replace eight fp32 ones of 32 x 32 x 2 (512 cycles), and the split's VALU work (5.5 instructions per element, once per
// operand tile) overlaps them on the other pipe.  Peak in fp32-equivalent arithmetic: 2.5 PF / 9 = 278 TFLOP/s against 157.3.
//   * operands arrive as in TileFetchBuf (mfma_gemm.h): 16-byte buffer loads at a per-lane offset that never changes plus the K-step's wave-uniform soffset,
//     rows beyond the matrix read zeros (range check); K and the K-slices are whole K-steps of 16;
//   * each operand is staged as THREE bf16 images.  k-contiguous operand: [rows][24 halves] (48-byte rows: a lane's fragment -- eight k of one row -- is one
//     ds_read_b128, conflict-free); row-contiguous operand: [16 k][rows + 32 halves], gathered by ds_read_b64_tr_b16 exactly as mfma_gemm_f16_buf.h does;
//   * block tile (32 TM WM) x (32 TN WN), two LDS stages (the tile of K-step kt + 1 is split and written behind the MFMAs of K-step kt), two register sets
//     of raw fp32 chunks in flight.  256 x 256 (2 x 4 waves of 4 x 2 tiles): 144 KB of LDS, one workgroup of eight waves per CU; 128 x 128 (2 x 2 waves of
//     2 x 2 tiles): 72 KB, two workgroups per CU.
// Inf / NaN: an infinite operand element becomes NaN (inf - inf in the split) where the fp32 instruction would keep the infinity.  Finite data is exact for
// |x| >= 2^-102; below that the last remainder's low bits fall under bf16's smallest denormal (an absolute error below 2^-133 per element).
#pragma once
#include "mfma_gemm_f16_buf.h"

namespace nnc {

struct SplitRegs { unsigned int hi[2], mid[2], lo[2], ya, yb; }; // a chunk on its way through the split (FetchSplit3::split_part)

template <bool KC, int ROWS, int NT, int BK>
struct FetchSplit3 {
	typedef SplitRegs Split;
	static_assert(BK == 16, "one 32x32x16 sub-step per K-step");
	static constexpr int KPITCH = BK + 8; // halves per row of a k-contiguous operand's image: 48 bytes = 12 dwords, ds_read_b128 of 16 consecutive rows covers the 64 banks once
	static constexpr int NPITCH = gemm16_npitch(ROWS);
	static constexpr int IMG = KC ? ROWS * KPITCH : BK * NPITCH; // halves per image
	static constexpr int LDS_HALVES = 3 * IMG;
	static constexpr int NCH = ROWS * (BK / 4) / NT; // 16-byte chunks (four floats) per thread and K-step
	static_assert(ROWS * (BK / 4) % NT == 0 && NCH >= 1, "whole chunks per thread");
	typedef unsigned int u4 __attribute__((ext_vector_type(4)));
	typedef unsigned int u2 __attribute__((ext_vector_type(2)));
	__amdgpu_buffer_rsrc_t rs;
	unsigned voff[NCH];
	unsigned kscale;
	__device__ __forceinline__ void init(const BufMatLoader<KC>& l, const int row0, const int t)
	{
		const long extent = KC ? (long)(l.R - 1) * l.ldr + l.K : (long)(l.K - 1) * l.ldk + l.R; // floats from p to the end of the matrix
		const long base = KC ? (long)row0 * l.ldr : (long)row0;
		const long left = (extent - base) * 4;
		rs = __builtin_amdgcn_make_buffer_rsrc((void*)(l.p + base), 0, (unsigned)(left > 0x7fffffffL ? 0x7fffffffL : (left < 0 ? 0 : left)), 0x00020000);
		kscale = KC ? 4u : (unsigned)l.ldk * 4u;
#pragma unroll
		for (int jj = 0; jj < NCH; jj++) {
			const int id = t + NT * jj;
			if (KC) { const int r = id / (BK / 4); voff[jj] = row0 + r < l.R ? (unsigned)r * (unsigned)l.ldr * 4u + (unsigned)(id % (BK / 4)) * 16u : 0x80000000u; }
			else { const int k = id / (ROWS / 4), r = (id % (ROWS / 4)) << 2; voff[jj] = row0 + r < l.R ? (unsigned)k * (unsigned)l.ldk * 4u + (unsigned)r * 4u : 0x80000000u; }
		}
	}
	__device__ __forceinline__ void issue(u4 (&r)[NCH], const int kbase) const
	{
#pragma unroll
		for (int jj = 0; jj < NCH; jj++) r[jj] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[jj], (unsigned)kbase * kscale, 0);
	}
	// Splitting one chunk of four floats into its three bf16 images (two packed dwords each) and writing them, in five parts the kernel deals between its
	// MFMAs: 0 / 2 = the hi pack and the first remainders of pair 0 / 1 (5 VALU), 1 / 3 = the mid and lo packs of that pair (6 VALU), 4 = three ds_write_b64.
	template <int PART>
	__device__ __forceinline__ void split_part(unsigned short* const lds, Split& s, const u4& r, const int t, const int jj) const
	{
		constexpr int p = (PART >> 1) & 1;
		if (PART < 4 && (PART & 1) == 0) {
			const unsigned xa = r[2 * p], xb = r[2 * p + 1];
			s.hi[p] = nnc_pack_hi16(xb, xa);
			s.ya = __float_as_uint(__uint_as_float(xa) - __uint_as_float(xa & 0xffff0000u));
			s.yb = __float_as_uint(__uint_as_float(xb) - __uint_as_float(xb & 0xffff0000u));
		} else if (PART < 4) {
			s.mid[p] = nnc_pack_hi16(s.yb, s.ya);
			const float sa = __uint_as_float(s.ya) - __uint_as_float(s.ya & 0xffff0000u), sb = __uint_as_float(s.yb) - __uint_as_float(s.yb & 0xffff0000u);
			s.lo[p] = nnc_pack_hi16(__float_as_uint(sb), __float_as_uint(sa));
		} else {
			const int id = t + NT * jj;
			unsigned short* const at = KC ? lds + (id / (BK / 4)) * KPITCH + ((id % (BK / 4)) << 2) : lds + (id / (ROWS / 4)) * NPITCH + ((id % (ROWS / 4)) << 2);
			*(u2*)at = u2{ s.hi[0], s.hi[1] };
			*(u2*)(at + IMG) = u2{ s.mid[0], s.mid[1] };
			*(u2*)(at + 2 * IMG) = u2{ s.lo[0], s.lo[1] };
		}
	}
	__device__ __forceinline__ void store_chunk(unsigned short* const lds, const u4& r, const int t, const int jj) const
	{
		Split s;
		split_part<0>(lds, s, r, t, jj); split_part<1>(lds, s, r, t, jj); split_part<2>(lds, s, r, t, jj); split_part<3>(lds, s, r, t, jj); split_part<4>(lds, s, r, t, jj);
	}
	__device__ __forceinline__ void store(unsigned short* const lds, const u4 (&r)[NCH], const int t) const
	{
#pragma unroll
		for (int jj = 0; jj < NCH; jj++) store_chunk(lds, r[jj], t, jj);
	}
	// the fragment (k = 8 lh + 0..7 of row base + li) of image `img` (0 hi, 1 mid, 2 lo)
	__device__ __forceinline__ static bf16x8_t frag(const unsigned short* const s_, const int img, const int base, const int li, const int lh)
	{
		const unsigned short* const s = s_ + img * IMG;
		if (KC) return *(const bf16x8_t*)(s + (base + li) * KPITCH + 8 * lh);
		const half_t* const blk = (const half_t*)s + (8 * lh) * NPITCH + base + 16 * (li >> 4);
		const halfx4 lo = tr_read4(blk, NPITCH, li & 15), hi = tr_read4(blk + 4 * NPITCH, NPITCH, li & 15);
		const halfx8 v = { lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3] };
		return __builtin_bit_cast(bf16x8_t, v);
	}
};

// grid: x = tiles (* split-K slices), XCD-swizzled exactly as mfma_gemm_f16_buf_kernel; z = batch.  K and the K-slices are whole K-steps (the host checks).
template <bool AKC, bool BKC, class EPI, int TM, int TN, int WM, int WN>
__global__ void __launch_bounds__(64 * WM * WN) mfma_gemm_bf16x3_kernel(BufMatLoader<AKC> la, BufMatLoader<BKC> lb, EPI epi, const int tiles_m, const int tiles_n, const int K, const int k_per_split, const int splits, const long a_zoff, const long b_zoff, const long c_zoff, const long bias_zoff)
{
	constexpr int BK = 16;
	constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, NT = 64 * WM * WN;
	typedef FetchSplit3<AKC, BM, NT, BK> FA;
	typedef FetchSplit3<BKC, BN, NT, BK> FB;
	constexpr int A_HALVES = FA::LDS_HALVES, B_HALVES = FB::LDS_HALVES;
	__shared__ __attribute__((aligned(16))) unsigned short lds[2][A_HALVES + B_HALVES];
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
	la.p += (long)zi * a_zoff;
	lb.p += (long)zi * b_zoff;
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
	// Two register sets of raw chunks: the loads of tile kt + 2 go out at the top of K-step kt (into the set tile kt came through); tile kt + 1 -- loaded a
	// whole K-step ago -- is split and written to the other LDS stage between the MFMA groups.
	typename FA::u4 ra[2][FA::NCH], rb[2][FB::NCH];
	if (nk > 0) {
		fa.issue(ra[0], k_begin);
		fb.issue(rb[0], k_begin);
		if (nk > 1) { fa.issue(ra[1], k_begin + BK); fb.issue(rb[1], k_begin + BK); }
		fa.store(lds[0], ra[0], t);
		fb.store(lds[0] + A_HALVES, rb[0], t);
	}
	__syncthreads();
	// One K-step: per tile row ti of the wave nine groups of TN MFMAs (the nine products a_i * b_j, smallest terms first; consecutive instructions hit
	// different accumulators), and BETWEEN the groups, fenced so that hipcc keeps them there: the next tile's split + LDS writes in parts of 5 - 6 VALU, and the
	// fragment reads of tile row ti + 1.  VALU, LDS and the bf16 matrix pipe are different units: what sits between two MFMAs of a wave runs in their shadow
	// (left to itself hipcc puts each chunk's 22 VALU + 3 writes + 3 reads in one clump behind 18 MFMAs and waits for the reads right there: measured 0.47 - 0.53
	// of the bf16 peak; profiles/r06_v1_bf16x3_bench.txt).
	auto kstep = [&](auto sid, const int kt) {
		constexpr int S = decltype(sid)::value;
		const unsigned short* const sa = lds[S];
		const unsigned short* const sb = lds[S] + A_HALVES;
		const bool more = kt + 1 < nk;
		bf16x8_t fb8[3][TN], fa8[2][3];
#pragma unroll
		for (int i = 0; i < 3; i++)
#pragma unroll
			for (int tj = 0; tj < TN; tj++) fb8[i][tj] = FB::frag(sb, i, col_b + 32 * tj, li, lh);
#pragma unroll
		for (int i = 0; i < 3; i++) fa8[0][i] = FA::frag(sa, i, row_a, li, lh);
		if (kt + 2 < nk) {
			fa.issue(ra[S], k_begin + (kt + 2) * BK);
			fb.issue(rb[S], k_begin + (kt + 2) * BK);
		}
		constexpr int PIECES = FA::NCH + FB::NCH, NC = PIECES / TM, NMS = 5 * NC;
		static_assert(PIECES % TM == 0, "whole chunks per tile row");
#pragma unroll
		for (int ti = 0; ti < TM; ti++) {
			// the transpose reads are asm: hipcc does not count them (the operands tie the MFMAs below behind the wait)
			if (!AKC || !BKC) {
				NNC_WAIT_LGKM0();
#pragma unroll
				for (int i = 0; i < 3; i++) NNC_PIN_VEC(fa8[ti & 1][i]);
				if (ti == 0) {
#pragma unroll
					for (int i = 0; i < 3; i++)
#pragma unroll
						for (int tj = 0; tj < TN; tj++) NNC_PIN_VEC(fb8[i][tj]);
				}
			}
			SplitRegs st[NC];
#pragma unroll
			for (int o = 0; o < 9; o++) {
				constexpr int IA[9] = { 2, 2, 1, 1, 2, 0, 1, 0, 0 }, IB[9] = { 2, 1, 2, 1, 0, 2, 0, 1, 0 };
#pragma unroll
				for (int tj = 0; tj < TN; tj++) acc[ti][tj] = nnc_mfma_bf16(fa8[ti & 1][IA[o]], fb8[IB[o]][tj], acc[ti][tj]);
				if (o == 0 && ti + 1 < TM) {
#pragma unroll
					for (int i = 0; i < 3; i++) fa8[(ti + 1) & 1][i] = FA::frag(sa, i, row_a + 32 * (ti + 1), li, lh);
				}
				if (more) {
#pragma unroll
					for (int ms = o * NMS / 9; ms < (o + 1) * NMS / 9; ms++) {
						const int pc = ti * NC + ms / 5; // chunk: A's first, then B's
						auto part = [&](auto pid) {
							constexpr int P = decltype(pid)::value;
							if (pc < FA::NCH) fa.template split_part<P>(lds[S ^ 1], st[ms / 5], ra[S ^ 1][pc], t, pc);
							else fb.template split_part<P>(lds[S ^ 1] + A_HALVES, st[ms / 5], rb[S ^ 1][pc - FA::NCH], t, pc - FA::NCH);
						};
						switch (ms % 5) { case 0: part(GroupId<0>()); break; case 1: part(GroupId<1>()); break; case 2: part(GroupId<2>()); break; case 3: part(GroupId<3>()); break; default: part(GroupId<4>()); break; }
					}
				}
				__builtin_amdgcn_sched_barrier(0);
			}
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
