// Fused Winograd F(3x3, 4x4) FILTER GRADIENT for gfx950, fp32 (round 3): dw = A'^T ( sum over tiles W[z] (x) V[z] ) A' with
//   V = B^T d B   (d: the 6x6 input patch of a 4x4 output-gradient tile)      -- what wino_input_kernel wrote to HBM
//   W = G' e G'^T (e: the 4x4 output-gradient tile)                             -- what wino_outgrad_kernel wrote to HBM
// computed in registers by the lanes that feed them to the MFMAs: neither V (2.25x the activations) nor W (2.25x the gradient) exists in
// HBM.  On VGG-D at batch 256 the via-HBM form spent 14.6 ms of the 88 ms step writing and re-reading them, plus the
// 64 / 128-channel contractions at 91 - 110 TFLOP/s because they are HBM-bound.  Spec of the arithmetic: the via-HBM path (winograd.h
// wino_input_kernel / wino_outgrad_kernel / wino_wgrad_final_kernel), itself pinned against the reference's direct loops
// (lib/nnc/cmd/convolution/ccv_nnc_conv_cpu_ref.c:134-262) by tests/test_parity_ops.py and tests/test_parity_fullsize.py.
//
// Shape.  The reduction index of dU[z][k][c] = sum_t W[z][t][k] V[z][t][c] is the TILE, so on v_mfma_f32_16x16x4_f32 one instruction takes
// four tiles: A[m = k][kk = tile] = W, B[kk = tile][n = c] = V, D[k][c] += .  A lane (ch = l & 15, slot = l >> 4) therefore owns ONE tile and one
// channel index on each side: it reads the 16 gradient values of its (tile, k) and the 36 input values of its (tile, c) and transforms them
// itself -- no staging of transformed data, no cross-lane traffic.  Accumulators: 36 positions x 16 k x 32 c per wave = 72 MFMA tiles = 288
// registers (the same budget as the fused forward kernel: 60 tiles pinned to AGPRs through asm operand classes); a workgroup of four waves
// covers a 32 k x 64 c block, and the grid is (K / 32) x (C / 64) blocks x `slices` ranges of tile groups; every workgroup writes its
// partial dU once at the end, wino_wgrad_fused_fold_kernel folds the slices in a fixed order and winograd.h's wino_wgrad_final_kernel applies A'^T . A'.
//
// Data path.  A trip = one 2 x 4 group of tiles (two MFMA k-steps): the 10 x 18 pixel input region x 64 channels (45 KB: whole 128-byte
// lines) and the 8 x 16 pixel gradient region x 32 channels (16 KB) arrive by LDS-DMA (buffer_load ... lds: no VGPR round trip, out-of-image
// pixels are out-of-range offsets and land as zeros -- a clipped tile's gradient is zero, so it adds nothing), double-buffered, the 61 pieces
// of the NEXT trip spread between the MFMAs of the current one (wino_fused.h: a queue of vector-memory instructions blocks the in-order wave,
// MFMAs included; the CU retires one 1 KB piece per ~65 cycles).  One barrier per trip.  Per trip and wave: 144 MFMAs (4608 cycles), ~820
// transform VALU, 176 LDS reads, <= 16 DMA pieces.
#pragma once
#include "wino_fused.h"

namespace nnc {

typedef float f2v __attribute__((ext_vector_type(2))); // two fp32 in an aligned register pair: hipcc selects v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 for its arithmetic

constexpr int WG_KB = 32, WG_CB = 64;                  // output / input channels per workgroup
constexpr int WG_GH = 2, WG_GW = 4;                    // tiles per trip (rows x columns): row = one MFMA k-step of four tiles
constexpr int WG_RH = 4 * WG_GH + 2, WG_RW = 4 * WG_GW + 2; // input region, pixels
constexpr int WG_A_BYTES = WG_RH * WG_RW * WG_CB * 4;  // 46080
constexpr int WG_G_BYTES = (4 * WG_GH) * (4 * WG_GW) * WG_KB * 4; // 16384
constexpr int WG_A_PIECES = WG_A_BYTES / 1024, WG_G_PIECES = WG_G_BYTES / 1024; // 45, 16
constexpr int WG_STAGE_BYTES = WG_A_BYTES + WG_G_BYTES;
static_assert(WG_A_BYTES % 1024 == 0 && WG_G_BYTES % 1024 == 0 && 2 * WG_STAGE_BYTES <= 160 * 1024, "two stages of whole DMA pieces in LDS");
constexpr int WG_A_PER_WAVE = (WG_A_PIECES + 3) / 4, WG_G_PER_WAVE = WG_G_PIECES / 4; // 12 (the last wave-slot of some waves is empty), 4
constexpr int WG_PIECES_PER_WAVE = WG_A_PER_WAVE + WG_G_PER_WAVE;                    // 16 issue slots per wave and trip

struct WinoWgradFusedArgs {
	const float* a;      // input activations, NHWC, channels dense
	const float* g;      // output gradient, NHWC, channels dense
	float* partial;      // [slices][36][K][C]
	float* bias_partial; // [slices][4 tile slots][K], or null
	long a_sn, a_sh, a_sw, g_sn, g_sh, g_sw; // element strides
	int H, W, OH, OW;    // input / gradient extents
	int pad_y, pad_x;
	int GYn, GXn;        // tile groups per image column / row
	int groups;          // N * GYn * GXn
	int C, K;
	int kblocks, cblocks; // K / 32, C / 64
	int slices;           // a multiple of 8 (one eighth per XCD)
	int per_slice;        // tile groups per slice
	unsigned a_image_bytes, g_image_bytes; // ranges of the per-image buffer descriptors
};

template <int DBG = 0>
__global__ void __launch_bounds__(256, 1) wino_wgrad_fused_kernel(const WinoWgradFusedArgs p)
{
	__shared__ __attribute__((aligned(16))) float lds[2 * WG_STAGE_BYTES / 4 + 256]; // two stages + 1 KB where the empty issue slots (waves 1 - 3 have 11 input pieces, not 12) drop their zeros
	const int t = threadIdx.x, lane = t & 63;
	const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
	const int wk = wave >> 1, wc = wave & 1;
	const int ch = lane & 15, slot = lane >> 4;
	// workgroup -> (slice, k block, c block): the blocks of one slice read the same regions, so they sit on ONE XCD (workgroup b runs on XCD b % 8)
	const int nb = p.kblocks * p.cblocks;
	const int xcd = (int)blockIdx.x & 7, r = (int)blockIdx.x >> 3;
	const int slice = xcd * (p.slices >> 3) + r / nb, blk = r % nb;
	const int kb = blk / p.cblocks, cb = blk - kb * p.cblocks;
	int g_first = slice * p.per_slice, g_end = g_first + p.per_slice;
	if (g_end > p.groups) g_end = p.groups;
	const int gpi = p.GYn * p.GXn;

	floatx4 acc[36][2];
#pragma unroll
	for (int z = 0; z < 36; z++)
#pragma unroll
		for (int j = 0; j < 2; j++) acc[z][j] = floatx4{ 0.f, 0.f, 0.f, 0.f };
	float bsum = 0.f;

	// ---- DMA: this wave's pieces.  a piece q = wave + 4 i: region pixel 4 q + (lane >> 4), channels 4 (lane & 15) .. + 3 of the c block;
	// g piece q = wave + 4 i: gradient pixel 8 q + (lane >> 3), channels 4 (lane & 7) .. + 3 of the k block.  Item-independent halves kept in registers.
	int ayx[WG_A_PER_WAVE], gyx[WG_G_PER_WAVE];
#pragma unroll
	for (int i = 0; i < WG_A_PER_WAVE; i++) {
		const int q = wave + 4 * i, pi = q * 4 + (lane >> 4);
		ayx[i] = q < WG_A_PIECES ? ((pi / WG_RW) << 8 | (pi % WG_RW)) : 0xff00;
	}
#pragma unroll
	for (int i = 0; i < WG_G_PER_WAVE; i++) {
		const int pj = (wave + 4 * i) * 8 + (lane >> 3);
		gyx[i] = (pj / (4 * WG_GW)) << 8 | (pj % (4 * WG_GW));
	}
	const int a_coff = (cb * WG_CB + (lane & 15) * 4) * 4, g_coff = (kb * WG_KB + (lane & 7) * 4) * 4; // bytes
	const int a_sh4 = (int)p.a_sh * 4, a_sw4 = (int)p.a_sw * 4, g_sh4 = (int)p.g_sh * 4, g_sw4 = (int)p.g_sw * 4;
	const unsigned lds0 = __builtin_amdgcn_readfirstlane(wf_lds_addr(lds));
	wf_rsrc_t rs_a, rs_g;
	// the group being FETCHED: the descriptors start at the region's origin (which may lie before the image: only in-image lanes get an in-range
	// offset), the in-image part of the region as uniform ranges in region coordinates -- a piece's address work is two extracts, two multiply-adds,
	// four compares against scalars and a select (the first version recomputed image coordinates per lane: 27 instructions per piece, 430 per trip)
	int aY0 = 0, aY1 = 0, aX0 = 0, aX1 = 0, gY1 = 0, gX1 = 0;
	auto set_group = [&](const int group) {
		const int n = group / gpi, gr = group - n * gpi;
		const int gy = gr / p.GXn, gx = gr - gy * p.GXn;
		const int ny0 = gy * 4 * WG_GH, nx0 = gx * 4 * WG_GW, nY0 = ny0 - p.pad_y, nX0 = nx0 - p.pad_x;
		rs_a = wf_make_rsrc(p.a + (long)n * p.a_sn + (long)nY0 * p.a_sh + (long)nX0 * p.a_sw, 0x7fff0000u);
		rs_g = wf_make_rsrc(p.g + (long)n * p.g_sn + (long)ny0 * p.g_sh + (long)nx0 * p.g_sw, 0x7fff0000u);
		aY0 = nY0 < 0 ? -nY0 : 0; aY1 = p.H - nY0 < WG_RH ? p.H - nY0 : WG_RH;
		aX0 = nX0 < 0 ? -nX0 : 0; aX1 = p.W - nX0 < WG_RW ? p.W - nX0 : WG_RW;
		gY1 = p.OH - ny0 < 4 * WG_GH ? p.OH - ny0 : 4 * WG_GH;
		gX1 = p.OW - nx0 < 4 * WG_GW ? p.OW - nx0 : 4 * WG_GW;
	};
	// piece i (0 .. 11: input, 12 .. 15: gradient) of the group set by set_group, into stage `st`
	auto dma_piece = [&](auto ic, const int st, const bool live) {
		constexpr int i = decltype(ic)::value;
		if constexpr (i < WG_A_PER_WAVE) {
			const int Y = ayx[i] >> 8, X = ayx[i] & 255; // (an empty slot holds Y = 255: never in range)
			const bool ok = live & (Y >= aY0) & (Y < aY1) & (X >= aX0) & (X < aX1);
			const unsigned voff = ok ? (unsigned)(Y * a_sh4 + X * a_sw4 + a_coff) : WF_OOB;
			const unsigned dst = wave + 4 * i < WG_A_PIECES ? lds0 + st * WG_STAGE_BYTES + (wave + 4 * i) * 1024 : lds0 + 2 * WG_STAGE_BYTES; // (scalar select: no branch in the MFMA stream)
			wf_dma16(rs_a, lds, dst, voff, 0u);
		} else {
			constexpr int j = i - WG_A_PER_WAVE;
			const int y = gyx[j] >> 8, x = gyx[j] & 255;
			const bool ok = live & (y < gY1) & (x < gX1);
			const unsigned voff = ok ? (unsigned)(y * g_sh4 + x * g_sw4 + g_coff) : WF_OOB;
			wf_dma16(rs_g, lds, lds0 + st * WG_STAGE_BYTES + WG_A_BYTES + (wave + 4 * j) * 1024, voff, 0u);
		}
	};

	if (g_first < g_end) { // (an empty slice still writes its zeros below: the fold reads every slice)
	set_group(g_first);
	NNC_ASM_NOPS("s_nop 4"); // descriptor words fresh from v_readfirstlane -> the first buffer_load reading them
	wf_static_for<WG_PIECES_PER_WAVE>([&](auto nc) {
		constexpr int n = decltype(nc)::value, piece = n < 7 ? n : (n < 11 ? WG_A_PER_WAVE + (n - 7) : n - 4); // the trips' issue order
		dma_piece(GroupId<piece>(), 0, true);
	});

	for (int grp = g_first; grp < g_end; grp++) {
		const int st = (grp - g_first) & 1;
		// (DBG, tools/wgrad_probe.cpp only: 1 no DMA in the loop, 2 no LDS reads / transforms, 16 no MFMAs, 32 no wait + barrier -- timing only)
		// Issue order of a trip's 16 pieces per wave (during the PREVIOUS trip, one behind every ninth MFMA -- the CU retires a piece per ~65 cycles,
		// a denser stream queues and blocks the in-order wave): first what k-step 0 needs -- input pieces 0 .. 6 (region rows 0 .. 5; + row 6 for wave 3)
		// and the four gradient pieces (both k-steps' W are made at the top) -- then input pieces 7 .. 11 (rows 6 .. 9, k-step 1 only).  So the wait at
		// the top leaves the last five in flight, and a second wait + barrier sits in front of k-step 1 (behind the eight pieces issued meanwhile).
		if constexpr (!(DBG & 32)) {
			WF_WAIT_VMCNT(5);                 // this wave's early pieces of the trip have landed ...
			__builtin_amdgcn_s_barrier();     // ... and so have everybody's; every wave is done reading the other stage
		}
		const bool has_next = grp + 1 < g_end;
		if (has_next) set_group(grp + 1);
		// lane (ch, slot): input channels 2 ch, 2 ch + 1 of the wave's 32 (the two MFMA column fragments: one 8-byte LDS read serves both, and the
		// transform runs on the pair -- packed fp32 where hipcc finds it), output channel ch of the wave's 16, tile (row = k-step, column = slot)
		const float* const ab = lds + st * (WG_STAGE_BYTES / 4) + (4 * slot) * WG_CB + wc * 32 + 2 * ch;               // + (Y * 18 + X) * 64
		const float* const gb = lds + st * (WG_STAGE_BYTES / 4) + WG_A_BYTES / 4 + (4 * slot) * WG_KB + wk * 16 + ch;  // + (y * 16 + x) * 32
		// W = G' e G'^T of this lane's (tile, k), BOTH k-steps at once (x: tile row 0, y: tile row 1)
		f2v W[36];
		if constexpr (DBG & 2) {
#pragma unroll
			for (int z = 0; z < 36; z++) W[z] = f2v{ 1.f + z, 2.f };
		} else {
			f2v e[4][4];
#pragma unroll
			for (int i = 0; i < 4; i++)
#pragma unroll
				for (int j = 0; j < 4; j++) e[i][j] = f2v{ gb[(i * (4 * WG_GW) + j) * WG_KB], gb[((4 + i) * (4 * WG_GW) + j) * WG_KB] };
			if (wc == 0 && cb == 0) {
#pragma unroll
				for (int i = 0; i < 4; i++) { const f2v r4 = (e[i][0] + e[i][1]) + (e[i][2] + e[i][3]); bsum += r4.x + r4.y; }
			}
			f2v t1[6][4];
#pragma unroll
			for (int j = 0; j < 4; j++) {
				const f2v col[4] = { e[0][j], e[1][j], e[2][j], e[3][j] };
				f2v y[6];
				wino_g4(col, y);
#pragma unroll
				for (int rr = 0; rr < 6; rr++) t1[rr][j] = y[rr];
			}
#pragma unroll
			for (int rr = 0; rr < 6; rr++) {
				f2v y[6];
				wino_g4(t1[rr], y);
#pragma unroll
				for (int q = 0; q < 6; q++) W[rr * 6 + q] = y[q];
			}
		}
#pragma unroll
		for (int z = 0; z < 36; z++) { NNC_PIN_V(W[z].x); NNC_PIN_V(W[z].y); }
		wf_static_for<2>([&](auto ksc) {
			constexpr int ks = decltype(ksc)::value; // tiles (ty = ks, tx = slot)
			if constexpr (ks == 1 && !(DBG & 32)) {
				WF_WAIT_VMCNT(8);             // the late five of THIS trip's pieces (rows 6 .. 9), behind the eight issued for the next trip so far
				__builtin_amdgcn_s_barrier();
			}
			// V = B^T d B of this lane's (tile, c = 2 ch, 2 ch + 1)
			f2v V[36];
			if constexpr (DBG & 2) {
#pragma unroll
				for (int z = 0; z < 36; z++) V[z] = f2v{ 3.f, 1.f + z };
			} else {
				f2v sm[6][6];
#pragma unroll
				for (int c6 = 0; c6 < 6; c6++) {
					f2v col[6], y[6];
#pragma unroll
					for (int r6 = 0; r6 < 6; r6++) col[r6] = *(const f2v*)(ab + ((4 * ks + r6) * WG_RW + c6) * WG_CB);
					wino_bt(col, y);
#pragma unroll
					for (int r6 = 0; r6 < 6; r6++) sm[r6][c6] = y[r6];
				}
#pragma unroll
				for (int r6 = 0; r6 < 6; r6++) {
					f2v y[6];
					wino_bt(sm[r6], y);
#pragma unroll
					for (int q = 0; q < 6; q++) V[r6 * 6 + q] = y[q];
				}
			}
#pragma unroll
			for (int z = 0; z < 36; z++) { NNC_PIN_V(V[z].x); NNC_PIN_V(V[z].y); }
			NNC_ASM_NOPS("s_nop 1"); // the last transform VALU -> an MFMA reading its result (asm MFMAs are outside hipcc's hazard handling)
			wf_static_for<72>([&](auto mc) {
				constexpr int mm = decltype(mc)::value, z = mm >> 1, jf = mm & 1; // an accumulator recurs every 72 MFMAs; W[z] feeds two in a row
				if constexpr (!(DBG & 16)) {
					if constexpr (ks == 0 && jf == 0) WF_MFMA(acc[z][0], W[z].x, V[z].x, z < WF_Z_AGPR);
					else if constexpr (ks == 0) WF_MFMA(acc[z][1], W[z].x, V[z].y, z < WF_Z_AGPR);
					else if constexpr (jf == 0) WF_MFMA(acc[z][0], W[z].y, V[z].x, z < WF_Z_AGPR);
					else WF_MFMA(acc[z][1], W[z].y, V[z].y, z < WF_Z_AGPR);
				}
				// the next trip's pieces, one behind every ninth MFMA, early ones first (see the top of the trip)
				// (the last trip of a slice issues them too, every lane out of range: zeros into a stage nobody reads -- no branch in the MFMA stream)
				constexpr int m = ks * 72 + mm;
				if constexpr (m % 9 == 0 && !(DBG & 1)) {
					constexpr int n = m / 9; // 0 .. 15
					constexpr int piece = n < 7 ? n : (n < 11 ? WG_A_PER_WAVE + (n - 7) : n - 4);
					dma_piece(GroupId<piece>(), st ^ 1, has_next);
				}
			});
		});
	}
	}

	NNC_ASM_NOPS("s_nop 15\n\ts_nop 15"); // the last MFMAs' results -> the compiler-visible reads below
	{
		// D layout of 16x16x4: lane holds rows 4 (lane >> 4) + i (k), column lane & 15 (c)
		float* const out = p.partial + (long)slice * 36 * p.K * p.C + (long)(kb * WG_KB + wk * 16 + 4 * slot) * p.C + cb * WG_CB + wc * 32 + 2 * ch; // column fragment j = input channel 2 ch + j
		const long plane = (long)p.K * p.C;
#pragma unroll
		for (int z = 0; z < 36; z++)
#pragma unroll
			for (int j = 0; j < 2; j++)
#pragma unroll
				for (int i = 0; i < 4; i++) out[z * plane + (long)i * p.C + j] = acc[z][j][i];
		if (p.bias_partial && wc == 0 && cb == 0) p.bias_partial[((long)slice * 4 + slot) * p.K + kb * WG_KB + wk * 16 + ch] = bsum;
	}
}

// dU[z][k][c] = sum over slices of partial[s][z][k][c], slices in order (deterministic): one thread per (z, k, c) element -- 36 K C threads walking
// `slices` coalesced rows.  (The first version folded inside the final transform, one thread per (k, c): 16 workgroups walking 128 slices x 36
// strided loads took 1.3 ms on conv1_2, a third of the contraction itself.)  dbias[k] (+)= the same fold of the bias partials, by the first K threads.
static __global__ void __launch_bounds__(256) wino_wgrad_fused_fold_kernel(const float* __restrict__ partial, const float* __restrict__ bias_partial, float* __restrict__ du, float* __restrict__ dbias, const long n, const int K, const int slices, const int accumulate)
{
	const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (idx < n) {
		float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f; // four loads in flight; the order of the additions is fixed
		int s = 0;
		for (; s + 4 <= slices; s += 4) {
			s0 += partial[(long)s * n + idx]; s1 += partial[(long)(s + 1) * n + idx];
			s2 += partial[(long)(s + 2) * n + idx]; s3 += partial[(long)(s + 3) * n + idx];
		}
		for (; s < slices; s++) s0 += partial[(long)s * n + idx];
		du[idx] = (s0 + s1) + (s2 + s3);
	}
	if (dbias && bias_partial && idx < K) {
		float s = 0.f;
		for (int i = 0; i < slices * 4; i++) s += bias_partial[(long)i * K + idx];
		dbias[idx] = accumulate ? dbias[idx] + s : s;
	}
}

} // namespace nnc
