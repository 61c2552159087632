// EXPERIMENT (round 3; tools/wf2_probe.cpp -- not part of the library).  Measured on the MI355X (profiles/r03_v7_issue_probes.txt): NOT faster than the one-wave
// kernel (conv1_2 3.59 vs 3.24 ms, conv2_2 2.88 vs 2.81) -- a VALU instruction costs matrix-pipe time whichever wave of the SIMD issues it (two waves, MFMA + one
// packed VALU each: 45 clocks per MFMA, as on one wave), so only the DMA issue and the LDS waits hide behind the sibling, and the pair's epilogue (12 workgroup
// barriers, partial sums through LDS) costs more than it saves.  One GPU parity case of the emulator-green kernel still differs (a hazard the synchronous
// emulator cannot show); it was not chased once the timing was known.
// Fused Winograd F(4x4, 3x3), two waves per SIMD: the same decomposition, LDS layout, U fragments and work distribution as wino_fused.h, with each
// 16-tile x 32-channel block split between a PAIR of waves by transform-domain column -- wave `role` of the pair owns the 18 positions
// z = zy * 6 + (3 role + zxl) (144 accumulator registers, all in AGPRs), so eight waves of <= 256 registers fit a CU and the SIMD always has a
// second wave to issue from.  What that buys (tools/coissue2_probe.cpp, tools/occ2_probe.cpp on the MI355X): the MFMAs of one wave run at the pipe's
// rate (32 clocks) whatever the sibling wave issues -- VALU, ds_read, LDS-DMA pieces all hide behind them -- whereas in ONE in-order wave every
// transform VALU (4 clocks), every DMA piece (the issue blocks while the texture path is busy: 44-90 clocks) and the whole epilogue add to the MFMA time
// (wino_fused.h: 0.47 of the fp32 MFMA peak).
//   * a pair shares its patch buffers (one DMA, both waves read); role 0 needs columns 0..2 of S = d B (outputs y0..y2 of the six-point transform:
//     reads x0..x4), role 1 columns 3..5 (y3..y5: reads x1..x5) -- six packed operations per row each, the same VALU per MFMA as the one-wave kernel;
//   * per trip a wave issues 72 MFMAs, 72 packed transform operations (one behind every MFMA), 18 U reads, 30 patch reads and 10 of the workgroup's
//     80 DMA pieces (role 0: patch pieces 0..4 + five U pieces, role 1: patch pieces 5..10 + four), all in the first ten iterations; one
//     s_waitcnt vmcnt(0) + workgroup barrier per trip makes every piece of the previous trip visible to every wave;
//   * epilogue: A^T M A is linear in the columns of M, so each wave transforms its three columns into a PARTIAL 4x4 output (role 0:
//     [w0 + s, d, s, d], s = w1 + w2, d = w1 - w2; role 1: [s, 2 d, 4 s, 8 d + w5], s = w3 + w4, d = w3 - w4) and the two partial sums meet in the LDS staging
//     area the one-wave kernel already stores through: per round each wave writes one channel half, adds the other on top of the sibling's, and
//     stores half of the pixels (16-byte lanes, bias / ReLU / mask bits there).
// Spec of the arithmetic: lib/nnc/cmd/convolution/cpu_opt/_ccv_nnc_conv_cpu_4x4_3x3_winograd.c:126- (same matrices as winograd.h).
#pragma once
#include "wino_fused.h"

namespace nnc {

typedef f2 f2v;
// hipcc gives a kernel of two waves per SIMD 128 VGPRs + 128 AGPRs (an even split of the 256; the "amdgpu-agpr-alloc" function attribute that would move the
// border is not reachable from HIP source): the accumulators of the first 16 iterations (128 registers) live in AGPRs, the last two iterations' (16) in VGPRs
constexpr int WF2_IT_AGPR = 16;
#ifdef NNC_HIP_EMULATOR
#define WF2_PIN(v) ((void)0)
#else
#define WF2_PIN(v) asm volatile("" : "+v"(v))
#endif

// The six-point transform y = B^T x on a channel pair, one packed operation (K = 0..11) at a time:
//   a = x4 - 4 x2, b = x3 - 4 x1, c = x4 - x2, t = x3 - x1, m = x4 - 5 x2, n = x5 - 5 x3
//   y0 = 4 x0 + m, y1 = a + b, y2 = a - b, y3 = c + 2 t, y4 = c - 2 t, y5 = 4 x1 + n
template <int K>
__device__ __forceinline__ void wf2_bt_op(const f2v& x0, const f2v& x1, const f2v& x2, const f2v& x3, const f2v& x4, const f2v& x5, f2v& y0, f2v& y1, f2v& y2, f2v& y3, f2v& y4, f2v& y5, f2v (&T)[2])
{ // (ordered so that two temporaries are live at a time)
	if constexpr (K == 0) { T[0] = x4 - 4.f * x2; WF2_PIN(T[0]); }
	else if constexpr (K == 1) { T[1] = x3 - 4.f * x1; WF2_PIN(T[1]); }
	else if constexpr (K == 2) { y1 = T[0] + T[1]; WF2_PIN(y1); }
	else if constexpr (K == 3) { y2 = T[0] - T[1]; WF2_PIN(y2); }
	else if constexpr (K == 4) { T[0] = x4 - x2; WF2_PIN(T[0]); }
	else if constexpr (K == 5) { T[1] = x3 - x1; WF2_PIN(T[1]); }
	else if constexpr (K == 6) { y3 = T[0] + 2.f * T[1]; WF2_PIN(y3); }
	else if constexpr (K == 7) { y4 = T[0] - 2.f * T[1]; WF2_PIN(y4); }
	else if constexpr (K == 8) { T[0] = x4 - 5.f * x2; WF2_PIN(T[0]); }
	else if constexpr (K == 9) { y0 = 4.f * x0 + T[0]; WF2_PIN(y0); }
	else if constexpr (K == 10) { T[1] = x5 - 5.f * x3; WF2_PIN(T[1]); }
	else { y5 = 4.f * x1 + T[1]; WF2_PIN(y5); }
}
// Half of it: ROLE 0 -> (o0, o1, o2) = (y0, y1, y2) from d[0..4] = x0..x4; ROLE 1 -> (y3, y4, y5) from d[0..4] = x1..x5.  K = 0..5; the inputs are dead
// after K = 3 (the next row's reads may overwrite them).
template <int ROLE, int K>
__device__ __forceinline__ void wf2_bt_half_op(const f2v (&d)[5], f2v& o0, f2v& o1, f2v& o2, f2v (&T)[2])
{
	if constexpr (ROLE == 0) { // d = x0 x1 x2 x3 x4; o0 doubles as the third temporary
		if constexpr (K == 0) { o0 = d[4] - 5.f * d[2]; WF2_PIN(o0); }
		else if constexpr (K == 1) { T[0] = d[4] - 4.f * d[2]; WF2_PIN(T[0]); }
		else if constexpr (K == 2) { T[1] = d[3] - 4.f * d[1]; WF2_PIN(T[1]); }
		else if constexpr (K == 3) { o0 = 4.f * d[0] + o0; WF2_PIN(o0); }
		else if constexpr (K == 4) { o1 = T[0] + T[1]; WF2_PIN(o1); }
		else { o2 = T[0] - T[1]; WF2_PIN(o2); }
	} else { // d = x1 x2 x3 x4 x5; o2 doubles as the third temporary
		if constexpr (K == 0) { T[0] = d[3] - d[1]; WF2_PIN(T[0]); }
		else if constexpr (K == 1) { T[1] = d[2] - d[0]; WF2_PIN(T[1]); }
		else if constexpr (K == 2) { o2 = d[4] - 5.f * d[2]; WF2_PIN(o2); }
		else if constexpr (K == 3) { o2 = 4.f * d[0] + o2; WF2_PIN(o2); }
		else if constexpr (K == 4) { o0 = T[0] + 2.f * T[1]; WF2_PIN(o0); }
		else { o1 = T[0] - 2.f * T[1]; WF2_PIN(o1); }
	}
}

// DBG (tools/wf2_probe.cpp only): 1 no DMA in the loop, 2 no patch reads, 4 no U reads, 8 no transform VALU, 16 no MFMAs, 32 no barrier / wait in the loop, 64 no epilogue
template <int GH, int GW, int DBG = 0, bool MASK = false>
__global__ void __launch_bounds__(512) wino_fused2_kernel(const WinoFusedArgs a)
{
	typedef WfGeom<GH, GW> G;
	constexpr int GWL = GW == 4 ? 2 : (GW == 8 ? 3 : (GW == 2 ? 1 : 0));
	static_assert(GWL >= 1 && GWL <= 3, "tile groups 8 x 2, 4 x 4, 2 x 8");
	constexpr int FSH = 3 - GWL; // which plane-row bit swaps a pixel's channel halves (wino_fused.h, patch reads)
	__shared__ __attribute__((aligned(16))) float lds[2 * WF_U_FLOATS + 8 * WF_P_FLOATS]; // 160 KB: [U ring x2][patch x2 per pair]
	const int t = threadIdx.x;
	const int lane = t & 63;
	const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
	const int pr = wave & 3;      // the pair = tile group of the quad; waves pr and pr + 4 land on the same SIMD
	const int role = wave >> 2;
	// work items: exactly wino_fused_kernel's (teams of workgroups on one XCD walking a range of tile-group quads, member m the k blocks m, m + team, ...)
	const int nquads = (a.groups + 3) / 4;
	const int team = a.team, kpm = a.KB / team;
	const int slot = (int)blockIdx.x >> 3, teams = ((int)gridDim.x >> 3) / team * 8;
	const int member = slot % team, team_id = ((int)blockIdx.x & 7) * (teams >> 3) + slot / team;
	const int qper = (nquads + teams - 1) / teams;
	const int q_first = team_id * qper;
	const int q_count = nquads - q_first < qper ? nquads - q_first : qper;
	const int count = q_count * kpm;
	if (count <= 0) return;
	const int gpi = a.GYn * a.GXn;
	const int ti = lane & 15, g = lane >> 4;
	const int ty = ti >> GWL, tx = ti & (GW - 1);

	float* const ubuf = lds;
	float* const pbuf = lds + 2 * WF_U_FLOATS + pr * 2 * WF_P_FLOATS;
	const unsigned lds0 = __builtin_amdgcn_readfirstlane(wf_lds_addr(lds));
	const unsigned p_lds = lds0 + (2 * WF_U_FLOATS + pr * 2 * WF_P_FLOATS) * 4;

	struct Item { int kb, n, gy, gx, live; };
	auto item_of = [&](const int idx) -> Item {
		Item r;
		const int gq = q_first + idx / kpm;
		r.kb = member + (idx % kpm) * team;
		int group = gq * 4 + pr;
		r.live = group < a.groups;
		if (!r.live) group = a.groups - 1; // computes a duplicate, stores nothing
		r.n = group / gpi;
		const int gr = group - r.n * gpi;
		r.gy = gr / a.GXn;
		r.gx = gr - r.gy * a.GXn;
		return r;
	};

	auto body = [&](auto rolec) {
		constexpr int ROLE = decltype(rolec)::value;
		constexpr int NP = ROLE == 0 ? 5 : 6, Q0 = ROLE == 0 ? 0 : 5; // this wave's patch pieces Q0 .. Q0 + NP - 1
		constexpr int NU = ROLE == 0 ? 5 : 4;                          // and its U pieces u_first .. u_first + NU - 1
		const unsigned u_first = ROLE == 0 ? (unsigned)pr * 5u : 20u + (unsigned)pr * 4u;
		wf_rsrc_t rs_src, rs_u;
		unsigned pvoff[NP];
		unsigned pyx2[(NP + 1) / 2]; // per piece 16 bits: X | Y << 6 | channel half << 12, 0xffff = no pixel; two pieces per register
#pragma unroll
		for (int q = 0; q < (NP + 1) / 2; q++) pyx2[q] = 0;
#pragma unroll
		for (int q = 0; q < NP; q++) {
			const int s = (Q0 + q) * 64 + lane;
			const int slot_ = s >> 1;
			const unsigned yx = wf_slot_tab<GH, GW>.yx[slot_];
			const unsigned hh = (unsigned)(s & 1) ^ ((yx >> (10 + FSH)) & 1u);
			const unsigned code = yx == 0xffffu ? 0xffffu : ((yx & 63u) | ((yx >> 8) & 63u) << 6 | hh << 12);
			pyx2[q >> 1] |= code << (16 * (q & 1));
		}
		const int sh4 = (int)a.s_sh * 4, sw4 = (int)a.s_sw * 4;
		auto set_patch = [&](const Item& it) {
			rs_src = wf_make_rsrc(a.src + (long)it.n * a.s_sn, a.src_image_bytes);
			const int Y0 = it.gy * GH * 4 - a.pad_y, X0 = it.gx * GW * 4 - a.pad_x;
#pragma unroll
			for (int q = 0; q < NP; q++) {
				const unsigned code = (pyx2[q >> 1] >> (16 * (q & 1))) & 0xffffu;
				const int Y = Y0 + (int)((code >> 6) & 63u), X = X0 + (int)(code & 63u);
				const bool ok = (code != 0xffffu) & (Y >= 0) & (Y < a.H) & (X >= 0) & (X < a.W);
				pvoff[q] = ok ? (unsigned)(Y * sh4 + X * sw4 + (int)((code >> 12) & 1u) * 16) : WF_OOB;
			}
		};
		auto set_u = [&](const Item& it) { rs_u = wf_make_rsrc(a.uf + (long)it.kb * (a.uf_kb_bytes / 4), a.uf_kb_bytes); };
		const unsigned uvoff = (unsigned)lane * 16u;
		// piece s = 0..9 of a trip: patch and U pieces alternating while both last.  p_dst: LDS address of the patch buffer being filled, sp: its chunk's
		// byte offset inside a pixel's channels; u_dst: LDS address of the U buffer being filled, su: the chunk's byte offset inside the k block's fragments
		auto dma_piece = [&](auto sc, const unsigned p_dst, const unsigned sp, const unsigned u_dst, const unsigned su) {
			constexpr int s = decltype(sc)::value;
			constexpr bool is_p = (s < 2 * NU) ? (s % 2 == 0) : true; // NU <= NP: the patch pieces outlast the U pieces
			constexpr int idx = (s < 2 * NU) ? s / 2 : s - NU;
			if constexpr (is_p) wf_dma16(rs_src, lds, p_dst + (unsigned)(Q0 + idx) * 1024u, pvoff[idx], sp);
			else wf_dma16(rs_u, lds, u_dst + (u_first + idx) * 1024u, uvoff, su + (u_first + idx) * 1024u);
		};

		floatx4 acc[18][2]; // [zxl * 6 + zy][column tile]
#pragma unroll
		for (int z = 0; z < 18; z++)
#pragma unroll
			for (int j = 0; j < 2; j++) acc[z][j] = floatx4{ 0.f, 0.f, 0.f, 0.f };

		const int h = g >> 1;
		const int pb0 = (h ^ ((ty >> FSH) & 1)) * 4 + 2 * (g & 1), pb1 = (h ^ (((ty + 1) >> FSH) & 1)) * 4 + 2 * (g & 1);
		const int b5[2] = { (ty * (GW + 1) + tx) * 8 + pb0, (ty * (GW + 1) + tx) * 8 + pb1 }, b4[2] = { (ty * GW + tx) * 8 + pb0, (ty * GW + tx) * 8 + pb1 };
		auto patch_read = [&](const float* const pb, const int r, const int c) -> f2v {
			const int off = (G::plane_off(r & 3, c & 3) + (r >> 2) * G::cx(c & 3) + (c >> 2)) * 8 + ((c & 3) < 2 ? b5[r >> 2] : b4[r >> 2]);
			return *(const f2v*)(pb + off);
		};

		Item cur = item_of(0);
		set_patch(cur);
		set_u(cur);
		f2v S[6][3], Vc[2][6], T[2], d[5]; // S[row][own column] = d B of the chunk being multiplied; Vc[column & 1][zy] = column of V = B^T S
#ifndef NNC_HIP_EMULATOR
		asm volatile("s_nop 4");
#endif
		// prologue: chunk 0's patch + U, chunk 1's patch
		wf_static_for<10>([&](auto sc) { dma_piece(sc, p_lds, 0u, lds0, 0u); });
		wf_static_for<NP>([&](auto qc) { constexpr int q = decltype(qc)::value; wf_dma16(rs_src, lds, p_lds + WF_P_FLOATS * 4 + (unsigned)(Q0 + q) * 1024u, pvoff[q], WF_CC * 4); });
		WF_WAIT_VMCNT(0);
		__builtin_amdgcn_s_barrier();
		{
#pragma unroll
			for (int r = 0; r < 6; r++) {
#pragma unroll
				for (int c = 0; c < 5; c++) d[c] = patch_read(pbuf, r, c + ROLE);
				wf_static_for<6>([&](auto kc) { wf2_bt_half_op<ROLE, decltype(kc)::value>(d, S[r][0], S[r][1], S[r][2], T); });
			}
			wf_static_for<12>([&](auto kc) { wf2_bt_op<decltype(kc)::value>(S[0][0], S[1][0], S[2][0], S[3][0], S[4][0], S[5][0], Vc[0][0], Vc[0][1], Vc[0][2], Vc[0][3], Vc[0][4], Vc[0][5], T); });
		}

		// ---- one trip: multiplies the current chunk (U buffer `par`, S in registers) while it fetches U of the next chunk (into U buffer par ^ 1) and the patch
		// of the one after (into patch buffer par), and transforms the next chunk's patch (buffer par ^ 1) into S.  18 iterations of 4 MFMAs, column by column
		// (iteration it: zxl = it / 6, zy = it % 6); behind every MFMA one packed transform operation:
		//   it 0 .. 2    column 1 of V from S -> Vc[1]          it 6 .. 8    column 2 -> Vc[0] (column 0's last MFMA was iteration 5's)
		//   it 9 .. 17   S'[r] = (d'[r] B)[own columns] of the next chunk, row by row (S is dead after iteration 8); the row's five patch reads go out behind the
		//                fourth operation of the row before (its last use of d)
		//   tail         column 0 of the next chunk's V -> Vc[0] (in use until iteration 17's MFMAs have issued)
		auto trip = [&](const int par, const unsigned sp, const unsigned su) {
			const float* const ub = ubuf + par * WF_U_FLOATS + lane * 4;
			const float* const pbn = pbuf + (par ^ 1) * WF_P_FLOATS;
			const unsigned p_dst = p_lds + par * (WF_P_FLOATS * 4), u_dst = lds0 + (par ^ 1) * (WF_U_FLOATS * 4);
			float4 u[2];
			if constexpr (DBG & 2) for (int i = 0; i < 5; i++) d[i] = f2v{ 1.f, 2.f };
			if constexpr (DBG & 4) u[0] = u[1] = make_float4(1.f, 2.f, 3.f, 4.f);
			else u[0] = *(const float4*)(ub + (ROLE * 3) * 256);
			wf_static_for<18>([&](auto itc) {
				constexpr int it = decltype(itc)::value;
				constexpr int zxl = it / 6, zy = it % 6;
				constexpr int vb = zxl == 1 ? 1 : 0; // column 0 -> Vc[0], 1 -> Vc[1], 2 -> Vc[0]
				wf_static_for<4>([&](auto kc) {
					constexpr int k = decltype(kc)::value;
					// u = { (j0, e0), (j0, e1), (j1, e0), (j1, e1) }; MFMA order j0 e0, j1 e0, j0 e1, j1 e1
					if constexpr (DBG & 16) { NNC_PIN_V(acc[it][k & 1][0]); }
					else if constexpr (k == 0) WF_MFMA(acc[it][0], Vc[vb][zy].x, u[it & 1].x, it < WF2_IT_AGPR);
					else if constexpr (k == 1) WF_MFMA(acc[it][1], Vc[vb][zy].x, u[it & 1].z, it < WF2_IT_AGPR);
					else if constexpr (k == 2) WF_MFMA(acc[it][0], Vc[vb][zy].y, u[it & 1].y, it < WF2_IT_AGPR);
					else WF_MFMA(acc[it][1], Vc[vb][zy].y, u[it & 1].w, it < WF2_IT_AGPR);
					constexpr int sl = it * 4 + k; // transform slot
					if constexpr (!(DBG & 8)) {
						if constexpr (sl < 12) wf2_bt_op<sl>(S[0][1], S[1][1], S[2][1], S[3][1], S[4][1], S[5][1], Vc[1][0], Vc[1][1], Vc[1][2], Vc[1][3], Vc[1][4], Vc[1][5], T);
						else if constexpr (sl >= 24 && sl < 36) wf2_bt_op<sl - 24>(S[0][2], S[1][2], S[2][2], S[3][2], S[4][2], S[5][2], Vc[0][0], Vc[0][1], Vc[0][2], Vc[0][3], Vc[0][4], Vc[0][5], T);
						else if constexpr (sl >= 36) {
							constexpr int r = (sl - 36) / 6, kk = (sl - 36) % 6;
							wf2_bt_half_op<ROLE, kk>(d, S[r][0], S[r][1], S[r][2], T);
						}
					}
					// LDS reads: the next iteration's U fragments behind the first MFMA; the patch rows as described
					if constexpr (k == 0 && it + 1 < 18 && !(DBG & 4)) {
						constexpr int itn = it + 1, zn = (itn % 6) * 6 + ROLE * 3 + itn / 6;
						u[itn & 1] = *(const float4*)(ub + zn * 256);
					}
					if constexpr (!(DBG & 2)) {
						if constexpr (sl == 30) { // row 0 (nothing of the previous trip's rows is live in d)
#pragma unroll
							for (int c = 0; c < 5; c++) d[c] = patch_read(pbn, 0, c + ROLE);
						} else if constexpr (sl >= 36 && (sl - 36) % 6 == 3 && (sl - 36) / 6 < 5) {
							constexpr int rn = (sl - 36) / 6 + 1;
#pragma unroll
							for (int c = 0; c < 5; c++) d[c] = patch_read(pbn, rn, c + ROLE);
						}
					}
					// DMA: this wave's ten pieces behind iterations 0 .. 9
					if constexpr (k == 3 && it < 10 && !(DBG & 1)) dma_piece(GroupId<it>(), p_dst, sp, u_dst, su);
					__builtin_amdgcn_sched_barrier(0);
				});
			});
			if constexpr (!(DBG & 8))
				wf_static_for<12>([&](auto kc) { wf2_bt_op<decltype(kc)::value>(S[0][0], S[1][0], S[2][0], S[3][0], S[4][0], S[5][0], Vc[0][0], Vc[0][1], Vc[0][2], Vc[0][3], Vc[0][4], Vc[0][5], T); });
		};

		// ---- the epilogue of an item (see the header): par = the U buffer the item's last trip has read, now the staging area (a quarter per pair)
		auto epilogue = [&](const Item& it, const int par) {
#ifndef NNC_HIP_EMULATOR
			asm volatile("s_nop 15\n\ts_nop 15"); // the last MFMAs' results -> the compiler-visible reads below
#endif
			if constexpr (DBG & 64) { if (a.bias == (const float*)1) a.dst[t] = acc[0][0][0] + acc[17][1][3] + acc[9][0][1]; return; }
			__builtin_amdgcn_s_barrier(); // every wave is done reading the U buffer the staging overwrites; no DMA is in flight (the trip ended with a count-zero wait)
			constexpr int TS = 16 * 32 + 16;
			float* const st = ubuf + par * WF_U_FLOATS + pr * 2304;
			static_assert(4 * TS <= 2304, "staging area of a pair: a quarter of a U buffer");
			typedef unsigned int u4 __attribute__((ext_vector_type(4)));
			const __amdgpu_buffer_rsrc_t rs_dst = __builtin_amdgcn_make_buffer_rsrc((void*)(a.dst + (long)it.n * a.d_sn), 0, a.dst_image_bytes, 0x00020000);
			const int kq = it.kb * WF_KT + (lane & 7) * 4;
			const bool kok = (it.live != 0) & (kq < a.K);
			float bv[4];
#pragma unroll
			for (int i = 0; i < 4; i++) bv[i] = (a.bias && kq < a.K) ? a.bias[kq + i] : 0.f;
			const int dh4 = (int)a.d_sh * 4, dw4 = (int)a.d_sw * 4;
			unsigned mb[4] = { ~0u, ~0u, ~0u, ~0u };
			if constexpr (MASK) {
				const unsigned* const mp = a.mask_bits + ((((long)it.n * a.GYn + it.gy) * a.GXn + it.gx) * a.KB + it.kb) * 256 + lane;
#pragma unroll
				for (int r = 0; r < 4; r++) mb[r] = it.live ? mp[r * 64] : 0u;
			}
			// the partial 4 x 4 output of (register r, column tile j): y[i][jj]
			auto partial = [&](const int r, const int j, float (&y)[4][4]) {
				float w[4][3];
#pragma unroll
				for (int zxl = 0; zxl < 3; zxl++) {
					const float col[6] = { acc[zxl * 6 + 0][j][r], acc[zxl * 6 + 1][j][r], acc[zxl * 6 + 2][j][r], acc[zxl * 6 + 3][j][r], acc[zxl * 6 + 4][j][r], acc[zxl * 6 + 5][j][r] };
					float yy[4];
					wino_at(col, yy);
#pragma unroll
					for (int i = 0; i < 4; i++) w[i][zxl] = yy[i];
				}
#pragma unroll
				for (int i = 0; i < 4; i++) {
					if constexpr (ROLE == 0) {
						const float s = w[i][1] + w[i][2], dd = w[i][1] - w[i][2];
						y[i][0] = w[i][0] + s; y[i][1] = dd; y[i][2] = s; y[i][3] = dd;
					} else {
						const float s = w[i][0] + w[i][1], dd = w[i][0] - w[i][1];
						y[i][0] = s; y[i][1] = 2.f * dd; y[i][2] = 4.f * s; y[i][3] = 8.f * dd + w[i][2];
					}
				}
			};
#pragma unroll
			for (int r = 0; r < 4; r++) {
				float y[4][4];
				// this wave's own channel half first (column tile j = ROLE): plain writes
				partial(r, ROLE, y);
#pragma unroll
				for (int i = 0; i < 4; i++)
#pragma unroll
					for (int jj = 0; jj < 4; jj++) st[g * TS + (i * 4 + jj) * 32 + ROLE * 16 + ti] = y[i][jj];
				partial(r, ROLE ^ 1, y);
				__builtin_amdgcn_s_barrier();
				// the other half on top of the sibling's
#pragma unroll
				for (int i = 0; i < 4; i++)
#pragma unroll
					for (int jj = 0; jj < 4; jj++) {
						float* const q = st + g * TS + (i * 4 + jj) * 32 + (ROLE ^ 1) * 16 + ti;
						*q = *q + y[i][jj];
					}
				__builtin_amdgcn_s_barrier();
				// read back and store: 4 tiles x 16 pixels x 8 channel quads = 512 float4; this wave the tile slots 2 ROLE, 2 ROLE + 1 (e = 4 ROLE .. 4 ROLE + 3 of the one-wave kernel's eight)
#pragma unroll
				for (int e4 = 0; e4 < 4; e4++) {
					const int e = ROLE * 4 + e4;
					const int pid = e * 8 + (lane >> 3);
					const int gp = pid >> 4, px = pid & 15;
					const int tile = 4 * gp + r;
					const int oy = (it.gy * GH + (tile >> GWL)) * 4 + (px >> 2), ox = (it.gx * GW + (tile & (GW - 1))) * 4 + (px & 3);
					const float4 v = *(const float4*)(st + gp * TS + px * 32 + (lane & 7) * 4);
					const unsigned voff = (kok & (oy < a.OH) & (ox < a.OW)) ? (unsigned)(oy * dh4 + ox * dw4 + kq * 4) : WF_OOB;
					float o0 = v.x + bv[0], o1 = v.y + bv[1], o2 = v.z + bv[2], o3 = v.w + bv[3];
					if (a.relu) { o0 = fmaxf(o0, 0.f); o1 = fmaxf(o1, 0.f); o2 = fmaxf(o2, 0.f); o3 = fmaxf(o3, 0.f); }
					if constexpr (MASK) {
						const unsigned m = mb[r] >> (4 * e);
						o0 = (m & 1) ? o0 : 0.f; o1 = (m & 2) ? o1 : 0.f; o2 = (m & 4) ? o2 : 0.f; o3 = (m & 8) ? o3 : 0.f;
					}
					__builtin_amdgcn_raw_buffer_store_b128(u4{ __float_as_uint(o0), __float_as_uint(o1), __float_as_uint(o2), __float_as_uint(o3) }, rs_dst, voff, 0, 0);
				}
				__builtin_amdgcn_s_barrier(); // the next round's writes (and the next trip's DMA) after this round's reads
			}
#pragma unroll
			for (int z = 0; z < 18; z++)
#pragma unroll
				for (int j = 0; j < 2; j++) acc[z][j] = floatx4{ 0.f, 0.f, 0.f, 0.f };
		};

		// ---- the stream
		int gtrip = 0;
		for (int ii = 0; ii < count; ii++) {
			const Item nxt = item_of(ii + 1 < count ? ii + 1 : ii); // (the last item fetches its own first chunks again: harmless)
			for (int cc = 0; cc < a.CCn; cc++, gtrip++) {
				const int par = gtrip & 1;
				// every piece of the previous trip has landed -- its U for this trip, its patch for this trip's transform, from all eight waves -- and every wave is
				// done with the buffers this trip's DMA overwrites (right after an epilogue both hold already)
				if constexpr (!(DBG & 32)) { if (cc != 0 || ii == 0) { WF_WAIT_VMCNT(0); __builtin_amdgcn_s_barrier(); } }
				if (cc == a.CCn - 2) set_patch(nxt);
				if (cc == a.CCn - 1) set_u(nxt);
				const int c2 = cc + 2 >= a.CCn ? cc + 2 - a.CCn : cc + 2, c1 = cc + 1 >= a.CCn ? 0 : cc + 1;
				trip(par, (unsigned)c2 * (WF_CC * 4), (unsigned)c1 * (WF_U_FLOATS * 4));
			}
			if constexpr (!(DBG & 32)) WF_WAIT_VMCNT(0);
			epilogue(cur, (gtrip - 1) & 1);
			cur = nxt;
		}
	};
	if (role == 0) body(GroupId<0>());
	else body(GroupId<1>());
}

} // namespace nnc
