// Fused Winograd F(4x4, 3x3) convolution for gfx950, fp32: input transform, the 36 contractions and the output transform in
// ONE kernel -- neither the transformed input V (2.25x the activations) nor the transformed output M ever exists in HBM.
// The via-HBM form (winograd.h) moves ~5.6x a convolution's algorithmic bytes for the 64/128-channel VGG-D layers
// (conv1_2 at batch 256: 36 GB for 6.6 GB of tensors) and its 64-channel contractions are HBM-bound at 60 TFLOP/s.
//
// What forces the shape of this kernel: the output transform needs all 36 transform-domain values of an (output tile,
// channel) pair at once, so a workgroup must hold 36 accumulators per output element: 36 x tiles x channels fp32 in
// registers.  With the CU's 128 K registers that is a 64 x 32 (tiles x channels) block at most, and the 160 KB LDS
// cannot hold V for it beyond a few channels.  So:
//   * MFMA v_mfma_f32_16x16x4_f32, one wave = 16 tiles x 32 channels x 36 positions = 72 accumulator tiles = 288 registers;
//     4 waves (one per SIMD) = 64 tiles x 32 channels per workgroup, up to 512 registers per lane.
//   * A operand (V) is never staged: lane (tile = l & 15, g = l >> 4) of the 16x16x4 A layout owns channels 2g, 2g+1 of an
//     8-channel chunk of ITS tile, computes B^T d B for them in registers and feeds the 36 x 2 values straight to the MFMAs
//     (the lane that transforms is the lane that supplies the fragment).
//   * the raw 6x6 patches arrive by LDS-DMA (buffer_load ... lds, no VGPR round trip): the (4 GH + 2) x (4 GW + 2) pixel
//     region of the wave's GH x GW tile group, 8 channels at a time, double buffered, private to the wave (no barrier);
//     out-of-image pixels are out-of-range buffer offsets and land as zeros.  The region is stored space-to-depth
//     (pixels with equal (y mod 4, x mod 4) adjacent) so the 16 tiles' reads of one patch position hit 16 distinct banks.
//   * B operand (U = G w G^T) is pre-arranged in HBM in FRAGMENT order -- [k block][c chunk][z][lane][j][e] -- by the weight
//     transform, so a chunk is one contiguous 36 KB piece: LDS-DMA copies it linearly (shared by the 4 waves, double
//     buffered, one barrier per chunk) and the four fragments of a position are ONE conflict-free linear ds_read_b128.
//   * the epilogue applies A^T M A + bias in registers (each lane holds all 36 z of its (tile, channel) outputs) and stores.
// Per 8-channel chunk a wave issues 144 MFMAs (4608 cycles) against 288 transform VALU, 108 LDS reads and 20 DMA issues.
// Spec of the arithmetic: lib/nnc/cmd/convolution/cpu_opt/_ccv_nnc_conv_cpu_4x4_3x3_winograd.c:126- (same matrices as winograd.h).
#pragma once
#include <utility>
#include "winograd.h"

namespace nnc {

constexpr int WF_KT = 32;                       // output channels per workgroup (2 MFMA column tiles per wave)
constexpr int WF_CC = 8;                        // reduction channels per chunk
constexpr int WF_U_FLOATS = 36 * 64 * 2 * 2;    // one chunk of U fragments: [z][lane][j][e], 36 KB
constexpr int WF_HP = 352;                      // pixel slots of a patch buffer (11 DMA pieces x 64 granules / 2)
constexpr int WF_P_FLOATS = 2 * WF_HP * 4;      // one patch buffer: [slot][8 channels], 11 KB -- a pixel's 32 bytes adjacent: lanes 2 i, 2 i + 1 of a piece fetch the two halves of ONE pixel
constexpr int WF_P_PIECES = 11;
constexpr int WF_Z_AGPR = 30;                     // positions whose two accumulator tiles live in AGPRs (60 tiles = 240 of the 256: hipcc needs slack there), the rest in VGPRs
constexpr unsigned WF_OOB = 0x7ffff000u;        // a buffer offset beyond every image: the DMA writes zeros

// compile-time loop: f(GroupId<0>()) ... f(GroupId<N - 1>()) -- every index a constant, whatever the unroller's budget says
template <class F, int... I> __device__ __forceinline__ void wf_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(GroupId<I>()), ...); }
template <int N, class F> __device__ __forceinline__ void wf_static_for(F&& f) { wf_static_for_impl(f, std::make_integer_sequence<int, N>()); }

// A channel pair in an aligned register pair: hipcc selects v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 for its arithmetic.  Next to the MFMAs a packed
// operation costs what a scalar one does (tools/coissue2_probe.cpp on the MI355X: one wave, MFMA + n independent VALU = 41.7 + 4.1 n clocks per MFMA for
// v_fma_f32 and v_pk_fma_f32 alike; the first VALU behind an MFMA costs ~14 clocks, hence the transforms in batches), so the transforms are half the VALU.
typedef float f2 __attribute__((ext_vector_type(2)));

// Space-to-depth slot of region pixel (Y, X) for a GH x GW tile group: planes by (Y & 3, X & 3), plane (yy, xx) holds
// (GH + (yy < 2)) x (GW + (xx < 2)) pixels row-major.  The 16 tiles of a wave read pixel (4 ty + r, 4 tx + c): same plane,
// slots ty * pitch + tx + const -- 16 distinct 16-byte granules (pitch = GW), or 16 with three pairs colliding (pitch = GW + 1).
template <int GH, int GW>
struct WfGeom {
	static constexpr int RH = 4 * GH + 2, RW = 4 * GW + 2, NPIX = RH * RW;
	static_assert(GH * GW == 16 && NPIX <= WF_HP, "tile group must be 16 tiles whose patch region fits the buffer");
	static constexpr __host__ __device__ int cy(int yy) { return GH + (yy < 2 ? 1 : 0); }
	static constexpr __host__ __device__ int cx(int xx) { return GW + (xx < 2 ? 1 : 0); }
	static constexpr __host__ __device__ int plane_off(int yy, int xx)
	{
		int o = 0;
		for (int p = 0; p < yy * 4 + xx; p++) o += cy(p >> 2) * cx(p & 3);
		return o;
	}
	static constexpr __host__ __device__ int slot(int Y, int X) { return plane_off(Y & 3, X & 3) + (Y >> 2) * cx(X & 3) + (X >> 2); }
};
// slot -> (Y << 8 | X), built at compile time; lives in the code object (no upload)
template <int GH, int GW>
struct WfSlotTab {
	unsigned short yx[WF_HP];
	constexpr WfSlotTab() : yx()
	{
		for (int i = 0; i < WF_HP; i++) yx[i] = 0xffff;
		for (int Y = 0; Y < WfGeom<GH, GW>::RH; Y++)
			for (int X = 0; X < WfGeom<GH, GW>::RW; X++) yx[WfGeom<GH, GW>::slot(Y, X)] = (unsigned short)(Y << 8 | X);
	}
};
template <int GH, int GW> __device__ __constant__ const WfSlotTab<GH, GW> wf_slot_tab = WfSlotTab<GH, GW>();

// U fragments: uf[(kb * CCn + cc) * 36 + z][lane = g * 16 + n][j][e] = (G w G^T)[z] for output channel
// k = kb * 32 + j * 16 + n (zero beyond K) and reduction channel c = cc * 8 + 2 g + e.   FLIP (dgrad): the roles of the
// weight tensor's two channel dimensions swap and the taps mirror, as in wino_weight_kernel.
// One thread per (k, c) of the PADDED k range.
template <bool FLIP>
static __global__ void __launch_bounds__(256) wino_weight_frag_kernel(const float* __restrict__ w, float* __restrict__ uf, const int Kout, const int Cred, const int Kw, const int Cw)
{
	// Kout / Cred: output and reduction channel counts of THIS convolution; w is [Kw][3][3][Cw] (forward: Kw = Kout, Cw = Cred; FLIP: Kw = Cred, Cw = Kout)
	const int KB = (Kout + WF_KT - 1) / WF_KT, CCn = Cred / WF_CC;
	const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (idx >= (long)KB * WF_KT * Cred) return;
	const int k = (int)(idx / Cred), c = (int)(idx - (long)k * Cred);
	float g[3][3];
#pragma unroll
	for (int i = 0; i < 3; i++)
#pragma unroll
		for (int j = 0; j < 3; j++)
			g[i][j] = k < Kout ? (FLIP ? w[((long)c * 9 + (2 - i) * 3 + (2 - j)) * Cw + k] : w[((long)k * 9 + i * 3 + j) * Cw + c]) : 0.f;
	(void)Kw;
	float t[6][3];
#pragma unroll
	for (int j = 0; j < 3; j++) {
		const float col[3] = { g[0][j], g[1][j], g[2][j] };
		float y[6];
		wino_g(col, y);
#pragma unroll
		for (int i = 0; i < 6; i++) t[i][j] = y[i];
	}
	const int kb = k / WF_KT, j16 = (k % WF_KT) / 16, n = k % 16;
	const int cc = c / WF_CC, gg = (c % WF_CC) / 2, e = c % 2;
	float* const dst = uf + (long)(kb * CCn + cc) * 36 * 256 + (gg * 16 + n) * 4 + j16 * 2 + e;
#pragma unroll
	for (int i = 0; i < 6; i++) {
		float y[6];
		wino_g(t[i], y);
#pragma unroll
		for (int jx = 0; jx < 6; jx++) dst[(long)(i * 6 + jx) * 256] = y[jx];
	}
}

struct WinoFusedArgs {
	const float* src;   // NHWC, channels dense
	float* dst;
	const float* uf;    // U fragments (wino_weight_frag_kernel)
	const float* bias;  // [K] or null
	long s_sn, s_sh, s_sw, d_sn, d_sh, d_sw; // element strides
	int H, W, OH, OW;   // source / destination extents
	int pad_y, pad_x;   // source padding (first tile's patch starts at -pad)
	int GYn, GXn;       // tile groups per image column / row
	int groups;         // N * GYn * GXn
	int C, K;           // reduction / output channels
	int CCn, KB;        // C / 8, ceil(K / 32)
	unsigned dst_image_bytes; // range of the per-image destination descriptor (pointer and strides are multiples of 16 bytes, K of 4: host-checked)
	int team;           // workgroups per team (divides KB; the grid is a multiple of 8 * team)
	unsigned src_image_bytes; // range of the per-image buffer descriptor
	unsigned uf_kb_bytes;     // bytes of one k block of U fragments (CCn * 36 KB)
	int relu;                 // epilogue writes max(0, A^T M A + bias) (NNC_MI355X_CONV_ALGO_FUSE_RELU)
	const unsigned* mask_bits; // or null.  Data gradient under a ReLU backward: one dword per (item, round, lane), bit 4 e + i set where the
	                          // element store e of that round writes at channel i is kept (wino_mask_pack_kernel makes them in the epilogue's own order)
};

// DMA piece n = 0..19 of a trip is issued behind iteration 9 n / 5 (0, 1, 3, 5, 7, 9, 10, ...): -1 = none behind iteration `it`
constexpr __host__ __device__ int wf_piece_at(int it)
{
	for (int n = 0; n < 20; n++)
		if (n * 9 / 5 == it) return n;
	return -1;
}

// ---- the PAIRED patch schedule (round 3; a NEGATIVE result, kept for tools/wf_probe.cpp -- the library does not instantiate it) -------------
// Hypothesis (round 2's reading of the knock-out probe): the loop is bound by L2 -> L1 line fills -- a chunk is 8 channels = a 32-byte run of a
// 128-byte line, and the next chunk's run of the same line is fetched a whole trip (~200 KB of other fills) later, so every patch line enters L1
// four times.  Measured on the MI355X (profiles/r03_v2_wf_probe_paired_schedule.txt): conv1_2 4.15 ms paired vs 3.37 classic, conv2_2 3.52 vs 2.85.
// Fetching the two halves of a 64-byte run back to back means 22 patch pieces in 11 consecutive iterations on all four waves at once, and what the
// loop is really bound by is the CU's vector-memory ISSUE rate: one 1 KB wave-instruction (DMA piece or 16-byte store alike) per ~60-80 cycles per
// CU whatever it touches; a queue of them blocks the in-order wave, MFMAs included.  The classic schedule's even pacing (one piece per 1.8
// iterations per wave) is what that limit wants; 80 pieces per 4608-cycle trip plus 128 stores per item leave this decomposition ~15 % above its
// vector-memory floor.  The schedule itself:  The 160 KB of LDS do not hold 16-channel chunks double-buffered,
// but they do not have to: the two 11 KB patch buffers of a wave become sub-chunk 0 / sub-chunk 1 of ONE 16-channel chunk, fetched TOGETHER --
// piece q of sub-chunk 0 (A_q) and piece q of sub-chunk 1 (B_q: the same pixels, soffset + 32 bytes) back to back, so the second finds the
// line the first brought in.  Single-buffered, which the dependences allow: a trip only reads the NEXT sub-chunk's buffer, late (iterations
// 23..32), so over a pair of trips (time t = 36 * half + iteration)
//   t = 34 .. 44          A_0 .. A_10 -> buffer 0 (free since the previous trip's transform), slot k = 1
//   t = 34 + SKEW ..      B_0 .. B_10 -> buffer 1 (free once the even trip has read it: last read t = 32), slot k = 3
//   t = 0, 3 .. 24        the even trip's nine U pieces (for the odd trip), slot k = 2;  t = 46, 48 .. 62 the odd trip's (for the next even trip)
// and the odd trip's transform (t = 59 on) waits for the A pieces with a counted wait.  Same pieces, same count, same LDS: only WHEN they fly.
// Piece code: -1 none, q patch A_q, 16 + q patch B_q, 32 + n U piece n.
template <int SKEW>
struct WfPair {
	static constexpr __host__ __device__ int at(int half, int it, int k)
	{
		const int t = half * 36 + it;
		if (k == 1) return (t >= 34 && t <= 44) ? t - 34 : -1;
		if (k == 3) return (t - SKEW >= 34 && t - SKEW <= 44) ? 16 + (t - SKEW - 34) : -1;
		if (k == 2) {
			if (half == 0) return (t % 3 == 0 && t / 3 < 9) ? 32 + t / 3 : -1;
			return (t >= 46 && (t - 46) % 2 == 0 && (t - 46) / 2 < 9) ? 32 + (t - 46) / 2 : -1;
		}
		return -1;
	}
	// pieces issued strictly after (t0, k0) and strictly before (t1, k1)
	static constexpr __host__ __device__ int issued_between(int t0, int k0, int t1, int k1)
	{
		int n = 0;
		for (int t = t0; t <= t1 && t < 72; t++)
			for (int k = 0; k < 4; k++) {
				if (t == t0 && k <= k0) continue;
				if (t == t1 && k >= k1) continue;
				if (at(t / 36, t % 36, k) >= 0) n++;
			}
		return n;
	}
	static constexpr int W_ODD_START = issued_between(24, 2, 36, 0);  // behind the even trip's last U piece: the patch pieces that may stay in flight
	static constexpr int W_XFORM = issued_between(44, 1, 59, 1);      // behind A_10, before the odd trip's first read of buffer 0 (iteration 23, slot 1)
	static_assert(SKEW >= 0 && 44 + SKEW < 59, "every B piece is issued before the odd trip's transform starts (and long before t = 72)");
};

// One third (PART) of the six-point transform y = B^T x on two channels at once, one f2 operation (K) at a time -- the kernel
// slots exactly one such operation (2 VALU) behind every MFMA.  T: a, b, c, t, m, n of
//   a = x4 - 4 x2, b = x3 - 4 x1, c = x4 - x2, t = x3 - x1, m = x4 - 5 x2, n = x5 - 5 x3
//   y0 = 4 x0 + m, y1 = a + b, y2 = a - b, y3 = c + 2 t, y4 = c - 2 t, y5 = 4 x1 + n          (= wino_bt)
// (each result is pinned where it is computed: hipcc otherwise sinks every transform to the end of the loop body, behind
// the last MFMA, whatever the sched_barrier fences say -- they bind the machine scheduler, not the IR passes before it)
#define WF_PIN2(v) NNC_PIN_VEC(v)
template <int PART, int K>
__device__ __forceinline__ void wf_bt_op(const f2& x0, const f2& x1, const f2& x2, const f2& x3, const f2& x4, const f2& x5, f2& y0, f2& y1, f2& y2, f2& y3, f2& y4, f2& y5, f2 (&T)[6])
{
	if (PART == 0) {
		if (K == 0) { T[0] = x4 - 4.f * x2; WF_PIN2(T[0]); }
		else if (K == 1) { T[1] = x3 - 4.f * x1; WF_PIN2(T[1]); }
		else if (K == 2) { T[2] = x4 - x2; WF_PIN2(T[2]); }
		else { T[3] = x3 - x1; WF_PIN2(T[3]); }
	} else if (PART == 1) {
		if (K == 0) { T[4] = x4 - 5.f * x2; WF_PIN2(T[4]); }
		else if (K == 1) { y0 = 4.f * x0 + T[4]; WF_PIN2(y0); }
		else if (K == 2) { y1 = T[0] + T[1]; WF_PIN2(y1); }
		else { y2 = T[0] - T[1]; WF_PIN2(y2); }
	} else {
		if (K == 0) { y3 = T[2] + 2.f * T[3]; WF_PIN2(y3); }
		else if (K == 1) { y4 = T[2] - 2.f * T[3]; WF_PIN2(y4); }
		else if (K == 2) { T[5] = x5 - 5.f * x3; WF_PIN2(T[5]); }
		else { y5 = 4.f * x1 + T[5]; WF_PIN2(y5); }
	}
}

// DBG (tools/wf_probe.cpp only; the library instantiates DBG = 0): knock parts of the loop out to attribute its time.
//   1 no DMA in the loop, 2 no patch reads, 4 no U fragment reads, 8 no transform VALU, 16 no MFMAs, 32 no barrier, 64 no epilogue,
//   256 no wait for the DMA (racy: timing only), 512 no patch DMA, 1024 no U DMA, 128 no stores in the epilogue (its arithmetic and staging stay),
//   2048 no output transform in the epilogue (the accumulators' z = 0 values are staged and stored instead)
//
// PERSISTENT: one workgroup per CU walks a contiguous range of work items (tile-group quad x k block, the k blocks of a quad
// consecutive: its patch lines stay in L1 / L2), and the chunk stream simply continues across items: the last two trips of an
// item already fetch and transform the first chunks of the next one, so only the very first item of a workgroup pays the DMA
// latency, the launch and the initial transform (measured before: 8-16 us per workgroup against 15 us of MFMAs for a
// 64-channel item).  Between two items sits the epilogue alone.
// MASK: the variant whose epilogue applies WinoFusedArgs::mask_bits (its own instantiation: the five registers it holds across the
// epilogue would otherwise spill in the plain kernel too).
// SCHED: 0 = the classic schedule (one chunk per trip, double-buffered patches); 1 + SKEW = the PAIRED schedule (WfPair<SKEW>; needs an even
// number of chunks, i.e. C % 16 == 0 -- the host picks).
template <int GH, int GW, int DBG = 0, bool MASK = false, int SCHED = 0>
__global__ void __launch_bounds__(256, 1) wino_fused_kernel(const WinoFusedArgs a)
{
	constexpr bool PAIR = SCHED > 0;
	// The stores are STREAMING (nt).  Measured with the L2 -> fabric request counters (tools/r05_policy.sh, profiles/r05_v7_store_policy.txt): a patch line
	// (128 bytes = 32 channels of a pixel) is consumed 32 bytes per trip, and with the default policy the output -- 2.1 MB per trip and XCD at conv1_2 -- pushes it
	// out of the XCD's 4 MB L2 between its chunk pairs: conv1_2 read 6.67 GB per launch for a 3.29 GB input, conv2_1 2.03 GB for 0.82.  With nt the output lines
	// leave first: 5.42 and 1.41 GB, the time unchanged at batch 256 (within 0.5 %) and 3..5 % better at batch 64.  (sc1, sc1 nt, sc0 sc1: the same reads as nt.)
	// DBG bits 14..16 select another policy for the probe (5 = the default policy).
	constexpr int ST_POLICY = ((DBG >> 14) & 7) == 0 ? 1 : (((DBG >> 14) & 7) == 5 ? 0 : ((DBG >> 14) & 7));
	constexpr bool NEWST = (DBG & 8192) == 0; // the epilogue's stores with scalar address arithmetic (round 5; DBG bit 8192 = the previous form, for tools/wf5_probe.cpp)
	// (also measured and NOT adopted: the output transform on packed pairs -- the two channel blocks of a position as one f2, 400 v_pk_* instead of 800 scalar
	// operations per item -- makes hipcc spill around the epilogue: conv1_2 4.68 vs 3.07 ms.  profiles/r05_v3_wf5_probe.txt)
	// (Round 5, measured and NOT adopted -- tools/wf5_probe.cpp, profiles/r05_v3_wf5_probe.txt: letting an item's last trip skip the next item's first transform
	// and running it behind the epilogue instead (S and V dead across the epilogue, which hipcc otherwise parks in the emptying AGPRs) is 3 % SLOWER,
	// conv1_2 3.21 vs 3.11 ms: the transform's patch reads are then waited for with nothing else to issue.)
	typedef WfPair<(SCHED > 0 ? SCHED - 1 : 0)> PS;
	typedef WfGeom<GH, GW> G;
	constexpr int GWL = GW == 4 ? 2 : (GW == 8 ? 3 : (GW == 2 ? 1 : (GW == 16 ? 4 : 0)));
	static_assert(GWL >= 1 && GWL <= 3, "tile groups 8 x 2, 4 x 4, 2 x 8");
	constexpr int FSH = 3 - GWL; // see the patch reads
	__shared__ __attribute__((aligned(16))) float lds[2 * WF_U_FLOATS + 8 * WF_P_FLOATS]; // 160 KB: [U ring x2][patch x2 per wave]
	const int t = threadIdx.x;
	const int lane = t & 63;
	const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
	// Work items = (tile-group quad gq, k block kb).  A TEAM of a.team workgroups on ONE XCD (workgroup b runs on XCD b % 8)
	// walks the same range of quads side by side, member m taking the k blocks m, m + team, ...: the members fetch the same
	// patch lines at about the same time, so all but the first find them in that XCD's L2 (a workgroup walking the k blocks
	// of a quad one after the other does not: the other 31 CUs of the XCD push 10 MB through its 4 MB L2 in between).
	const int nquads = (a.groups + 3) / 4;
	const int team = a.team, kpm = a.KB / team;              // k blocks per member
	const int slot = (int)blockIdx.x >> 3, teams = ((int)gridDim.x >> 3) / team * 8; // (host: gridDim.x % (8 * team) == 0)
	const int member = slot % team, team_id = ((int)blockIdx.x & 7) * (teams >> 3) + slot / team;
	const int qper = (nquads + teams - 1) / teams;
	const int q_first = team_id * qper;
	const int q_count = nquads - q_first < qper ? nquads - q_first : qper;
	const int first = 0, count = q_count * kpm;              // this workgroup's items, numbered 0 .. count - 1
	if (count <= 0) return;
	const int gpi = a.GYn * a.GXn;
	const int ti = lane & 15, g = lane >> 4;
	const int ty = ti >> GWL, tx = ti & (GW - 1);

	float* const ubuf = lds;
	float* const pbuf = lds + 2 * WF_U_FLOATS + wave * 2 * WF_P_FLOATS;
	// LDS byte addresses of the DMA destinations, as integers (wave-uniform: SGPRs)
	const unsigned lds0 = __builtin_amdgcn_readfirstlane(wf_lds_addr(lds));
	const unsigned p_lds = lds0 + (2 * WF_U_FLOATS + wave * 2 * WF_P_FLOATS) * 4; // this wave's patch buffers
	const unsigned u_lds = lds0 + wave * 9 * 1024;                               // this wave's ninth of a U buffer

	// ---- work items (everything wave-uniform)
	struct Item { int kb, n, gy, gx, live; };
	auto item_of = [&](const int idx) -> Item {
		Item r;
		const int gq = q_first + idx / kpm;
		r.kb = member + (idx % kpm) * team;
		int group = gq * 4 + wave;
		r.live = group < a.groups;
		if (!r.live) group = a.groups - 1; // computes a duplicate, stores nothing
		r.n = group / gpi;
		const int gr = group - r.n * gpi;
		r.gy = gr / a.GXn;
		r.gx = gr - r.gy * a.GXn;
		return r;
	};
	// ---- DMA descriptors of the item whose chunks are being FETCHED (runs ahead of the item being multiplied)
	wf_rsrc_t rs_src, rs_u;
	unsigned pvoff[WF_P_PIECES]; // byte offset of the lane's 16 bytes of each patch piece inside the image (or WF_OOB)
	int pyx[WF_P_PIECES];        // item-independent half of it: region pixel (Y << 8 | X) and channel half of the lane's granule, -1 = none
#pragma unroll
	for (int q = 0; q < WF_P_PIECES; q++) {
		const int s = q * 64 + lane;
		const int slot = s >> 1; // granule s of the buffer belongs to pixel slot s / 2: the two lanes of a pixel fetch 32 contiguous bytes (one line, one sector)
		const unsigned yx = wf_slot_tab<GH, GW>.yx[slot];
		const int hh = (s & 1) ^ (int)((yx >> (10 + FSH)) & 1); // ... its channel halves swapped in every other band of plane rows (f below): bank spreading for the reads
		pyx[q] = yx == 0xffffu ? -1 : (int)(yx | (unsigned)hh << 16);
	}
	const int sh4 = (int)a.s_sh * 4, sw4 = (int)a.s_sw * 4; // an image spans < 2^31 bytes (host-checked): 32-bit offsets
	auto set_patch = [&](const Item& it) {
		rs_src = wf_make_rsrc(a.src + (long)it.n * a.s_sn, a.src_image_bytes);
		const int Y0 = it.gy * GH * 4 - a.pad_y, X0 = it.gx * GW * 4 - a.pad_x; // region origin in source coordinates
#pragma unroll
		for (int q = 0; q < WF_P_PIECES; q++) {
			const int Y = Y0 + ((pyx[q] >> 8) & 255), X = X0 + (pyx[q] & 255);
			const bool ok = (pyx[q] >= 0) & (Y >= 0) & (Y < a.H) & (X >= 0) & (X < a.W);
			pvoff[q] = ok ? (unsigned)(Y * sh4 + X * sw4 + ((pyx[q] >> 16) & 1) * 16) : WF_OOB;
		}
	};
	auto set_u = [&](const Item& it) { rs_u = wf_make_rsrc(a.uf + (long)it.kb * (a.uf_kb_bytes / 4), a.uf_kb_bytes); };
	const unsigned uvoff = (unsigned)lane * 16u;
	// One LDS-DMA piece (1 KB): q < 11 a piece of a patch chunk (soffset sp = chunk * 32 bytes) to LDS address p_dst + q KB;
	// q >= 11 one of this wave's nine pieces of a U chunk (su = chunk * 36 KB + wave * 9 KB) to u_dst + (q - 11) KB.
	auto dma_piece = [&](auto qc, const unsigned p_dst, const unsigned sp, const unsigned u_dst, const unsigned su) {
		constexpr int q = decltype(qc)::value;
		if constexpr (q < WF_P_PIECES) wf_dma16(rs_src, lds, p_dst + q * 1024, pvoff[q], sp);
		else wf_dma16(rs_u, lds, u_dst + (q - WF_P_PIECES) * 1024, uvoff, su + (q - WF_P_PIECES) * 1024);
	};

	floatx4 acc[36][2]; // zeroed once (defined values for hipcc); every item's first chunk multiplies with C = 0
#pragma unroll
	for (int z = 0; z < 36; z++)
#pragma unroll
		for (int j = 0; j < 2; j++) acc[z][j] = floatx4{ 0.f, 0.f, 0.f, 0.f };

	// per-lane read offsets (floats) into a patch buffer: slot(4 ty + r, 4 tx + c) = const(r, c) + ty * cx(c & 3) + tx
	const int h = g >> 1;
	// (a pixel is 32 bytes, so tiles L and L + 8 of a plane would share their banks: the halves of a pixel are stored swapped where f(py) = bit FSH of the
	// pixel's plane row py = Y >> 2 is set -- tiles ty and ty + 8 / GW then read different halves of the 64 banks; py = ty + (r >> 2), hence two bases per lane)
	const int pb0 = (h ^ ((ty >> FSH) & 1)) * 4 + 2 * (g & 1), pb1 = (h ^ (((ty + 1) >> FSH) & 1)) * 4 + 2 * (g & 1);
	const int b5[2] = { (ty * (GW + 1) + tx) * 8 + pb0, (ty * (GW + 1) + tx) * 8 + pb1 }, b4[2] = { (ty * GW + tx) * 8 + pb0, (ty * GW + tx) * 8 + pb1 };
	auto patch_read = [&](const float* const pb, const int r, const int c) -> f2 {
		const int off = (G::plane_off(r & 3, c & 3) + (r >> 2) * G::cx(c & 3) + (c >> 2)) * 8 + ((c & 3) < 2 ? b5[r >> 2] : b4[r >> 2]);
		const float2 v = *(const float2*)(pb + off);
		return f2{ v.x, v.y };
	};

	// (Round 5, measured and NOT adopted -- profiles/r05_v8_stagger_probe.txt: the teams started out of phase, spread over 0.5 / 1 / 2 items' durations, so that the
	// chip's epilogues would not all store at once: slower by the idle time it adds (+0.3 .. +3 %), nothing recovered -- the epilogues' stores are not waiting for
	// each other across workgroups.)
	// ---- prologue (once per workgroup): the first item's chunk 0 transformed completely (S, then V columns 0..4; column 5 is
	// the first work of the loop), its chunk 1 in flight
	Item cur = item_of(first);
	set_patch(cur);
	set_u(cur);
	f2 S[6][6], Vc[2][6], T[6]; // S[row][column] = d B of the chunk being multiplied; Vc[zx & 1][zy] = column zx of V = B^T S, two columns at a time
	NNC_ASM_NOPS("s_nop 4"); // descriptor words fresh from v_readfirstlane -> the first buffer_load reading them (5 wait states)
	wf_static_for<WF_P_PIECES + 9>([&](auto qc) { dma_piece(qc, p_lds, 0, u_lds, (unsigned)wave * 9216u); });
	wf_static_for<WF_P_PIECES>([&](auto qc) { dma_piece(qc, p_lds + WF_P_FLOATS * 4, WF_CC * 4, 0, 0); }); // (PAIR: sub-chunk 1 of the first 16-channel chunk)
	WF_WAIT_VMCNT(WF_P_PIECES); // chunk 0's patch and this wave's share of U(0) have landed; chunk 1's patch may still fly
	// the transform an item's FIRST trip starts from, out of patch buffer pb (S, and column 0 of V): at the start of the workgroup,
	// and after every epilogue -- S is not carried across the epilogue (72 registers next to the epilogue's own temporaries); the
	// item's last trip skips its share of transform work instead, so the count of VALU per item is unchanged
	auto xform_first = [&](const float* const pb) {
#pragma unroll
		for (int r = 0; r < 6; r++) {
			f2 d[6];
#pragma unroll
			for (int c = 0; c < 6; c++) d[c] = patch_read(pb, r, c);
			wino_bt(d, S[r]);
		}
		const f2 col[6] = { S[0][0], S[1][0], S[2][0], S[3][0], S[4][0], S[5][0] };
		wino_bt(col, Vc[0]);
	};
	xform_first(pbuf);

#define WF_XFORM(X0, X1, X2, X3, X4, X5, Y0, Y1, Y2, Y3, Y4, Y5) do { \
		wf_bt_op<0, 0>(X0, X1, X2, X3, X4, X5, Y0, Y1, Y2, Y3, Y4, Y5, T); wf_bt_op<0, 1>(X0, X1, X2, X3, X4, X5, Y0, Y1, Y2, Y3, Y4, Y5, T); \
		wf_bt_op<0, 2>(X0, X1, X2, X3, X4, X5, Y0, Y1, Y2, Y3, Y4, Y5, T); wf_bt_op<0, 3>(X0, X1, X2, X3, X4, X5, Y0, Y1, Y2, Y3, Y4, Y5, T); \
		wf_bt_op<1, 0>(X0, X1, X2, X3, X4, X5, Y0, Y1, Y2, Y3, Y4, Y5, T); wf_bt_op<2, 2>(X0, X1, X2, X3, X4, X5, Y0, Y1, Y2, Y3, Y4, Y5, T); \
		wf_bt_op<1, 1>(X0, X1, X2, X3, X4, X5, Y0, Y1, Y2, Y3, Y4, Y5, T); wf_bt_op<1, 2>(X0, X1, X2, X3, X4, X5, Y0, Y1, Y2, Y3, Y4, Y5, T); \
		wf_bt_op<1, 3>(X0, X1, X2, X3, X4, X5, Y0, Y1, Y2, Y3, Y4, Y5, T); wf_bt_op<2, 0>(X0, X1, X2, X3, X4, X5, Y0, Y1, Y2, Y3, Y4, Y5, T); \
		wf_bt_op<2, 1>(X0, X1, X2, X3, X4, X5, Y0, Y1, Y2, Y3, Y4, Y5, T); wf_bt_op<2, 3>(X0, X1, X2, X3, X4, X5, Y0, Y1, Y2, Y3, Y4, Y5, T); \
	} while (0)
	// ---- one trip: multiplies the current chunk (U buffer `par`; its S is in registers) while it fetches U of the next chunk of
	// the stream (into U buffer par ^ 1) and the patch of the one after (into patch buffer par), and finally transforms the next
	// chunk's patch (buffer par ^ 1) into S.  36 iterations of 4 MFMAs (iteration it: position z = (it % 6) * 6 + it / 6, column
	// by column); between the MFMAs one U read per iteration (all four fragments of the next position: ds_read_b128), the patch
	// reads and the DMA pieces; the VALU in twelve batches of one six-point transform (24 VALU) each, fenced so hipcc keeps
	// them there -- fp32 MFMA and fp32 VALU do not overlap on gfx950 (tools/coissue_probe.cpp), so WHERE a batch sits costs
	// nothing and registers decide: V exists two columns at a time, S is rewritten once its last column has been used:
	//   it 1, 7, 13, 19, 25   column c + 1 of V from S, during column c's MFMAs (c = it / 6)
	//   it 26 .. 33           S'[r] = d'[r] B of the NEXT chunk, row by row (S is dead after it 25), patch reads from it 23 on
	//   it 34                 column 0 of the next chunk's V
	// MODE (bit 0 FIRST: an item's first chunk multiplies with C = 0 instead of zeroed accumulators; bit 1 LAST: an item's last
	// chunk transforms nothing) exists in the code but only MODE 0 is instantiated: with several trip variants behind a branch
	// hipcc's register allocation falls apart (every accumulator tile spilled to scratch around the merges: 1.1 KB per lane, the
	// kernel 2x slower), so the accumulators are zeroed after each epilogue and the last trip of an item transforms the NEXT
	// item's first chunk -- its S and column 0 of V then live across the epilogue.
	auto trip = [&](auto modec, auto halfc, const int par, const unsigned sp, const unsigned su) {
		constexpr bool FIRST = (decltype(modec)::value & 1) != 0, LAST = (decltype(modec)::value & 2) != 0;
		constexpr int HALF = decltype(halfc)::value; // PAIR: 0 = the even trip of a pair, 1 = the odd one (= par); classic: unused
		const float* const ub = ubuf + par * WF_U_FLOATS + lane * 4;
		const float* const pbn = pbuf + (par ^ 1) * WF_P_FLOATS;
		const unsigned p_dst = p_lds + par * (WF_P_FLOATS * 4), u_dst = u_lds + (par ^ 1) * (WF_U_FLOATS * 4);
		f2 d[2][6];
		float4 u[2];
		if constexpr (DBG & 2) for (int i = 0; i < 12; i++) d[i / 6][i % 6] = f2{ 1.f, 2.f };
		if constexpr (DBG & 4) u[1] = make_float4(1.f, 2.f, 3.f, 4.f);
		u[0] = *(const float4*)(ub);
		wf_static_for<36>([&](auto itc) {
			constexpr int it = decltype(itc)::value;
			constexpr int zx = it / 6, zy = it % 6, z = zy * 6 + zx;
			// the row of the next chunk whose transform sits behind this iteration (-1: none), and the one whose patch reads do
			constexpr int xrow = it == 26 ? 0 : it == 27 ? 1 : it == 29 ? 2 : it == 30 ? 3 : it == 32 ? 4 : it == 33 ? 5 : -1;
			wf_static_for<4>([&](auto kc) {
				constexpr int k = decltype(kc)::value;
				// u = { (j0, e0), (j0, e1), (j1, e0), (j1, e1) }; MFMA order j0 e0, j1 e0, j0 e1, j1 e1: an accumulator recurs every 64 cycles (40 needed)
				if constexpr (DBG & 16) { NNC_PIN_V(acc[z][k & 1][0]); }
				else if constexpr (k == 0 && FIRST) WF_MFMA0(acc[z][0], Vc[zx & 1][zy].x, u[it & 1].x, z < WF_Z_AGPR);
				else if constexpr (k == 1 && FIRST) WF_MFMA0(acc[z][1], Vc[zx & 1][zy].x, u[it & 1].z, z < WF_Z_AGPR);
				else if constexpr (k == 0) WF_MFMA(acc[z][0], Vc[zx & 1][zy].x, u[it & 1].x, z < WF_Z_AGPR);
				else if constexpr (k == 1) WF_MFMA(acc[z][1], Vc[zx & 1][zy].x, u[it & 1].z, z < WF_Z_AGPR);
				else if constexpr (k == 2) WF_MFMA(acc[z][0], Vc[zx & 1][zy].y, u[it & 1].y, z < WF_Z_AGPR);
				else WF_MFMA(acc[z][1], Vc[zx & 1][zy].y, u[it & 1].w, z < WF_Z_AGPR);
				// (a) the transform batches
				if constexpr (k == 3 && !(DBG & 8)) {
					if constexpr (it % 6 == 1 && it < 30) {
						constexpr int c = it / 6 + 1;
						WF_XFORM(S[0][c], S[1][c], S[2][c], S[3][c], S[4][c], S[5][c], Vc[c & 1][0], Vc[c & 1][1], Vc[c & 1][2], Vc[c & 1][3], Vc[c & 1][4], Vc[c & 1][5]);
					} else if constexpr (xrow >= 0 && !LAST) {
						WF_XFORM(d[xrow & 1][0], d[xrow & 1][1], d[xrow & 1][2], d[xrow & 1][3], d[xrow & 1][4], d[xrow & 1][5], S[xrow][0], S[xrow][1], S[xrow][2], S[xrow][3], S[xrow][4], S[xrow][5]);
					} else if constexpr (it == 34 && !LAST) {
						WF_XFORM(S[0][0], S[1][0], S[2][0], S[3][0], S[4][0], S[5][0], Vc[0][0], Vc[0][1], Vc[0][2], Vc[0][3], Vc[0][4], Vc[0][5]);
					}
				}
				// (b) LDS reads: slot 0 the NEXT iteration's four U fragments; slots 1, 2 the next chunk's patch, two neighbouring
				// elements at a time (same base register, so hipcc merges the pair into one ds_read2_b64): row r's three pairs in
				// three consecutive slots, the first one after the transform of row r - 2 has released the registers
				if constexpr (k == 0 && it + 1 < 36 && !(DBG & 4)) {
					constexpr int itn = it + 1, zn = (itn % 6) * 6 + itn / 6;
					u[itn & 1] = *(const float4*)(ub + zn * 256);
				}
				if constexpr (!PAIR && k == 1 && it == 23 && !(DBG & 256)) WF_WAIT_VMCNT(13); // the next chunk's patch (previous trip's pieces): this trip has issued 13 pieces so far
				// PAIR, odd trip: sub-chunk 0 of the next 16 channels (A_0 .. A_10, the last one issued at iteration 8 of this trip); the even trip's
				// reads are of sub-chunk 1, confirmed by the wait of the odd trip before it and the count-zero wait at this trip's start
				if constexpr (PAIR && HALF == 1 && k == 1 && it == 23 && !(DBG & 256)) WF_WAIT_VMCNT(PS::W_XFORM);
				if constexpr ((k == 1 || k == 2) && !(DBG & 2) && !LAST) {
					constexpr int s = it * 2 + (k - 1); // read slot; rows start at slots 46, 49, 54, 57, 60, 63
					constexpr int r = s >= 63 ? 5 : s >= 60 ? 4 : s >= 57 ? 3 : s >= 54 ? 2 : s >= 49 ? 1 : s >= 46 ? 0 : -1;
					constexpr int s0 = r == 5 ? 63 : r == 4 ? 60 : r == 3 ? 57 : r == 2 ? 54 : r == 1 ? 49 : 46;
					if constexpr (r >= 0 && s - s0 < 3) {
						constexpr int c = (s - s0) * 2;
						d[r & 1][c] = patch_read(pbn, r, c);
						d[r & 1][c + 1] = patch_read(pbn, r, c + 1);
					}
				}
				// (c) DMA: the 20 pieces spread evenly over the 36 iterations (piece n behind iteration 9 n / 5; U pieces and patch
				// pieces alternating).  A patch piece gathers 32-byte runs out of 32 different cache lines and keeps the CU's address
				// path busy for ~110 cycles; issued back to back by four waves they queue and the ISSUE blocks (measured: 450 cycles
				// per piece, 1 ms of conv1_2's 4.2).
				if constexpr (!PAIR && k == 3 && !(DBG & 1) && wf_piece_at(it) >= 0) {
					constexpr int n = wf_piece_at(it) < 0 ? 0 : wf_piece_at(it); // (the clamp only matters in the discarded instantiation)
					constexpr int q = (n & 1) ? (n >> 1) : (n <= 16 ? WF_P_PIECES + (n >> 1) : WF_P_PIECES - 1); // even n <= 16: U piece n / 2; odd n: patch piece n / 2; n = 18: the eleventh patch piece
					if constexpr (q < WF_P_PIECES ? !(DBG & 512) : !(DBG & 1024)) dma_piece(GroupId<q>(), p_dst, sp, u_dst, su);
				}
				// (c') PAIR: the pieces of WfPair -- A_q into buffer 0 at soffset sp, B_q into buffer 1 at sp + 32 bytes, U pieces as above
				if constexpr (PAIR && !(DBG & 1) && PS::at(HALF, it, k) >= 0) {
					constexpr int code = PS::at(HALF, it, k) < 0 ? 0 : PS::at(HALF, it, k);
					if constexpr (code < 16) { if constexpr (!(DBG & 512)) dma_piece(GroupId<code>(), p_lds, sp, 0u, 0u); }
					else if constexpr (code < 32) { if constexpr (!(DBG & 512)) dma_piece(GroupId<code - 16>(), p_lds + WF_P_FLOATS * 4, sp + WF_CC * 4, 0u, 0u); }
					else { if constexpr (!(DBG & 1024)) dma_piece(GroupId<WF_P_PIECES + (code - 32)>(), 0u, 0u, u_dst, su); }
				}
				__builtin_amdgcn_sched_barrier(0);
			});
		});
	};

	// ---- the epilogue of an item: b = A^T M A (+ bias) straight from the accumulators -- each lane holds all 36 positions of
	// its 4 tiles x 2 channels (D layout of 16x16x4: column (channel) = lane & 15, row (tile) = 4 * (lane >> 4) + r) -- then
	// through LDS so that a lane stores 16 bytes and a wave whole 128-byte lines: 128 four-byte stores per lane straight from
	// the registers were store-issue bound (1.7 ms of conv1_2's 5.4 at batch 256).  Staging: the U buffer the item's last trip
	// has just read (the DMA in flight targets the other one), one round per r (the 4 tiles 4 g' + r, g' = 0..3: 4 tiles x 16
	// pixels x 32 channels + padding < 9 KB per wave, private: no barrier between rounds).
	auto epilogue = [&](const Item& it, const int par) {
		NNC_ASM_NOPS("s_nop 15\n\ts_nop 15"); // the last MFMAs' results -> the compiler-visible reads below
		if constexpr (DBG & 64) { if (a.bias == (const float*)1) a.dst[t] = acc[0][0][0] + acc[35][1][3] + acc[17][0][1]; return; }
		if constexpr (!(DBG & 32)) __builtin_amdgcn_s_barrier(); // every wave is done reading the U buffer the staging overwrites (the DMA in flight targets the other one)
		constexpr int TS = 16 * 32 + 16; // floats per tile in the staging area: the 4 tiles of a ds_write land on 2 x 16 banks
		float* const st = ubuf + par * WF_U_FLOATS + wave * 2304;
		static_assert(4 * TS <= 2304, "staging area of a wave: a quarter of a U buffer");
		// stores: a buffer descriptor over the item's destination image and per-lane byte offsets; a lane outside the image, the
		// channel range or a dead tile group gets an out-of-range offset and the hardware drops its store -- no branch (the first
		// version's bounds checks were 296 branches and 5 scalar-store fallbacks per store site: 3100 instructions per epilogue)
		typedef unsigned int u4 __attribute__((ext_vector_type(4)));
		const __amdgpu_buffer_rsrc_t rs_dst = __builtin_amdgcn_make_buffer_rsrc((void*)(a.dst + (long)it.n * a.d_sn), 0, a.dst_image_bytes, 0x00020000);
		const int kq = it.kb * WF_KT + (lane & 7) * 4; // the read-back lane owns channels kq .. kq + 3 (K % 4 == 0: all in or all out) of 8 pixels per round
		const bool kok = (it.live != 0) & (kq < a.K);
		float bv[4];
#pragma unroll
		for (int i = 0; i < 4; i++) bv[i] = (a.bias && kq < a.K) ? a.bias[kq + i] : 0.f;
		const int dh4 = (int)a.d_sh * 4, dw4 = (int)a.d_sw * 4;
		// (NEWST) the item's scalars: descriptor, byte offset of its origin + k block, rows / columns left in the image from its origin; the lane's offset or, for a
		// lane whose channels are outside K (or a dead tile group), the out-of-range one
		const wf_rsrc_t rs_item = wf_make_rsrc(a.dst + (long)it.n * a.d_sn, a.dst_image_bytes);
		const int base_y = it.gy * GH * 4, base_x = it.gx * GW * 4;
		const unsigned s_item = (unsigned)base_y * (unsigned)dh4 + (unsigned)base_x * (unsigned)dw4 + (unsigned)it.kb * (WF_KT * 4);
		const int hy0 = a.OH - base_y, hx0 = a.OW - base_x;
		const int ly = lane >> 5, lx = (lane >> 3) & 3;
		const unsigned lane_off_item = kok ? (unsigned)(ly * dh4 + lx * dw4 + (lane & 7) * 16) : WF_OOB;
		const float relu_lo = a.relu ? 0.f : -__builtin_inff();
		// mask bits of the item's four rounds: four 4-byte loads per lane, issued here (nothing stored yet: they come back during round 0's
		// transforms; the wait before the first store covers them)
		unsigned mb[4] = { ~0u, ~0u, ~0u, ~0u };
		if constexpr (MASK) {
			const unsigned* const mp = a.mask_bits + ((((long)it.n * a.GYn + it.gy) * a.GXn + it.gx) * a.KB + it.kb) * 256 + lane;
#pragma unroll
			for (int r = 0; r < 4; r++) mb[r] = it.live ? mp[r * 64] : 0u;
		}
#pragma unroll
		for (int r = 0; r < 4; r++) {
#pragma unroll
			for (int j = 0; j < 2; j++) {
				float s[4][6]; // columns transformed vertically
				if constexpr (DBG & 2048) {
#pragma unroll
					for (int i = 0; i < 4; i++)
#pragma unroll
						for (int jj = 0; jj < 4; jj++) st[g * TS + (i * 4 + jj) * 32 + j * 16 + ti] = acc[i * 4 + jj][j][r];
					continue;
				}
#pragma unroll
				for (int zx = 0; zx < 6; zx++) {
					const float col[6] = { acc[0 * 6 + zx][j][r], acc[1 * 6 + zx][j][r], acc[2 * 6 + zx][j][r], acc[3 * 6 + zx][j][r], acc[4 * 6 + zx][j][r], acc[5 * 6 + zx][j][r] };
					float y[4];
					wino_at(col, y);
#pragma unroll
					for (int i = 0; i < 4; i++) s[i][zx] = y[i];
				}
#pragma unroll
				for (int i = 0; i < 4; i++) {
					float y[4];
					wino_at(s[i], y);
#pragma unroll
					for (int jj = 0; jj < 4; jj++) st[g * TS + (i * 4 + jj) * 32 + j * 16 + ti] = y[jj];
				}
			}
			__builtin_amdgcn_wave_barrier(); // the reads below are other lanes' writes: LDS serves a wave's accesses in order, hipcc must not reorder them
			if (r == 0) WF_WAIT_VMCNT(0);    // before the first store: every DMA piece of the item's last trip has landed, so the next trip
			                                 // starts without a wait (a counted wait behind stores would not be safe)
			// read back: 4 tiles x 16 pixels x 8 channel quads = 512 float4 = 8 per lane; lane -> (channel quad = lane & 7, pixel-in-wave-instruction = lane >> 3)
			if constexpr (NEWST) {
				// Round 5: everything about a store's address that does not depend on the lane is SCALAR -- store e of round r is pixel (e & 1) * 8 + (lane >> 3) of
				// tile 4 (e >> 1) + r, so its row / column inside the item are compile-time constants plus the lane's (ly, lx) -- and goes into the instruction's
				// scalar offset; the lane keeps ONE byte offset for the whole kernel and only swaps it for the out-of-range offset where its pixel is outside the
				// image (two compares with scalars, one select).  Before: a 32 x 32 -> 64-bit multiply-add, a quarter-rate 32-bit multiply, an add-shift, five
				// compares / selects and the descriptor's 64-bit base rebuilt on the scalar unit -- per store, 32 stores per item.
#pragma unroll
				for (int e = 0; e < 8; e++) {
					const int tile = 4 * (e >> 1) + r;
					const int cy = (tile >> GWL) * 4 + (e & 1) * 2, cx = (tile & (GW - 1)) * 4; // the store's pixel = item origin + (cy + ly, cx + lx)
					const float4 v = *(const float4*)(st + (e >> 1) * TS + ((e & 1) * 8 + (lane >> 3)) * 32 + (lane & 7) * 4);
					const unsigned soff = s_item + (unsigned)cy * (unsigned)dh4 + (unsigned)cx * (unsigned)dw4;
					const bool ok = (ly < hy0 - cy) & (lx < hx0 - cx);
					const unsigned voff = ok ? lane_off_item : WF_OOB;
					float o0 = wf_max(v.x + bv[0], relu_lo), o1 = wf_max(v.y + bv[1], relu_lo), o2 = wf_max(v.z + bv[2], relu_lo), o3 = wf_max(v.w + bv[3], relu_lo);
					if constexpr (MASK) {
						const unsigned m = mb[r] >> (4 * e);
						o0 = (m & 1) ? o0 : 0.f; o1 = (m & 2) ? o1 : 0.f; o2 = (m & 4) ? o2 : 0.f; o3 = (m & 8) ? o3 : 0.f;
					}
					if constexpr (DBG & 128) { NNC_PIN_V(o0); NNC_PIN_V(o1); NNC_PIN_V(o2); NNC_PIN_V(o3); }
					else wf_store16<ST_POLICY>(rs_item, voff, soff, o0, o1, o2, o3);
				}
			} else {
#pragma unroll
			for (int e = 0; e < 8; e++) {
				const int pid = e * 8 + (lane >> 3);   // 0..63: tile slot pid >> 4 (= g' of the writers), pixel pid & 15
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
				if constexpr (DBG & 128) { NNC_PIN_V(o0); NNC_PIN_V(o1); NNC_PIN_V(o2); NNC_PIN_V(o3); }
				else __builtin_amdgcn_raw_buffer_store_b128(u4{ __float_as_uint(o0), __float_as_uint(o1), __float_as_uint(o2), __float_as_uint(o3) }, rs_dst, voff, 0, 0);
			}
			}
			__builtin_amdgcn_wave_barrier(); // next round's writes after this round's reads
		}
#pragma unroll
		for (int z = 0; z < 36; z++)
#pragma unroll
			for (int j = 0; j < 2; j++) acc[z][j] = floatx4{ 0.f, 0.f, 0.f, 0.f }; // the next item starts from zero (one trip variant: see trip)
	};

	// ---- the stream, PAIRED schedule: an item is CCn / 2 pairs of trips; pair m multiplies sub-chunks 2m, 2m + 1 and fetches the 16 channels
	// of pair m + 1 (the last pair of an item: the next item's first 16) -- see WfPair
	if constexpr (PAIR) {
		const int M = a.CCn >> 1; // (host: CCn even)
		for (int ii = 0; ii < count; ii++) {
			const Item nxt = item_of(ii + 1 < count ? first + ii + 1 : first + ii);
			for (int m = 0; m < M; m++) {
				const bool last = m == M - 1;
				// even trip.  Every DMA piece of the previous pair has to have landed (its U pieces for this trip, its B pieces for this trip's
				// transform); right after an epilogue there is nothing to wait for -- the epilogue confirmed every piece, and a count-zero wait
				// there would wait for its stores' acknowledgements
				if constexpr (!(DBG & 256)) { if (m != 0 || ii == 0) WF_WAIT_VMCNT(0); }
				if constexpr (!(DBG & 32)) __builtin_amdgcn_s_barrier();
				if (last) set_patch(nxt); // this pair's patch pieces (from t = 34 on) are the next item's first 16 channels
				const unsigned sp = last ? 0u : (unsigned)(m + 1) * (2 * WF_CC * 4);
				trip(GroupId<0>(), GroupId<0>(), 0, sp, (unsigned)(2 * m + 1) * (WF_U_FLOATS * 4) + (unsigned)wave * 9216u);
				// odd trip: the even trip's U pieces have landed, its last patch pieces may fly
				if constexpr (!(DBG & 256)) WF_WAIT_VMCNT(PS::W_ODD_START);
				if constexpr (!(DBG & 32)) __builtin_amdgcn_s_barrier();
				if (last) set_u(nxt);
				trip(GroupId<0>(), GroupId<1>(), 1, sp, (last ? 0u : (unsigned)(2 * m + 2) * (WF_U_FLOATS * 4)) + (unsigned)wave * 9216u);
			}
			epilogue(cur, 1);
			cur = nxt;
		}
	} else {
	// ---- the stream, classic schedule
	int gtrip = 0;
	for (int ii = 0; ii < count; ii++) {
		const Item nxt = item_of(ii + 1 < count ? first + ii + 1 : first + ii); // (the last item fetches its own first chunks again: harmless)
		for (int cc = 0; cc < a.CCn; cc++, gtrip++) {
			const int par = gtrip & 1;
			// The previous trip's U pieces (the next chunk's patch is waited for where the trip first reads it): its last THREE
			// pieces (n = 17, 18, 19: patch pieces 8, 10, 9) are patch pieces and may stay in flight; n = 16 is U piece 8 and must
			// have landed (a count of 4 let it fly: wrong values whenever U came from HBM instead of L2 -- found by smoke() on the
			// MI355X, first touch of freshly written fragments; the emulator's DMA is synchronous).  Counted waits are used
			// only where the newest outstanding operations are all loads -- loads retire in order, stores need not retire in order
			// with them; after an epilogue (stores) there is nothing to wait for, the epilogue has confirmed every piece.
			if constexpr (!(DBG & 256)) { if (ii == 0 && cc == 0) WF_WAIT_VMCNT(0); else if (cc != 0) WF_WAIT_VMCNT(3); }
			if constexpr (!(DBG & 32)) __builtin_amdgcn_s_barrier(); // => all of this chunk's U is in LDS, and every wave is done with the buffers this trip's DMA overwrites
			if (cc == a.CCn - 2) set_patch(nxt); // the patch fetched from now on (chunk cc + 2) belongs to the next item
			if (cc == a.CCn - 1) set_u(nxt);     // and so does the U chunk
			const int c2 = cc + 2 >= a.CCn ? cc + 2 - a.CCn : cc + 2, c1 = cc + 1 >= a.CCn ? 0 : cc + 1;
			const unsigned sp = (unsigned)c2 * (WF_CC * 4), su = (unsigned)c1 * (WF_U_FLOATS * 4) + (unsigned)wave * 9216u;
			trip(GroupId<0>(), GroupId<0>(), par, sp, su);
		}
		epilogue(cur, (gtrip - 1) & 1); // (S and column 0 of V of the next item's first chunk, made by the last trip, stay in registers across it)
		cur = nxt;
	}
	}
#undef WF_XFORM
}

// The epilogue's mask bits (WinoFusedArgs::mask_bits) from the map a ReLU wrote: block = item in (n, gy, gx, kb) order, thread = (round,
// lane), bit 4 e + i = mask > 0 at the element the epilogue's store e of that round writes at channel i -- the same index arithmetic as
// the epilogue's read-back loop.  Reads the map once in 128-byte runs (eight lanes x 16 bytes per pixel), writes 1/32 of its size.
template <int GH, int GW>
__global__ void __launch_bounds__(256) wino_mask_pack_kernel(const float* __restrict__ mask, const long m_sn, const long m_sh, const long m_sw, unsigned* __restrict__ bits, const int OH, const int OW, const int K, const int GYn, const int GXn, const int KB)
{
	constexpr int GWL = GW == 4 ? 2 : (GW == 8 ? 3 : (GW == 2 ? 1 : (GW == 16 ? 4 : 0)));
	static_assert(GH * GW == 16, "16 tiles per group");
	int b = (int)blockIdx.x;
	const int kb = b % KB; b /= KB;
	const int gx = b % GXn; b /= GXn;
	const int gy = b % GYn;
	const int n = b / GYn;
	const int r = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
	const int kq = kb * WF_KT + (lane & 7) * 4;
	const float* const mp = mask + (long)n * m_sn + kq;
	unsigned m = 0;
#pragma unroll
	for (int e = 0; e < 8; e++) {
		const int pid = e * 8 + (lane >> 3);
		const int gp = pid >> 4, px = pid & 15;
		const int tile = 4 * gp + r;
		const int oy = (gy * GH + (tile >> GWL)) * 4 + (px >> 2), ox = (gx * GW + (tile & (GW - 1))) * 4 + (px & 3);
		if ((kq < K) & (oy < OH) & (ox < OW)) {
			const float4 v = *(const float4*)(mp + (long)oy * m_sh + (long)ox * m_sw);
			m |= ((v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u)) << (4 * e);
		}
	}
	bits[(size_t)blockIdx.x * 256 + threadIdx.x] = m;
}

} // namespace nnc
