// fp32 MFMA contraction core for gfx950: C[M][N] = sum_k A(m,k) * B(k,n), fp32 in / fp32 accumulate on
// v_mfma_f32_32x32x2_f32 (exact fmaf chain, 157 TFLOP/s peak -- MI355X_MICROARCH.md "Matrix cores").
//
// One kernel template serves every contraction on the hot path; what differs is how the two operand
// tiles are *gathered* from HBM (the "loader" functors) and where the result goes (the epilogue functor):
//   conv forward   A = im2col(a)        (k-contiguous gather)   B = w[K][kh][kw][C]   (k-contiguous)
//   conv dgrad     A = im2col(g), flipped taps                  B = w viewed (tap,ko) x c (n-contiguous)
//   conv wgrad     A = g viewed ko x pixel (m-contiguous)       B = im2col(a) pixel x (tap,c) (n-contiguous)
//   GEMM fwd/bwd   plain strided matrices in any of the four transpose combinations
//
// Geometry: (64*WM) x (64*WN) block tile (WM, WN in {1, 2}: 128x128 for the bulk, 128x64 / 64x128 when an output
// dimension is 64 channels so no MFMA issues on padding), BK = 32, 256 threads = 4 waves (2x2), each wave owns
// WM x WN MFMA tiles of 32x32.  Operand tiles are staged HBM -> registers -> LDS (double buffered, one barrier per
// K-step; the global loads run two tiles ahead of the MFMAs, the LDS writes one tile ahead, all issued between MFMAs).
// Gathers are BRANCH-FREE and SELECT-FREE: every lane always issues its 16-byte load; a lane whose element is padding /
// out of range points its load at a 16-byte page of zeros instead, so nothing consumes the loaded registers until the
// LDS store AFTER the MFMA block -- the HBM/L2 latency of tile t+1 hides behind the 64 MFMAs of tile t (a select on the
// loaded value would force the s_waitcnt in front of the MFMAs).  Integer divisions by runtime geometry use precomputed
// multiply-shift constants; per-row and per-k address parts are computed once and added per chunk.
// LDS images:
//   k-contiguous operand:   [rows][36]      (row stride 36 floats: ds_read_b128 by 16-lane groups is conflict free,
//                                            36*r mod 64 hits 16 distinct 4-bank slots for 16 distinct rows)
//   row-contiguous operand: [32 k][rows]    (ds_read_b32, lanes 0-31 read 32 consecutive banks, the other half-wave
//                                            is a different k row: no conflicts)
// Both images feed the same k permutation: MFMA number (q,e) of a K-step consumes k = 8q+e (lanes 0-31) and
// k = 8q+4+e (lanes 32-63), so a k-contiguous operand needs ONE ds_read_b128 per four MFMAs.
// Workgroup -> tile mapping is XCD aware: consecutive tile ids (which share im2col halo rows / weight panels)
// are dispatched to the same XCD so its private 4 MiB L2 serves the re-reads.
#pragma once
#include <hip/hip_runtime.h>
#include "isa.h"

namespace nnc {


constexpr int GEMM_BK = 32, GEMM_THREADS = 256;
constexpr int GEMM_LDK = 36; // row stride of a k-contiguous LDS image

// What a masked-off lane loads instead of its element (see "SELECT-FREE" above): every loader carries `zoff`, the element
// offset from its own base pointer `p` to a 16-byte-aligned page of zeros in the same device's HBM (device_rt.cpp), so a
// masked lane only swaps the OFFSET and the access stays a plain global_load off `p`.
__device__ __forceinline__ float4 ld16(const float* q) { return *(const float4*)q; }


// Exact n / d for 0 <= n < 2^31, d >= 1 as multiply + shift (Granlund-Montgomery): m = ceil(2^(31+s) / d), s = ceil(log2 d).
struct FastDiv {
	unsigned m;
	int sh, d;
	void init(int divisor)
	{
		d = divisor < 1 ? 1 : divisor;
		int s = 0;
		while ((1ll << s) < d) s++;
		sh = 31 + s;
		m = (unsigned)(((1ull << sh) + (unsigned long long)d - 1) / (unsigned long long)d);
	}
	__host__ __device__ __forceinline__ int div(int n) const { return (int)(((unsigned long long)(unsigned)n * m) >> sh); }
};

// Workgroup b of a one-dimensional grid runs on XCD b % 8 (each XCD has its own 4 MB L2).  Kernels whose NEIGHBOURING blocks read overlapping lines -- the 6 x 6
// patches of 4 x 4 Winograd tiles, the rows two 3 x 3 / 2 pooling windows share -- take their work from this LOGICAL block index instead: XCD x gets the x-th
// contiguous eighth of the blocks, so the overlap is found in that XCD's L2 instead of being fetched by up to four XCDs (round 5: profiles/r05_v8_xcd_blocks.txt).
// A bijection of 0 .. grid - 1 for any grid.
__device__ __forceinline__ unsigned nnc_xcd_block(const unsigned bid, const unsigned grid)
{
	const unsigned x = bid & 7, q = grid >> 3, r = grid & 7;
	return x * q + (x < r ? x : r) + (bid >> 3);
}

// Order in which a conv contraction walks its reduction index k = (tap, channel), in K-steps of GEMM_BK channels.
//   natural (taps == 0): tap-major -- a workgroup streams ALL channels of tap 0, then of tap 1, ...; the nine taps re-read
//     the same pixels, but a tap apart lie channels/32 K-steps of the XCD's 64 resident workgroups, i.e. > 4 MB of other
//     lines: every tap misses the XCD's L2 (PMC FETCH_SIZE: 3-4x the algorithmic bytes for forward / dgrad)
//   chunk-major (taps = kh * kw, channels % 32 == 0): k' = (channel chunk, tap, channel in chunk): the taps of one
//     32-channel chunk are consecutive K-steps, so the re-reads come out of L2.  Pure re-association of the sum; both
//     operands' loaders are handed the mapped K offset, nothing in them changes.
struct KOrder {
	int taps = 0, C = 0, K = 0;
	FastDiv d;
	void init(int taps_, int C_) { taps = taps_; C = C_; K = taps_ * C_; d.init(taps_); }
};

// ---------------------------------------------------------------------------------------------- loaders
// Concept:
//   static constexpr bool KCONTIG;  const float* p;
//   Ctx  make(int r) const;                 per-row state (KCONTIG) or per-row-chunk state (!KCONTIG), computed once
//   KCtx kctx(int k, int klimit) const;     per-k state, computed once per K-step (KCONTIG) / per chunk (!KCONTIG)
//   long   offset(const Ctx&, const KCtx&) const;   VEC only: element offset from p of the chunk (or zoff when masked)
//   static constexpr bool INCR; void advance(KCtx&, int klimit) const;   INCR: step a k state by one K-step (BK = 32) with
//     adds, compares and selects only -- the steady state then carries NO integer multiply or division (on gfx950
//     v_mul_lo_u32 / v_mad_u64_u32 are quarter-rate and their issue time is NOT hidden behind a wave's own MFMAs: the
//     division-based wgrad loop spent 56 of them per K-step).  The host may pick INCR only when one step wraps each counter
//     at most once (channels >= 32 per tap, images at least 32/OW + 1 rows, ...).
//     MEASURED NEGATIVE (round 1): the incremental form needs more instructions (selects on 64-bit offsets) and ran
//     98 vs 107 TFLOP/s on wgrad, 114 vs 124 on dgrad -- instruction count, not multiply latency, is what the loop pays.
//     The launchers therefore instantiate INC = false; the code stays as the record of the experiment.
//   float4 load(const Ctx&, const KCtx&) const;     !VEC only: the chunk gathered by four scalar loads
//     KCONTIG:  elements (r, k..k+3)   (k multiple of 4);   !KCONTIG: elements (r..r+3, k)   (r multiple of 4)
// Rows >= R, k >= klimit and im2col padding read as zero (from the page of zeros).  VEC = one 16-byte global load per chunk
// (needs the alignment / divisibility the host checks before picking it); !VEC = four scalar loads.

// Plain matrix: element(r, k) = p[r * ldr + k * ldk].  KC => ldk == 1, otherwise ldr == 1.
template <bool KC, bool VEC>
struct MatLoader {
	static constexpr bool KCONTIG = KC, VECTOR = VEC, INCR = false;
	const float* p;
	long zoff;
	long ldr, ldk;
	int R, K;
	void finish() {}
	struct Ctx { long off; int r; };
	struct KCtx { long off; int k, klimit; };
	__device__ __forceinline__ void advance(KCtx& x, int klimit) const
	{
		NNC_PIN_V(x.k);
		x.k += GEMM_BK; x.klimit = klimit;
		x.off += KC ? (long)GEMM_BK : (long)GEMM_BK * ldk; // one wave-uniform 64-bit product, hoisted out of the loop by hipcc
	}
	__device__ __forceinline__ Ctx make(int r) const
	{
		Ctx c;
		c.r = r;
		c.off = KC ? (long)r * ldr : (long)r;
		return c;
	}
	__device__ __forceinline__ KCtx kctx(int k, int klimit) const
	{
		KCtx x;
		x.k = k; x.klimit = klimit;
		x.off = KC ? (long)k : (long)k * ldk;
		return x;
	}
	__device__ __forceinline__ void pin(Ctx& c) const { NNC_PIN_V(c.r); }
	__device__ __forceinline__ long offset(const Ctx& c, const KCtx& x) const
	{
		const bool ok = (c.r < R) & (x.k < x.klimit);
		return ok ? c.off + x.off : zoff;
	}
	__device__ __forceinline__ float4 load(const Ctx& c, const KCtx& x) const
	{
		const bool ok = (c.r < R) & (x.k < x.klimit);
		float v[4];
#pragma unroll
		for (int e = 0; e < 4; e++) {
			const bool oke = ok & (KC ? x.k + e < x.klimit : c.r + e < R);
			v[e] = p[oke ? c.off + x.off + e : zoff];
		}
		return make_float4(v[0], v[1], v[2], v[3]);
	}
};

// Rows of P-element planes, the reduction index running over (image n, position p): element(r, k = n * P + p) = p[n * s_n + r * ldr + p].
// Serves the filter gradient of a 1x1 convolution on NCHW tensors (both operands: channel rows of [N][C][P] tensors, reduced over
// every position of every image) without re-laying the tensors out.  VEC needs P % 4 == 0 (a chunk stays inside one plane).
template <bool VEC>
struct PlaneKC {
	static constexpr bool KCONTIG = true, VECTOR = VEC, INCR = false;
	const float* p;
	long zoff;
	long ldr, s_n;
	int R, K, P; // K = N * P
	FastDiv d_p;
	void finish() { d_p.init(P); }
	struct Ctx { long off; int r; };
	struct KCtx { long off; int k, klimit; };
	__device__ __forceinline__ void advance(KCtx&, int) const {}
	__device__ __forceinline__ Ctx make(int r) const { Ctx c; c.r = r; c.off = (long)r * ldr; return c; }
	__device__ __forceinline__ KCtx kctx(int k, int klimit) const
	{
		KCtx x;
		x.k = k; x.klimit = klimit;
		const int kk = k < klimit ? k : 0;
		const int n = d_p.div(kk);
		x.off = (long)n * s_n + (kk - n * P);
		return x;
	}
	__device__ __forceinline__ void pin(Ctx& c) const { NNC_PIN_V(c.r); }
	__device__ __forceinline__ long offset(const Ctx& c, const KCtx& x) const
	{
		const bool ok = (c.r < R) & (x.k < x.klimit);
		return ok ? c.off + x.off : zoff;
	}
	__device__ __forceinline__ float4 load(const Ctx& c, const KCtx& x) const
	{
		float v[4];
#pragma unroll
		for (int e = 0; e < 4; e++) { // the four k's may straddle a plane boundary when P % 4 != 0
			const int k = x.k + e;
			const bool ok = (c.r < R) & (k < x.klimit);
			const int kk = ok ? k : 0;
			const int n = d_p.div(kk);
			v[e] = p[ok ? c.off + (long)n * s_n + (kk - n * P) : zoff];
		}
		return make_float4(v[0], v[1], v[2], v[3]);
	}
};

// im2col gather with the reduction index running (tap_y, tap_x, channel), channel fastest: rows are output
// pixels m = (n, oy, ox).  Serves conv forward (source = a) and conv dgrad (source = g, taps walked backwards).
//   t_y = oy * my + oy_off + i * ty ;  source y = t_y / dv_y, valid iff t_y >= 0, t_y % dv_y == 0, y < H   (same for x)
//   forward: my = stride, oy_off = -border, ty = +dilation, dv = 1
//   dgrad:   my = 1, oy_off = +border, ty = -dilation, dv = stride          (STRIDED <=> some dv != 1)
// Address = row part (n, oy, ox: once per tile) + tap part (i, j, ch: once per K-step) when !STRIDED.
// VEC: one 16-byte load per chunk (C % 4 == 0 keeps a chunk inside one tap); !VEC: the four k's of a chunk are resolved
// one by one (they may straddle taps: conv1_1 has C = 3).
template <bool VEC, bool STRIDED, bool INC = false>
struct Im2colKC {
	static constexpr bool KCONTIG = true, VECTOR = VEC, INCR = INC;
	const float* p;
	long zoff;
	long s_n;
	int s_h, s_w;
	int H, W;
	int OW, OHW, M;
	int C, KWC, K;
	int my, mx, oy_off, ox_off, ty, tx, dv_y, dv_x;
	FastDiv d_ohw, d_ow, d_kwc, d_c, d_dvy, d_dvx;
	int KWt, w1_off, w2_off; // INC: kw * tx; offset corrections when the channel / the tap column wraps
	void finish()
	{
		d_ohw.init(OHW); d_ow.init(OW); d_kwc.init(KWC); d_c.init(C); d_dvy.init(dv_y); d_dvx.init(dv_x);
		const int KW = KWC / (C > 0 ? C : 1);
		KWt = KW * tx;
		w1_off = STRIDED ? -C : tx * s_w - C;
		w2_off = STRIDED ? 0 : ty * s_h - KW * tx * s_w;
	}
	static bool incr_ok(int C) { return C >= GEMM_BK; } // at most one tap wrap per K-step
	struct Ctx { long base; int iy0, ix0; }; // !STRIDED: base already includes iy0 * s_h + ix0 * s_w
	struct K1 { int dy, dx, off; bool ok; int ch, k; }; // !STRIDED: off = dy * s_h + dx * s_w + ch;  STRIDED: off = ch
	struct KCtx { K1 e[VEC ? 1 : 4]; };
	__device__ __forceinline__ Ctx make(int m) const
	{
		Ctx c;
		if (m >= M) { c.base = 0; c.iy0 = -(1 << 28); c.ix0 = -(1 << 28); return c; } // every tap fails the y >= 0 test
		const int n = d_ohw.div(m);
		const int rem = m - n * OHW;
		const int oy = d_ow.div(rem);
		const int ox = rem - oy * OW;
		c.iy0 = oy * my + oy_off;
		c.ix0 = ox * mx + ox_off;
		c.base = (long)n * s_n;
		if (!STRIDED) c.base += (long)(c.iy0 * s_h + c.ix0 * s_w); // one image spans < 2^31 elements (host-checked)
		return c;
	}
	__device__ __forceinline__ K1 k1(int k, int klimit) const
	{
		K1 x;
		x.ok = k < klimit;
		const int kk = x.ok ? k : 0;
		const int i = d_kwc.div(kk);
		const int r = kk - i * KWC;
		const int j = d_c.div(r);
		const int ch = r - j * C;
		x.ch = ch; x.k = k;
		x.dy = i * ty;
		x.dx = j * tx;
		x.off = STRIDED ? ch : x.dy * s_h + x.dx * s_w + ch;
		return x;
	}
	// k += 32 with at most one wrap of the channel counter into the next tap column and of the column into the next tap row
	__device__ __forceinline__ void advance(KCtx& kc, int klimit) const
	{
		K1& x = kc.e[0];
		NNC_PIN_V(x.k);
		x.k += GEMM_BK;
		x.ok = x.k < klimit;
		x.ch += GEMM_BK;
		x.off += GEMM_BK;
		const bool w1 = x.ch >= C;
		x.ch -= w1 ? C : 0;
		x.dx += w1 ? tx : 0;
		x.off += w1 ? w1_off : 0;
		const bool w2 = tx > 0 ? x.dx >= KWt : x.dx <= KWt; // tx < 0 for dgrad (taps walked backwards)
		x.dx -= w2 ? KWt : 0;
		x.dy += w2 ? ty : 0;
		x.off += w2 ? w2_off : 0;
	}
	__device__ __forceinline__ KCtx kctx(int k, int klimit) const
	{
		KCtx x;
#pragma unroll
		for (int e = 0; e < (VEC ? 1 : 4); e++) x.e[e] = k1(k + e, klimit);
		return x;
	}
	__device__ __forceinline__ long locate(const Ctx& c, const K1& x) const
	{
		const int y = c.iy0 + x.dy, xx = c.ix0 + x.dx;
		if (!STRIDED) {
			const bool ok = x.ok & ((unsigned)y < (unsigned)H) & ((unsigned)xx < (unsigned)W);
			return ok ? c.base + x.off : zoff;
		}
		bool ok = x.ok & (y >= 0) & (xx >= 0);
		const int yy = ok ? y : 0, xq = ok ? xx : 0;
		const int qy = d_dvy.div(yy), qx = d_dvx.div(xq);
		ok = ok & (qy * dv_y == yy) & (qx * dv_x == xq) & (qy < H) & (qx < W);
		return ok ? c.base + (long)(qy * s_h + qx * s_w + x.off) : zoff;
	}
	__device__ __forceinline__ void pin(Ctx& c) const { NNC_PIN_V(c.iy0); }
	__device__ __forceinline__ long offset(const Ctx& c, const KCtx& x) const { return locate(c, x.e[0]); }
	__device__ __forceinline__ float4 load(const Ctx& c, const KCtx& x) const
	{
		float v[4];
#pragma unroll
		for (int e = 0; e < (VEC ? 1 : 4); e++) v[e] = p[locate(c, x.e[e])];
		return make_float4(v[0], v[1], v[2], v[3]);
	}
};

// conv dgrad weights: B(k = (tap, ko), n = c) = w[ko][tap][c]  (n contiguous).
template <bool VEC, bool INC = false>
struct WgtDgradNC {
	static constexpr bool KCONTIG = false, VECTOR = VEC, INCR = INC;
	const float* p;
	long zoff;
	long ko_stride; // kh*kw*C
	int C, Ko, K;   // K = kh*kw*Ko
	FastDiv d_ko;
	void finish() { d_ko.init(Ko); }
	static bool incr_ok(int Ko) { return Ko >= GEMM_BK; }
	struct Ctx { int c; };
	struct KCtx { long off; bool ok; int ko, k; };
	__device__ __forceinline__ void advance(KCtx& x, int klimit) const
	{
		NNC_PIN_V(x.k);
		x.k += GEMM_BK;
		x.ok = x.k < klimit;
		x.ko += GEMM_BK;
		x.off += (long)GEMM_BK * ko_stride;
		const bool w = x.ko >= Ko;
		x.ko -= w ? Ko : 0;
		x.off += w ? (long)C - (long)Ko * ko_stride : 0L;
	}
	__device__ __forceinline__ Ctx make(int c) const { Ctx x; x.c = c; return x; }
	__device__ __forceinline__ KCtx kctx(int k, int klimit) const
	{
		KCtx x;
		x.ok = k < klimit;
		const int kk = x.ok ? k : 0;
		const int tap = d_ko.div(kk);
		const int ko = kk - tap * Ko;
		x.ko = ko; x.k = k;
		x.off = (long)ko * ko_stride + (long)tap * C;
		return x;
	}
	__device__ __forceinline__ void pin(Ctx& c) const { NNC_PIN_V(c.c); }
	__device__ __forceinline__ long offset(const Ctx& c, const KCtx& x) const
	{
		const bool ok = x.ok & (c.c < C);
		return ok ? x.off + c.c : zoff;
	}
	__device__ __forceinline__ float4 load(const Ctx& c, const KCtx& x) const
	{
		const bool ok = x.ok & (c.c < C);
		float v[4];
#pragma unroll
		for (int e = 0; e < 4; e++) {
			const bool oke = ok & (c.c + e < C);
			v[e] = p[oke ? x.off + c.c + e : zoff];
		}
		return make_float4(v[0], v[1], v[2], v[3]);
	}
};

// conv wgrad activations: B(k = pixel (n, oy, ox), nn = (tap_y, tap_x, c)) = a[n, oy*sy - py + i*dy, ox*sx - px + j*dx, c].
// Address = pixel part (once per k) + column part (tap and channel: once per tile).
template <bool VEC, bool INC = false>
struct Im2colNC {
	static constexpr bool KCONTIG = false, VECTOR = VEC, INCR = INC;
	const float* p;
	long zoff;
	long s_n;
	int s_h, s_w;
	int H, W;
	int OW, OHW;
	int C, KWC, NN, K; // NN = kh*kw*C, K = N*OH*OW
	int sy, sx, py, px, dy, dx;
	FastDiv d_ohw, d_ow;
	// INC: one K-step = 32 pixels further = q32 rows and r32 columns, then at most one column wrap and one image wrap
	int OH, q32, r32, st_by, st_bx, w1_by, w1_bx, w2_by;
	long st_base, w1_base, w2_base;
	void finish()
	{
		d_ohw.init(OHW); d_ow.init(OW);
		OH = OHW / (OW > 0 ? OW : 1);
		q32 = GEMM_BK / OW; r32 = GEMM_BK % OW;
		st_by = q32 * sy; st_bx = r32 * sx; st_base = (long)st_by * s_h + (long)st_bx * s_w;
		w1_by = sy; w1_bx = -OW * sx; w1_base = (long)sy * s_h - (long)OW * sx * s_w;
		w2_by = -OH * sy; w2_base = s_n - (long)OH * sy * s_h;
	}
	static bool incr_ok(int OH, int OW) { return GEMM_BK / OW + 1 <= OH; }
	struct Ctx { int off_y[VEC ? 1 : 4], off_x[VEC ? 1 : 4], off[VEC ? 1 : 4]; bool ok[VEC ? 1 : 4]; }; // per column: i*dy - py, j*dx - px, their offset + c
	struct KCtx { long base; int by, bx; bool ok; int ox, oy, k; }; // base = n * s_n + by * s_h + bx * s_w
	__device__ __forceinline__ void advance(KCtx& x, int klimit) const
	{
		NNC_PIN_V(x.k);
		x.k += GEMM_BK;
		x.ok = x.k < klimit;
		x.ox += r32; x.oy += q32; x.bx += st_bx; x.by += st_by; x.base += st_base;
		const bool w1 = x.ox >= OW;
		x.ox -= w1 ? OW : 0; x.oy += w1 ? 1 : 0;
		x.bx += w1 ? w1_bx : 0; x.by += w1 ? w1_by : 0; x.base += w1 ? w1_base : 0L;
		const bool w2 = x.oy >= OH;
		x.oy -= w2 ? OH : 0;
		x.by += w2 ? w2_by : 0; x.base += w2 ? w2_base : 0L;
	}
	__device__ __forceinline__ Ctx make(int nn) const
	{
		Ctx c;
#pragma unroll
		for (int e = 0; e < (VEC ? 1 : 4); e++) { // VEC: the four columns share a tap and are channel-consecutive
			const int q = nn + e;
			const int i = q / KWC;
			const int r = q - i * KWC;
			const int j = r / C;
			c.off_y[e] = i * dy - py;
			c.off_x[e] = j * dx - px;
			c.off[e] = c.off_y[e] * s_h + c.off_x[e] * s_w + (r - j * C);
			c.ok[e] = q < NN;
		}
		return c;
	}
	__device__ __forceinline__ KCtx kctx(int k, int klimit) const
	{
		KCtx x;
		x.ok = k < klimit;
		const int kk = x.ok ? k : 0;
		const int n = d_ohw.div(kk);
		const int rem = kk - n * OHW;
		const int oy = d_ow.div(rem);
		const int ox = rem - oy * OW;
		x.ox = ox; x.oy = oy; x.k = k;
		x.by = oy * sy;
		x.bx = ox * sx;
		x.base = (long)n * s_n + (long)(x.by * s_h + x.bx * s_w);
		return x;
	}
	__device__ __forceinline__ long locate(const Ctx& c, const KCtx& x, const int e) const
	{
		const int y = x.by + c.off_y[e], xx = x.bx + c.off_x[e];
		const bool ok = x.ok & c.ok[e] & ((unsigned)y < (unsigned)H) & ((unsigned)xx < (unsigned)W);
		return ok ? x.base + c.off[e] : zoff;
	}
	__device__ __forceinline__ void pin(Ctx& c) const { NNC_PIN_V(c.off_y[0]); }
	__device__ __forceinline__ long offset(const Ctx& c, const KCtx& x) const { return locate(c, x, 0); }
	__device__ __forceinline__ float4 load(const Ctx& c, const KCtx& x) const
	{
		float v[4];
#pragma unroll
		for (int e = 0; e < (VEC ? 1 : 4); e++) v[e] = p[locate(c, x, e)];
		return make_float4(v[0], v[1], v[2], v[3]);
	}
};

// ---------------------------------------------------------------------------------------------- epilogues
// Two ways out of the accumulators:
//   * operator()(m, n, v): one element -- any output strides; a lane of the 32x32 MFMA holds a COLUMN of its tile (16 rows of one n), so a wave's store
//     instruction covers 2 rows x 32 consecutive n: 128-byte segments of fp32, 64-byte segments of halves, 16 store instructions per tile;
//   * store4(m, n, v): four consecutive n of row m in one 16-byte (8-byte for halves) access -- the kernels stage the block tile through LDS (free by
//     then) and read it back ROW-major (epi_flush_rows below), so a wave instruction writes 1 KB / 512 B contiguous.  `vec` (set by the launcher)
//     says the output allows it: n-contiguous (ldn == 1), N, ldm and every base offset multiples of 4 elements, 16-byte (8-byte) aligned.
//     Measured on the MI355X (tools/conv1x1_bench.py, profiles/r04_v4_conv1x1_bench_*.txt): the 1x1 convolutions of ResNet-50 that WRITE the wide tensor
//     (64 -> 256 at 56^2: 1.0 GB written per launch) ran 1.25 TB/s (f16) / 1.7 TB/s (fp32) with the scalar stores while the same shape's data
//     gradient, which READS the wide tensor, ran 4.5 / 3.9 TB/s.
// Both forms apply alpha, bias and the accumulate flag in the same order, so their results are bit-identical.
// Direct store: c[m*ldm + n*ldn] = alpha * acc (+ bias[n]) (+ old c when accumulating).
struct EpiStore {
	float* c;
	long ldm, ldn;
	const float* bias; // bias[m * bias_ldm + n * bias_ldn]; (0, 1): one row broadcast over m; (1, 0): one value per output ROW; may be null
	float alpha;
	int accumulate;
	int M, N;
	long bias_ldm, bias_ldn;
	int vec = 0;
	static constexpr int FLUSH_UNROLL = 2;
	__device__ __forceinline__ void operator()(int m, int n, float v) const
	{
		if (m < M && n < N) {
			const long o = (long)m * ldm + (long)n * ldn;
			v *= alpha;
			if (bias) v += bias[(long)m * bias_ldm + (long)n * bias_ldn];
			if (accumulate) v += c[o];
			c[o] = v;
		}
	}
	__device__ __forceinline__ void store4(int m, int n, float4 v) const
	{
		if (m < M && n < N) {
			const long o = (long)m * ldm + (long)n;
			v.x *= alpha; v.y *= alpha; v.z *= alpha; v.w *= alpha;
			if (bias) {
				const float* const b = bias + (long)m * bias_ldm + (long)n * bias_ldn;
				v.x += b[0]; v.y += b[bias_ldn]; v.z += b[2 * bias_ldn]; v.w += b[3 * bias_ldn];
			}
			if (accumulate) { const float4 u = *(const float4*)(c + o); v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
			*(float4*)(c + o) = v;
		}
	}
};
// Split-K partial: slab[blockIdx.y][m][n] = acc; splitk_reduce_kernel finishes (deterministic order).
struct EpiPartial {
	float* c; // workspace; the kernel advances it to this block's slab
	const float* bias; // unused (applied by splitk_reduce_kernel); keeps the epilogue concept uniform
	long slab; // M*N
	int M, N;
	int vec = 0;
	static constexpr int FLUSH_UNROLL = 1;
	__device__ __forceinline__ void operator()(int m, int n, float v) const
	{
		if (m < M && n < N) c[(long)m * N + n] = v;
	}
	__device__ __forceinline__ void store4(int m, int n, const float4 v) const
	{
		if (m < M && n < N) *(float4*)(c + (long)m * N + n) = v;
	}
};

// The row-major read-back of a staged block-tile slice: `cs` holds ROWS x BN accumulators (row pitch BN + 8 floats: the two half-waves of a staging write --
// four rows apart -- land 32 banks apart, and rows stay 16-byte aligned); thread t takes 16-byte vectors t, t + NT, ... and hands each to the epilogue's
// store4 with the output row the staged row stands for (row_of) -- consecutive lanes = consecutive 16 bytes of one output row.
// the same in vectors of EIGHT (half-precision outputs with vec == 2: one 16-byte store per lane)
template <class EPI> struct epi_has_store8 { template <class E> static auto test(int) -> decltype(&E::store8, char()); template <class E> static long test(...); static constexpr bool value = sizeof(test<EPI>(0)) == 1; };
template <int NT, int ROWS, int BN, class EPI, class ROWOF>
__device__ __forceinline__ void epi_flush_rows8(const float* const cs, const EPI& epi, const int m0, const int n0, const int t, const ROWOF& row_of)
{
	constexpr int PITCH = BN + 8, V = BN / 8, TOTAL = ROWS * V;
	static_assert(TOTAL % NT == 0, "whole vectors per thread");
#pragma unroll EPI::FLUSH_UNROLL
	for (int j = 0; j < TOTAL / NT; j++) {
		const int id = t + NT * j;
		const int sr = id / V, c8 = id - sr * V;
		const float4 lo = *(const float4*)(cs + sr * PITCH + 8 * c8), hi = *(const float4*)(cs + sr * PITCH + 8 * c8 + 4);
		epi.store8(m0 + row_of(sr), n0 + 8 * c8, lo, hi);
	}
}
template <int NT, int ROWS, int BN, class EPI, class ROWOF>
__device__ __forceinline__ void epi_flush_rows(const float* const cs, const EPI& epi, const int m0, const int n0, const int t, const ROWOF& row_of)
{
	if constexpr (epi_has_store8<EPI>::value) {
		if (epi.vec == 2) { epi_flush_rows8<NT, ROWS, BN>(cs, epi, m0, n0, t, row_of); return; }
	}
	constexpr int PITCH = BN + 8, V = BN / 4, TOTAL = ROWS * V;
	static_assert(TOTAL % NT == 0, "whole vectors per thread");
	// (EPI::FLUSH_UNROLL vectors in flight per thread -- two where the epilogue loads a bias / the old value, one for the plain slab stores: fully unrolled, the eight read-backs of a 128-column slice and their bias / old-value loads cost ~20 more
	// VGPRs than the K loop needs and the half-precision kernels drop from three waves per SIMD to two -- measured slower on the 3 x 3 layers)
#pragma unroll EPI::FLUSH_UNROLL
	for (int j = 0; j < TOTAL / NT; j++) {
		const int id = t + NT * j;
		const int sr = id / V, c4 = id - sr * V;
		const float4 v = *(const float4*)(cs + sr * PITCH + 4 * c4);
		epi.store4(m0 + row_of(sr), n0 + 4 * c4, v);
	}
}

// ---------------------------------------------------------------------------------------------- kernel
// The NCH chunks (of 4 floats) one thread stages for an operand tile of ROWS = 32 * NCH rows at K offset kbase.
//   KCONTIG:  chunk id = t + 256*jj -> row = id >> 3 (differs per jj), k chunk = (id & 7) * 4 (same for all jj)
//   !KCONTIG: chunk id = t + 256*jj -> k = id / (ROWS/4) (differs per jj), row chunk = (id % (ROWS/4)) * 4 (same for all jj)
template <class L, int NCH>
struct TileFetch {
	static constexpr int ROWS = NCH * 32;
	static constexpr int NCTX = L::KCONTIG ? NCH : 1;
	typename L::Ctx ctx[NCTX];
	int koff[NCH];
	long off[L::VECTOR ? NCH : 1]; // VECTOR: the chunk offsets prep_*() resolved for the tile issue() will load
	static constexpr int NKC = L::VECTOR ? (L::KCONTIG ? 1 : NCH) : 1;
	typename L::KCtx kcs[NKC];     // VECTOR: the k state of the tile being prepared: one shared by all chunks (KCONTIG) or one per chunk
	int kb, kl;
	__device__ __forceinline__ void init(const L& l, int row0, int t)
	{
#pragma unroll
		for (int jj = 0; jj < NCH; jj++) {
			const int id = t + GEMM_THREADS * jj;
			if (L::KCONTIG) { ctx[jj % NCTX] = l.make(row0 + (id >> 3)); koff[jj] = (id & 7) << 2; }
			else { if (jj == 0) ctx[0] = l.make(row0 + ((id % (ROWS / 4)) << 2)); koff[jj] = id / (ROWS / 4); }
		}
	}
	// Address phase of the tile at K offset kbase: pure integer VALU, cut into 1 + NCH pieces so the kernel can slot
	// them between the MFMAs of the previous tile.  prep_k first, then prep_chunk(jj) in any order.
	// FIRST = the tile's k states are computed from scratch (multiply-shift divisions); otherwise INCR loaders step the
	// states of the previously prepared tile by one K-step (the caller prepares tiles strictly in order).
	template <bool FIRST>
	__device__ __forceinline__ void prep_k(const L& l, int kbase, int klimit)
	{
		NNC_PIN_S(kbase);
		kb = kbase; kl = klimit;
		if (L::VECTOR && L::KCONTIG) {
			if (L::INCR && !FIRST) l.advance(kcs[0], klimit);
			else kcs[0] = l.kctx(kbase + koff[0], klimit);
		}
	}
	template <bool FIRST>
	__device__ __forceinline__ void prep_chunk(const L& l, const int jj)
	{
		if (!L::VECTOR) return;
		if (L::KCONTIG) { l.pin(ctx[jj % NCTX]); off[L::VECTOR ? jj : 0] = l.offset(ctx[jj % NCTX], kcs[0]); }
		else {
			typename L::KCtx& kc = kcs[L::VECTOR && !L::KCONTIG ? jj : 0];
			if (L::INCR && !FIRST) l.advance(kc, kl);
			else { NNC_PIN_V(koff[jj]); kc = l.kctx(kb + koff[jj], kl); }
			off[L::VECTOR ? jj : 0] = l.offset(ctx[0], kc);
		}
	}
	template <bool FIRST>
	__device__ __forceinline__ void prep(const L& l, int kbase, int klimit)
	{
		prep_k<FIRST>(l, kbase, klimit);
#pragma unroll
		for (int jj = 0; jj < NCH; jj++) prep_chunk<FIRST>(l, jj);
	}
	// Load phase of chunk jj of the tile at K offset kbase: VECTOR = ONE 16-byte global load off the offset prep_chunk(jj)
	// resolved for that tile (kbase is not looked at); !VECTOR = address arithmetic + four scalar loads on the spot.
	__device__ __forceinline__ void issue_chunk(const L& l, float4 (&r)[NCH], const int jj, const int kbase, const int klimit) const
	{
		if (L::VECTOR) r[jj] = ld16(l.p + off[L::VECTOR ? jj : 0]);
		else if (L::KCONTIG) r[jj] = l.load(ctx[jj % NCTX], l.kctx(kbase + koff[0], klimit));
		else r[jj] = l.load(ctx[0], l.kctx(kbase + koff[jj], klimit));
	}
	__device__ __forceinline__ void issue(const L& l, float4 (&r)[NCH], const int kbase, const int klimit) const
	{
#pragma unroll
		for (int jj = 0; jj < NCH; jj++) issue_chunk(l, r, jj, kbase, klimit);
	}
	__device__ __forceinline__ void store_chunk(float* lds, const float4 (&r)[NCH], int t, const int jj) const
	{
		const int id = t + GEMM_THREADS * jj;
		if (L::KCONTIG) *(float4*)(lds + (id >> 3) * GEMM_LDK + ((id & 7) << 2)) = r[jj];
		else *(float4*)(lds + (id / (ROWS / 4)) * ROWS + ((id % (ROWS / 4)) << 2)) = r[jj];
	}
	__device__ __forceinline__ void store(float* lds, const float4 (&r)[NCH], int t) const
	{
#pragma unroll
		for (int jj = 0; jj < NCH; jj++) store_chunk(lds, r, t, jj);
	}
};

// ---- plain matrices through BUFFER loads --------------------------------------------------------------------------------------------------
// Next to fp32 MFMAs every VALU instruction of the wave costs matrix-pipe time (tools/coissue2_probe.cpp on the MI355X: the first VALU behind an MFMA
// ~14 clocks, each further one ~4; integer or float, scalar or packed alike), and the pointer path above spends ~5 VALU per 16-byte chunk per K-step on
// 64-bit address arithmetic and the zero-page select (85-190 VALU per 128 MFMAs in the Winograd GEMMs).  A plain matrix needs none of it:
// element(r, k) = p[r * ldr + k] (KC) or p[k * ldk + r] (!KC) is a per-lane byte offset that never changes (row part) plus a wave-uniform one
// (the K-step's k: an SGPR, the instruction's soffset), and rows beyond R are an out-of-range offset the buffer's range check turns into zeros.
// Conditions (gemm_run checks them): 16-byte chunks (the VEC conditions), K and the K-slices whole K-steps (no k tail to mask), R % 4 == 0 for !KC,
// offsets inside 31 bits.  The descriptor's base is the block tile's first row; its range ends with the matrix.
template <bool KC>
struct BufMatLoader {
	static constexpr bool KCONTIG = KC, VECTOR = true, INCR = false;
	const float* p;
	long zoff; // (unused: gemm_run sets it on every loader)
	long ldr, ldk;
	int R, K;
	void finish() {}
};
template <class L> struct is_buffer_loader { static constexpr bool value = false; };
template <bool KC> struct is_buffer_loader<BufMatLoader<KC>> { static constexpr bool value = true; };

template <class L, int NCH>
struct TileFetchBuf {
	static constexpr int ROWS = NCH * 32;
	typedef unsigned int u4 __attribute__((ext_vector_type(4)));
	__amdgpu_buffer_rsrc_t rs;
	unsigned voff[NCH];
	unsigned kscale; // bytes per unit of k
	__device__ __forceinline__ void init(const L& l, int row0, int t)
	{
		const long extent = L::KCONTIG ? (long)(l.R - 1) * l.ldr + l.K : (long)(l.K - 1) * l.ldk + l.R; // floats from l.p to the end of the matrix
		const long base = L::KCONTIG ? (long)row0 * l.ldr : (long)row0;
		const long left = (extent - base) * 4;
		rs = __builtin_amdgcn_make_buffer_rsrc((void*)(l.p + base), 0, (unsigned)(left > 0x7fffffffL ? 0x7fffffffL : (left < 0 ? 0 : left)), 0x00020000);
		kscale = L::KCONTIG ? 4u : (unsigned)l.ldk * 4u;
#pragma unroll
		for (int jj = 0; jj < NCH; jj++) {
			const int id = t + GEMM_THREADS * jj;
			if (L::KCONTIG) {
				const int r = id >> 3;
				voff[jj] = row0 + r < l.R ? (unsigned)r * (unsigned)l.ldr * 4u + (unsigned)((id & 7) << 4) : 0x80000000u;
			} else {
				const int r = (id % (ROWS / 4)) << 2, k = id / (ROWS / 4);
				voff[jj] = row0 + r < l.R ? (unsigned)k * (unsigned)l.ldk * 4u + (unsigned)r * 4u : 0x80000000u;
			}
		}
	}
	template <bool FIRST> __device__ __forceinline__ void prep_k(const L&, int, int) {}
	template <bool FIRST> __device__ __forceinline__ void prep_chunk(const L&, const int) {}
	template <bool FIRST> __device__ __forceinline__ void prep(const L&, int, int) {}
	__device__ __forceinline__ void issue_chunk(const L&, float4 (&r)[NCH], const int jj, const int kbase, const int) const
	{
		const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[jj], (unsigned)kbase * kscale, 0);
		r[jj] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
	}
	__device__ __forceinline__ void issue(const L& l, float4 (&r)[NCH], const int kbase, const int klimit) const
	{
#pragma unroll
		for (int jj = 0; jj < NCH; jj++) issue_chunk(l, r, jj, kbase, klimit);
	}
	__device__ __forceinline__ void store_chunk(float* lds, const float4 (&r)[NCH], int t, const int jj) const
	{
		const int id = t + GEMM_THREADS * jj;
		if (L::KCONTIG) *(float4*)(lds + (id >> 3) * GEMM_LDK + ((id & 7) << 2)) = r[jj];
		else *(float4*)(lds + (id / (ROWS / 4)) * ROWS + ((id % (ROWS / 4)) << 2)) = r[jj];
	}
	__device__ __forceinline__ void store(float* lds, const float4 (&r)[NCH], int t) const
	{
#pragma unroll
		for (int jj = 0; jj < NCH; jj++) store_chunk(lds, r, t, jj);
	}
};
template <class L, int NCH, bool BUF = is_buffer_loader<L>::value> struct FetchOf { typedef TileFetch<L, NCH> type; };
template <class L, int NCH> struct FetchOf<L, NCH, true> { typedef TileFetchBuf<L, NCH> type; };

// Fragment reads of quarter q (k = 8q .. 8q+7) of a K-step for the W 32-row MFMA tiles of a wave whose rows start at `base`.
// Which row of the wave's span an MFMA tile's lane stands for is free as long as the epilogue agrees (frag_row below):
//   KC  ([row][k] image): tile ti, lane li = row 32 * ti + li; its four k values are ONE ds_read_b128
//   !KC ([k][row] image): tile ti, lane li = row W * li + ti, so that the W tiles' values of one k are adjacent in LDS and
//        come in ONE ds_read_b32 * W (b64 for the 2-tile wave) instead of W scalar reads -- the fragment reads of a
//        row-contiguous operand are 4 * W per quarter otherwise, and every non-MFMA instruction costs issue cycles.
template <bool KC, int W>
__device__ __forceinline__ int frag_row(const int ti, const int li) { return KC ? 32 * ti + li : W * li + ti; }
template <bool KC, int W, int ROWS>
__device__ __forceinline__ void read_frags(const float* s, const int base, const int li, const int q, const int lh, float (&f)[W][4])
{
	if (KC) {
#pragma unroll
		for (int ti = 0; ti < W; ti++) {
			const float4 v = *(const float4*)(s + (base + ti * 32 + li) * GEMM_LDK + 8 * q + 4 * lh);
			f[ti][0] = v.x; f[ti][1] = v.y; f[ti][2] = v.z; f[ti][3] = v.w;
		}
	} else {
		typedef float vecw __attribute__((ext_vector_type(W)));
#pragma unroll
		for (int e = 0; e < 4; e++) {
			const float* const a = s + (8 * q + 4 * lh + e) * ROWS + base + W * li;
			if (W == 1) f[0][e] = *a;
			else {
				const vecw v = *(const vecw*)a;
#pragma unroll
				for (int ti = 0; ti < W; ti++) f[ti][e] = v[ti];
			}
		}
	}
}

template <int G> struct GroupId { static constexpr int value = G; };
struct NoSideWork { template <class G> __device__ __forceinline__ void operator()(G) const {} };

// One K-step (32 deep) of the wave's WM x WN tiles out of the LDS images sa / sb: 16 groups of WM * WN MFMAs (one group
// per k pair), hand-ordered.  A wave's MFMA occupies the SIMD's matrix pipe for 64 cycles while the wave itself is free to
// issue other instructions, so everything else the K-step needs is slotted BETWEEN the groups and fenced there
// (sched_barrier) so hipcc can neither hoist it in front of the first MFMA nor sink it behind the last one:
//   * the fragment ds_reads of quarter q+1 go out with the first group of quarter q (double-buffered fragment registers)
//   * side(GroupId<g>) carries the caller's other work for group g = 0..15 (address VALU of tile kt+2, LDS writes of
//     tile kt+1): what remains outside the MFMA stream of a K-step is the barrier and the first fragment read.
template <bool AKC, bool BKC, int WM, int WN, bool NO_MFMA, class SIDE>
__device__ __forceinline__ void mfma_kstep(const float* sa, const float* sb, const int row_a, const int col_b, const int li, const int lh, floatx16 (&acc)[WM][WN], const SIDE& side)
{
	constexpr int BM = 64 * WM, BN = 64 * WN;
	float fa_[2][WM][4], fb_[2][WN][4];
	read_frags<AKC, WM, BM>(sa, row_a, li, 0, lh, fa_[0]);
	read_frags<BKC, WN, BN>(sb, col_b, li, 0, lh, fb_[0]);
#pragma unroll
	for (int q = 0; q < 4; q++) {
		if (q < 3) {
			read_frags<AKC, WM, BM>(sa, row_a, li, q + 1, lh, fa_[(q + 1) & 1]);
			read_frags<BKC, WN, BN>(sb, col_b, li, q + 1, lh, fb_[(q + 1) & 1]);
		}
#pragma unroll
		for (int e = 0; e < 4; e++) {
#pragma unroll
			for (int ti = 0; ti < WM; ti++)
#pragma unroll
				for (int tj = 0; tj < WN; tj++)
					if (NO_MFMA) acc[ti][tj][0] += fa_[q & 1][ti][e] * fb_[q & 1][tj][e];
					else acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_[q & 1][ti][e], fb_[q & 1][tj][e], acc[ti][tj], 0, 0, 0);
			if (q == 0 && e == 0) side(GroupId<0>()); else if (q == 0 && e == 1) side(GroupId<1>()); else if (q == 0 && e == 2) side(GroupId<2>()); else if (q == 0 && e == 3) side(GroupId<3>());
			else if (q == 1 && e == 0) side(GroupId<4>()); else if (q == 1 && e == 1) side(GroupId<5>()); else if (q == 1 && e == 2) side(GroupId<6>()); else if (q == 1 && e == 3) side(GroupId<7>());
			else if (q == 2 && e == 0) side(GroupId<8>()); else if (q == 2 && e == 1) side(GroupId<9>()); else if (q == 2 && e == 2) side(GroupId<10>()); else if (q == 2 && e == 3) side(GroupId<11>());
			else if (q == 3 && e == 0) side(GroupId<12>()); else if (q == 3 && e == 1) side(GroupId<13>()); else if (q == 3 && e == 2) side(GroupId<14>()); else side(GroupId<15>());
			__builtin_amdgcn_sched_barrier(0);
		}
	}
}

__device__ __forceinline__ long M_N_slab(const EpiStore&) { return 0; }
__device__ __forceinline__ long M_N_slab(const EpiPartial& e) { return e.slab; }

// grid: x = tiles (* split-K slices), XCD-swizzled; z = batch / conv group.  WM / WN = 32x32 MFMA tiles per wave.
// DBG (tools/kprobe.cpp only; the library always instantiates DBG = 0): knock out parts of the steady state to attribute time.
//   1 no global loads, 2 no LDS writes, 4 no barrier, 8 no address prep, 16 no MFMAs, 32 no A loads, 64 no B loads

// Batched launches without split-K (`splits` < 0: the batch has -splits entries and the grid is ONE dimension of 8 * ceil(entries / 8) * tiles workgroups).
// Workgroup b runs on XCD b % 8 (round-robin dispatch); XCD x walks the entries x, x + 8, ... tile by tile, so the tiles of one entry -- which share its operands:
// every row block of a 1 x 1 convolution's output reads the same image planes -- meet in ONE L2 and the entry leaves HBM once.  (With the entries on grid z the
// tiles of an entry were dealt over all eight XCDs: ResNet-50's 14^2 layers, 16 tiles per image, fetched each image's planes up to eight times --
// tools/store_probe.cpp, profiles/r06_v14_store_probe.txt.)  false: no entry for this workgroup (the last round of eight).
__device__ __forceinline__ bool gemm_batch_xcd_map(const int bid, const int tiles, const int entries, int* const tile, int* const z)
{
	const int xcd = bid & 7, idx = bid >> 3;
	const int round = idx / tiles;
	*tile = idx - round * tiles;
	*z = round * 8 + xcd;
	return *z < entries;
}
template <class LA, class LB, class EPI, int WM, int WN, int DBG = 0>
__global__ void __launch_bounds__(GEMM_THREADS) mfma_gemm_f32_kernel(LA la, LB lb, EPI epi, const int tiles_m, const int tiles_n, const int K, const int k_per_split, const int splits, const long a_zoff, const long b_zoff, const long c_zoff, const long bias_zoff, const KOrder ko)
{
	constexpr int BM = 64 * WM, BN = 64 * WN;
	constexpr int A_FLOATS = LA::KCONTIG ? BM * GEMM_LDK : GEMM_BK * BM;
	constexpr int B_FLOATS = LB::KCONTIG ? BN * GEMM_LDK : GEMM_BK * BN;
	__shared__ __attribute__((aligned(16))) float lds[2][A_FLOATS + B_FLOATS]; // [buffer][A | B]
	const int t = threadIdx.x;
	const int lane = t & 63, wave = t >> 6;
	const int wm = wave >> 1, wn = wave & 1;
	const int li = lane & 31, lh = lane >> 5;
	// XCD-aware, bijective remap of the linear workgroup id (cdna_hip_programming.md T1): workgroup b runs on XCD b % 8.
	//   no split-K : each XCD gets a contiguous run of tiles (neighbours share im2col halos / weight panels in its L2)
	//   split-K    : (grid = tiles * splits, splits % 8 == 0) each XCD gets WHOLE K-slices -- slice s = xcd + 8 * j with all
	//                of its tiles resident together, so the slice's pixel range of both operands streams through that
	//                XCD's L2 once while every tile of the slice reads it in lockstep (wgrad: 36-144 tiles per slice)
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
	la.p += (long)zi * a_zoff; la.zoff -= (long)zi * a_zoff;
	lb.p += (long)zi * b_zoff; lb.zoff -= (long)zi * b_zoff;
	epi.c += (long)zi * c_zoff;
	if (splits > 1) epi.c += (long)slice * M_N_slab(epi);
	const int k_begin = slice * k_per_split;
	const int k_end = (k_begin + k_per_split < K) ? k_begin + k_per_split : K;
	const int nk = (k_end - k_begin + GEMM_BK - 1) / GEMM_BK;

	// K offset (wave-uniform, a multiple of GEMM_BK) of the loop's position kb -> the operands' k (see KOrder)
	auto kmap = [&](const int kb) -> int {
		if (!ko.taps) return kb;
		if (kb >= k_end) return ko.K; // past-the-end prefetch: fully masked
		const int s = kb / GEMM_BK;
		const int cc = ko.d.div(s);
		return (s - cc * ko.taps) * ko.C + cc * GEMM_BK;
	};
	const int klim = ko.taps ? ko.K : k_end;
	typename FetchOf<LA, WM * 2>::type fa;
	typename FetchOf<LB, WN * 2>::type fb;
	fa.init(la, m0, t);
	fb.init(lb, n0, t);
	floatx16 acc[WM][WN];
#pragma unroll
	for (int i = 0; i < WM; i++)
#pragma unroll
		for (int j = 0; j < WN; j++)
#pragma unroll
			for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

	const int row_a = wm * (32 * WM), col_b = wn * (32 * WN); // first row / column of the wave's span inside the block tile
	// Software pipeline, prefetch distance TWO tiles: while tile kt is multiplied out of LDS buffer kt & 1, the chunks of
	// tile kt+1 (register set (kt+1) & 1, loaded during the previous K-step) are written to the other LDS buffer, the
	// chunks of tile kt+2 are loaded into register set kt & 1, and the addresses of tile kt+3 are computed -- every one
	// of these instructions is issued BETWEEN two MFMA groups, at most one global load per group:
	//   * a wave that issues its 8 loads back to back queues behind the other 7 waves' bursts in the CU's address
	//     pipeline and stalls at the issue for ~1000 cycles per K-step, MFMAs blocked behind it (measured: 121 vs 146
	//     TFLOP/s with the loads knocked out, independent of the prefetch distance); one load per >= 256 MFMA-cycles
	//     never finds the queue occupied;
	//   * a load has a full K-step (>= 4000 cycles) to land before its LDS write waits for it.
	constexpr int NA = WM * 2, NB = WN * 2;
	float4 ra[2][NA], rb[2][NB];
	if (nk > 0) {
		const int k0 = kmap(k_begin), k1 = kmap(k_begin + GEMM_BK), k2 = kmap(k_begin + 2 * GEMM_BK);
		fa.template prep<true>(la, k0, klim);
		fb.template prep<true>(lb, k0, klim);
		fa.issue(la, ra[0], k0, klim);
		fb.issue(lb, rb[0], k0, klim);
		fa.template prep<false>(la, k1, klim);
		fb.template prep<false>(lb, k1, klim);
		fa.issue(la, ra[1], k1, klim); // past-the-end tiles are fully masked: they load the page of zeros
		fb.issue(lb, rb[1], k1, klim);
		fa.store(lds[0], ra[0], t);
		fb.store(lds[0] + A_FLOATS, rb[0], t);
		fa.template prep<false>(la, k2, klim);
		fb.template prep<false>(lb, k2, klim);
	}
	__syncthreads();
	// One steady-state K-step (S = kt & 1, a compile-time constant so the register sets stay in fixed registers).
	// Side work of MFMA group g (operand A in groups 0-7(8), operand B in groups 8(7)-15; chunk jj of an operand with NCH chunks
	// owns the group pair 2 * jj * 4 / NCH):
	//   even group of the pair:  load chunk jj of tile kt+2 into set S          (uses the offset resolved one K-step ago)
	//   odd group of the pair:   LDS-write chunk jj of tile kt+1 from set S^1, then resolve chunk jj's offset of tile kt+3
	//   group 0 / 8 additionally: the per-K-step k state of tile kt+3 (prep_k)
	auto kstep = [&](auto sid, const int kt) {
		constexpr int S = decltype(sid)::value;
		float* const da = lds[S ^ 1];
		float* const db = da + A_FLOATS;
		const int kb2 = kmap(k_begin + (kt + 2) * GEMM_BK), kb3 = kmap(k_begin + (kt + 3) * GEMM_BK);
		auto side = [&](auto gid) {
			constexpr int g = decltype(gid)::value;
			constexpr int GB = NB == 8 ? 7 : 8; // eight B chunks: loads in groups 7-14, LDS writes in 8-15
#pragma unroll
			for (int jj = 0; jj < NA; jj++) {
				if (g == jj * 8 / NA && !(DBG & 1) && !(DBG & 32)) fa.issue_chunk(la, ra[S], jj, kb2, klim);
				if (g == jj * 8 / NA + 1) {
					if (!(DBG & 2)) fa.store_chunk(da, ra[S ^ 1], t, jj);
					if (!(DBG & 8)) fa.template prep_chunk<false>(la, jj);
				}
			}
#pragma unroll
			for (int jj = 0; jj < NB; jj++) {
				if (g == GB + jj * 8 / NB && !(DBG & 1) && !(DBG & 64)) fb.issue_chunk(lb, rb[S], jj, kb2, klim);
				if (g == GB + jj * 8 / NB + 1) {
					if (!(DBG & 2)) fb.store_chunk(db, rb[S ^ 1], t, jj);
					if (!(DBG & 8)) fb.template prep_chunk<false>(lb, jj);
				}
			}
			if (g == 0 && !(DBG & 8)) fa.template prep_k<false>(la, kb3, klim);
			if (g == GB && !(DBG & 8)) fb.template prep_k<false>(lb, kb3, klim);
		};
		mfma_kstep<LA::KCONTIG, LB::KCONTIG, WM, WN, (DBG & 16) != 0>(lds[S], lds[S] + A_FLOATS, row_a, col_b, li, lh, acc, side);
		if (!(DBG & 4)) __syncthreads();
	};
	{ // pairs of K-steps with no control flow between them (a branch in the middle makes hipcc copy the in-flight register
	  // set at the merge, and a copy of a register a load is still writing costs an s_waitcnt vmcnt(0))
		int kt = 0;
		for (; kt + 2 < nk; kt += 2) {
			kstep(GroupId<0>(), kt);
			kstep(GroupId<1>(), kt + 1);
		}
		if (kt + 1 < nk) kstep(GroupId<0>(), kt);
	}
	if (nk > 0) {
		const int cur = (nk - 1) & 1;
		mfma_kstep<LA::KCONTIG, LB::KCONTIG, WM, WN, false>(lds[cur], lds[cur] + A_FLOATS, row_a, col_b, li, lh, acc, NoSideWork());
	}
	// D layout of 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), both in fragment-lane terms
	// (frag_row maps a tile's lane to the row / column of the wave's span it stands for).
	if (epi.bias) epi.bias += (long)zi * bias_zoff;
	if (epi.vec) {
		// through LDS, one tile row of every wave per pass: 64 staged rows (wave row wm, fragment row q) x BN columns, read back row-major (epi_flush_rows)
		constexpr int PITCH = BN + 8;
		static_assert(64 * PITCH <= 2 * (A_FLOATS + B_FLOATS), "the staged slice fits the operand buffers");
		float* const cs = &lds[0][0];
#pragma unroll
		for (int ti = 0; ti < WM; ti++) {
			__syncthreads(); // every wave is done with the operand images (first pass) / with reading the previous slice
#pragma unroll
			for (int tj = 0; tj < WN; tj++) {
				const int col = col_b + frag_row<LB::KCONTIG, WN>(tj, li);
#pragma unroll
				for (int r = 0; r < 16; r++) cs[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * PITCH + col] = acc[ti][tj][r];
			}
			__syncthreads();
			epi_flush_rows<GEMM_THREADS, 64, BN>(cs, epi, m0, n0, t, [&](const int sr) { return (sr >> 5) * (32 * WM) + frag_row<LA::KCONTIG, WM>(ti, sr & 31); });
		}
		return;
	}
#pragma unroll
	for (int ti = 0; ti < WM; ti++)
#pragma unroll
		for (int tj = 0; tj < WN; tj++) {
			const int n = n0 + col_b + frag_row<LB::KCONTIG, WN>(tj, li);
#pragma unroll
			for (int r = 0; r < 16; r++) {
				const int m = m0 + row_a + frag_row<LA::KCONTIG, WM>(ti, (r & 3) + 8 * (r >> 2) + 4 * lh);
				epi(m, n, acc[ti][tj][r]);
			}
		}
}

// Finish a split-K contraction: c = alpha * sum_s slab[s] (+ bias[n]) (+ old c).  Fixed summation order => deterministic.
// grid.y = batch entry z: its slab set starts at ws + z * splits * slab, its output at c + z * c_zoff.  T = float, or _Float16 for a half-precision result.
// A workgroup takes 64 consecutive outputs x 4 phases: wave `ph` sums the slabs ph, ph + 4, ... (four independent loads in flight), the phases meet in LDS
// as (p0 + p1) + (p2 + p3).  Round 3's form -- one thread per output walking all `splits` slabs, one dependent load-add chain of up to 512 -- took 56 us per
// call on ResNet-50's filter gradients (86 calls per step: small outputs, hundreds of slices), as long as some of the contractions it finishes.
template <class T>
static __global__ void __launch_bounds__(256) splitk_reduce_kernel(const float* ws, const int splits, const long slab, T* c, const long ldm, const long ldn, const T* bias, const long bias_ldm, const float alpha, const int accumulate, const int M, const int N, const long c_zoff, const long bias_zoff, const long bias_ldn)
{
	__shared__ float red[4][64];
	ws += (long)blockIdx.y * splits * slab;
	c += (long)blockIdx.y * c_zoff;
	if (bias) bias += (long)blockIdx.y * bias_zoff;
	const int lane = threadIdx.x & 63, ph = threadIdx.x >> 6;
	for (long base = (long)blockIdx.x * 64; base < slab; base += (long)gridDim.x * 64) {
		const long idx = base + lane;
		float v = 0.f;
		if (idx < slab) {
			const float* const p = ws + idx;
			int s = ph;
			for (; s + 12 < splits; s += 16) {
				const float a0 = p[(long)s * slab], a1 = p[(long)(s + 4) * slab], a2 = p[(long)(s + 8) * slab], a3 = p[(long)(s + 12) * slab];
				v += a0; v += a1; v += a2; v += a3;
			}
			for (; s < splits; s += 4) v += p[(long)s * slab];
		}
		red[ph][lane] = v;
		__syncthreads();
		if (ph == 0 && idx < slab) {
			v = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
			const int m = (int)(idx / N), n = (int)(idx - (long)m * N);
			v *= alpha;
			if (bias) v += (float)bias[(long)m * bias_ldm + (long)n * bias_ldn];
			const long o = (long)m * ldm + (long)n * ldn;
			if (accumulate) v += (float)c[o];
			c[o] = (T)v;
		}
		__syncthreads();
	}
}

} // namespace nnc
