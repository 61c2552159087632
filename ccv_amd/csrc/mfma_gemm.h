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
// K-step; the global loads of tile t+1 are issued before the MFMAs of tile t and written to LDS after them).
// Gathers are BRANCH-FREE: every lane always issues its 16-byte load (address clamped to the operand base when the
// element is padding / out of range) and selects zero afterwards, so the compiler can batch all loads of a K-step in
// front of the MFMA block; integer divisions by runtime geometry use precomputed multiply-shift constants.
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

namespace nnc {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int GEMM_BK = 32, GEMM_THREADS = 256;
constexpr int GEMM_LDK = 36; // row stride of a k-contiguous LDS image

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4sel(bool ok, float4 v) { return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f); }

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

// ---------------------------------------------------------------------------------------------- loaders
// Concept:
//   static constexpr bool KCONTIG;  const float* p;
//   Ctx  make(int r) const;                 per-row state (KCONTIG) or per-row-chunk state (!KCONTIG), computed once
//   KCtx kctx(int k, int klimit) const;     per-k state, computed once per K-step (KCONTIG) / per chunk (!KCONTIG)
//   float4 load(const Ctx&, const KCtx&) const;
//     KCONTIG:  elements (r, k..k+3)   (k multiple of 4);   !KCONTIG: elements (r..r+3, k)   (r multiple of 4)
// Rows >= R, k >= klimit and im2col padding read as zero.  VEC = one 16-byte global load per chunk (needs the
// alignment / divisibility the host checks before picking it); !VEC = four guarded scalar loads.

// Plain matrix: element(r, k) = p[r * ldr + k * ldk].  KC => ldk == 1, otherwise ldr == 1.
template <bool KC, bool VEC>
struct MatLoader {
	static constexpr bool KCONTIG = KC;
	const float* p;
	long ldr, ldk;
	int R, K;
	void finish() {}
	struct Ctx { long off; int r; };
	struct KCtx { long off; int k, klimit; };
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
	__device__ __forceinline__ float4 load(const Ctx& c, const KCtx& x) const
	{
		const bool ok = c.r < R && x.k < x.klimit;
		if (VEC) return f4sel(ok, *(const float4*)(p + (ok ? c.off + x.off : 0)));
		float v[4];
#pragma unroll
		for (int e = 0; e < 4; e++) {
			const bool oke = ok && (KC ? x.k + e < x.klimit : c.r + e < R);
			const float u = p[oke ? c.off + x.off + e : 0];
			v[e] = oke ? u : 0.f;
		}
		return make_float4(v[0], v[1], v[2], v[3]);
	}
};

// im2col gather with the reduction index running (tap_y, tap_x, channel), channel fastest: rows are output
// pixels m = (n, oy, ox).  Serves conv forward (source = a) and conv dgrad (source = g, taps walked backwards).
//   t_y = oy * my + oy_off + i * ty ;  source y = t_y / dv_y, valid iff t_y >= 0, t_y % dv_y == 0, y < H   (same for x)
//   forward: my = stride, oy_off = -border, ty = +dilation, dv = 1
//   dgrad:   my = 1, oy_off = +border, ty = -dilation, dv = stride
// VEC: one 16-byte load per chunk (C % 4 == 0 keeps a chunk inside one tap); !VEC: the four k's of a chunk are resolved
// one by one (they may straddle taps: conv1_1 has C = 3).
template <bool VEC>
struct Im2colKC {
	static constexpr bool KCONTIG = true;
	const float* p;
	long s_n;
	int s_h, s_w;
	int H, W;
	int OW, OHW, M;
	int C, KWC, K;
	int my, mx, oy_off, ox_off, ty, tx, dv_y, dv_x;
	FastDiv d_ohw, d_ow, d_kwc, d_c, d_dvy, d_dvx;
	void finish() { d_ohw.init(OHW); d_ow.init(OW); d_kwc.init(KWC); d_c.init(C); d_dvy.init(dv_y); d_dvx.init(dv_x); }
	struct Ctx { long base; int iy0, ix0; };
	struct K1 { int dy, dx, ch; bool ok; };
	struct KCtx { K1 e[VEC ? 1 : 4]; };
	__device__ __forceinline__ Ctx make(int m) const
	{
		Ctx c;
		if (m >= M) { c.base = 0; c.iy0 = -(1 << 28); c.ix0 = -(1 << 28); return c; } // every tap fails the y >= 0 test
		const int n = d_ohw.div(m);
		const int rem = m - n * OHW;
		const int oy = d_ow.div(rem);
		const int ox = rem - oy * OW;
		c.base = (long)n * s_n;
		c.iy0 = oy * my + oy_off;
		c.ix0 = ox * mx + ox_off;
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
		x.ch = r - j * C;
		x.dy = i * ty;
		x.dx = j * tx;
		return x;
	}
	__device__ __forceinline__ KCtx kctx(int k, int klimit) const
	{
		KCtx x;
#pragma unroll
		for (int e = 0; e < (VEC ? 1 : 4); e++) x.e[e] = k1(k + e, klimit);
		return x;
	}
	__device__ __forceinline__ bool locate(const Ctx& c, const K1& x, long& off) const
	{
		int y = c.iy0 + x.dy, xx = c.ix0 + x.dx;
		bool ok = x.ok && y >= 0 && xx >= 0;
		if (dv_y != 1 || dv_x != 1) { // wave-uniform: strided dgrad only
			const int yy = ok ? y : 0, xq = ok ? xx : 0;
			const int qy = d_dvy.div(yy), qx = d_dvx.div(xq);
			ok = ok && qy * dv_y == yy && qx * dv_x == xq;
			y = qy; xx = qx;
		}
		ok = ok && y < H && xx < W;
		off = ok ? c.base + (long)(y * s_h + xx * s_w + x.ch) : 0; // one image spans < 2^31 elements (host-checked)
		return ok;
	}
	__device__ __forceinline__ float4 load(const Ctx& c, const KCtx& x) const
	{
		if (VEC) {
			long off;
			const bool ok = locate(c, x.e[0], off);
			return f4sel(ok, *(const float4*)(p + off));
		}
		float v[4];
#pragma unroll
		for (int e = 0; e < (VEC ? 1 : 4); e++) {
			long off;
			const bool ok = locate(c, x.e[e], off);
			const float u = p[off];
			v[e] = ok ? u : 0.f;
		}
		return make_float4(v[0], v[1], v[2], v[3]);
	}
};

// conv dgrad weights: B(k = (tap, ko), n = c) = w[ko][tap][c]  (n contiguous).
template <bool VEC>
struct WgtDgradNC {
	static constexpr bool KCONTIG = false;
	const float* p;
	long ko_stride; // kh*kw*C
	int C, Ko, K;   // K = kh*kw*Ko
	FastDiv d_ko;
	void finish() { d_ko.init(Ko); }
	struct Ctx { int c; };
	struct KCtx { long off; bool ok; };
	__device__ __forceinline__ Ctx make(int c) const { Ctx x; x.c = c; return x; }
	__device__ __forceinline__ KCtx kctx(int k, int klimit) const
	{
		KCtx x;
		x.ok = k < klimit;
		const int kk = x.ok ? k : 0;
		const int tap = d_ko.div(kk);
		const int ko = kk - tap * Ko;
		x.off = (long)ko * ko_stride + (long)tap * C;
		return x;
	}
	__device__ __forceinline__ float4 load(const Ctx& c, const KCtx& x) const
	{
		const bool ok = x.ok && c.c < C;
		if (VEC) return f4sel(ok, *(const float4*)(p + (ok ? x.off + c.c : 0)));
		float v[4];
#pragma unroll
		for (int e = 0; e < 4; e++) {
			const bool oke = ok && c.c + e < C;
			const float u = p[oke ? x.off + c.c + e : 0];
			v[e] = oke ? u : 0.f;
		}
		return make_float4(v[0], v[1], v[2], v[3]);
	}
};

// conv wgrad activations: B(k = pixel (n, oy, ox), nn = (tap_y, tap_x, c)) = a[n, oy*sy - py + i*dy, ox*sx - px + j*dx, c].
template <bool VEC>
struct Im2colNC {
	static constexpr bool KCONTIG = false;
	const float* p;
	long s_n;
	int s_h, s_w;
	int H, W;
	int OW, OHW;
	int C, KWC, NN, K; // NN = kh*kw*C, K = N*OH*OW
	int sy, sx, py, px, dy, dx;
	FastDiv d_ohw, d_ow;
	void finish() { d_ohw.init(OHW); d_ow.init(OW); }
	struct Ctx { int nn; int off_y[VEC ? 1 : 4], off_x[VEC ? 1 : 4], ch[VEC ? 1 : 4]; }; // per column (i*dy - py, j*dx - px, c)
	struct KCtx { long base; int by, bx; bool ok; };
	__device__ __forceinline__ Ctx make(int nn) const
	{
		Ctx c;
		c.nn = nn;
#pragma unroll
		for (int e = 0; e < (VEC ? 1 : 4); e++) { // VEC: the four columns share a tap and are channel-consecutive
			const int q = nn + e;
			const int i = q / KWC;
			const int r = q - i * KWC;
			const int j = r / C;
			c.off_y[e] = i * dy - py;
			c.off_x[e] = j * dx - px;
			c.ch[e] = r - j * C;
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
		x.base = (long)n * s_n;
		x.by = oy * sy;
		x.bx = ox * sx;
		return x;
	}
	__device__ __forceinline__ float4 load(const Ctx& c, const KCtx& x) const
	{
		if (VEC) {
			const int y = x.by + c.off_y[0], xx = x.bx + c.off_x[0];
			const bool ok = x.ok && c.nn < NN && y >= 0 && xx >= 0 && y < H && xx < W;
			return f4sel(ok, *(const float4*)(p + (ok ? x.base + (long)(y * s_h + xx * s_w + c.ch[0]) : 0)));
		}
		float v[4];
#pragma unroll
		for (int e = 0; e < (VEC ? 1 : 4); e++) {
			const int y = x.by + c.off_y[e], xx = x.bx + c.off_x[e];
			const bool ok = x.ok && c.nn + e < NN && y >= 0 && xx >= 0 && y < H && xx < W;
			const float u = p[ok ? x.base + (long)(y * s_h + xx * s_w + c.ch[e]) : 0];
			v[e] = ok ? u : 0.f;
		}
		return make_float4(v[0], v[1], v[2], v[3]);
	}
};

// ---------------------------------------------------------------------------------------------- epilogues
// Direct store: c[m*ldm + n*ldn] = alpha * acc (+ bias[n]) (+ old c when accumulating).
struct EpiStore {
	float* c;
	long ldm, ldn;
	const float* bias; // per n, may be null
	float alpha;
	int accumulate;
	int M, N;
	__device__ __forceinline__ void operator()(int m, int n, float v) const
	{
		if (m < M && n < N) {
			const long o = (long)m * ldm + (long)n * ldn;
			v *= alpha;
			if (bias) v += bias[n];
			if (accumulate) v += c[o];
			c[o] = v;
		}
	}
};
// Split-K partial: slab[blockIdx.y][m][n] = acc; splitk_reduce_kernel finishes (deterministic order).
struct EpiPartial {
	float* c; // workspace
	const float* bias; // unused (applied by splitk_reduce_kernel); keeps the epilogue concept uniform
	long slab; // M*N
	int M, N;
	__device__ __forceinline__ void operator()(int m, int n, float v) const
	{
		if (m < M && n < N) c[(long)blockIdx.y * slab + (long)m * N + n] = v;
	}
};

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
	__device__ __forceinline__ void init(const L& l, int row0, int t)
	{
#pragma unroll
		for (int jj = 0; jj < NCH; jj++) {
			const int id = t + GEMM_THREADS * jj;
			if (L::KCONTIG) { ctx[jj % NCTX] = l.make(row0 + (id >> 3)); koff[jj] = (id & 7) << 2; }
			else { if (jj == 0) ctx[0] = l.make(row0 + ((id % (ROWS / 4)) << 2)); koff[jj] = id / (ROWS / 4); }
		}
	}
	__device__ __forceinline__ void fetch(const L& l, int kbase, int klimit, float4 (&r)[NCH]) const
	{
		if (L::KCONTIG) {
			const typename L::KCtx kc = l.kctx(kbase + koff[0], klimit);
#pragma unroll
			for (int jj = 0; jj < NCH; jj++) r[jj] = l.load(ctx[jj % NCTX], kc);
		} else {
#pragma unroll
			for (int jj = 0; jj < NCH; jj++) r[jj] = l.load(ctx[0], l.kctx(kbase + koff[jj], klimit));
		}
	}
	__device__ __forceinline__ void store(float* lds, const float4 (&r)[NCH], int t) const
	{
#pragma unroll
		for (int jj = 0; jj < NCH; jj++) {
			const int id = t + GEMM_THREADS * jj;
			if (L::KCONTIG) *(float4*)(lds + (id >> 3) * GEMM_LDK + ((id & 7) << 2)) = r[jj];
			else *(float4*)(lds + (id / (ROWS / 4)) * ROWS + ((id % (ROWS / 4)) << 2)) = r[jj];
		}
	}
};

// grid: x = tiles (XCD-swizzled), y = split-K slices, z = batch / conv group.  WM / WN = 32x32 MFMA tiles per wave.
template <class LA, class LB, class EPI, int WM, int WN>
__global__ void __launch_bounds__(GEMM_THREADS) mfma_gemm_f32_kernel(LA la, LB lb, EPI epi, const int tiles_m, const int tiles_n, const int K, const int k_per_split, const long a_zoff, const long b_zoff, const long c_zoff, const long bias_zoff)
{
	constexpr int BM = 64 * WM, BN = 64 * WN;
	constexpr int A_FLOATS = LA::KCONTIG ? BM * GEMM_LDK : GEMM_BK * BM;
	constexpr int B_FLOATS = LB::KCONTIG ? BN * GEMM_LDK : GEMM_BK * BN;
	__shared__ __attribute__((aligned(16))) float lds[2][A_FLOATS + B_FLOATS]; // [buffer][A | B]
	const int t = threadIdx.x;
	const int lane = t & 63, wave = t >> 6;
	const int wm = wave >> 1, wn = wave & 1;
	const int li = lane & 31, lh = lane >> 5;
	// XCD-aware, bijective remap of the linear workgroup id (cdna_hip_programming.md T1).
	const int nwg = gridDim.x;
	const int bid = blockIdx.x;
	int tile;
	{
		const int xcd = bid & 7, idx = bid >> 3;
		const int q = nwg >> 3, r = nwg & 7;
		tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
	}
	const int tile_m = tile / tiles_n, tile_n = tile - tile_m * tiles_n;
	(void)tiles_m;
	const int m0 = tile_m * BM, n0 = tile_n * BN;
	la.p += (long)blockIdx.z * a_zoff;
	lb.p += (long)blockIdx.z * b_zoff;
	epi.c += (long)blockIdx.z * c_zoff;
	const int k_begin = blockIdx.y * k_per_split;
	const int k_end = (k_begin + k_per_split < K) ? k_begin + k_per_split : K;
	const int nk = (k_end - k_begin + GEMM_BK - 1) / GEMM_BK;

	TileFetch<LA, WM * 2> fa;
	TileFetch<LB, WN * 2> fb;
	fa.init(la, m0, t);
	fb.init(lb, n0, t);
	floatx16 acc[WM][WN];
#pragma unroll
	for (int i = 0; i < WM; i++)
#pragma unroll
		for (int j = 0; j < WN; j++)
#pragma unroll
			for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

	float4 ra[WM * 2], rb[WN * 2];
	if (nk > 0) {
		fa.fetch(la, k_begin, k_end, ra);
		fb.fetch(lb, k_begin, k_end, rb);
		fa.store(lds[0], ra, t);
		fb.store(lds[0] + A_FLOATS, rb, t);
	}
	__syncthreads();
	for (int kt = 0; kt < nk; kt++) {
		const int cur = kt & 1;
		if (kt + 1 < nk) {
			fa.fetch(la, k_begin + (kt + 1) * GEMM_BK, k_end, ra);
			fb.fetch(lb, k_begin + (kt + 1) * GEMM_BK, k_end, rb);
		}
		const float* sa = lds[cur];
		const float* sb = lds[cur] + A_FLOATS;
#pragma unroll
		for (int q = 0; q < 4; q++) {
			float fa_[WM][4], fb_[WN][4];
#pragma unroll
			for (int ti = 0; ti < WM; ti++) {
				const int row = wm * (32 * WM) + ti * 32 + li;
				if (LA::KCONTIG) {
					const float4 v = *(const float4*)(sa + row * GEMM_LDK + 8 * q + 4 * lh);
					fa_[ti][0] = v.x; fa_[ti][1] = v.y; fa_[ti][2] = v.z; fa_[ti][3] = v.w;
				} else {
#pragma unroll
					for (int e = 0; e < 4; e++) fa_[ti][e] = sa[(8 * q + 4 * lh + e) * BM + row];
				}
			}
#pragma unroll
			for (int tj = 0; tj < WN; tj++) {
				const int col = wn * (32 * WN) + tj * 32 + li;
				if (LB::KCONTIG) {
					const float4 v = *(const float4*)(sb + col * GEMM_LDK + 8 * q + 4 * lh);
					fb_[tj][0] = v.x; fb_[tj][1] = v.y; fb_[tj][2] = v.z; fb_[tj][3] = v.w;
				} else {
#pragma unroll
					for (int e = 0; e < 4; e++) fb_[tj][e] = sb[(8 * q + 4 * lh + e) * BN + col];
				}
			}
#pragma unroll
			for (int e = 0; e < 4; e++)
#pragma unroll
				for (int ti = 0; ti < WM; ti++)
#pragma unroll
					for (int tj = 0; tj < WN; tj++)
						acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_[ti][e], fb_[tj][e], acc[ti][tj], 0, 0, 0);
		}
		if (kt + 1 < nk) {
			fa.store(lds[cur ^ 1], ra, t);
			fb.store(lds[cur ^ 1] + A_FLOATS, rb, t);
		}
		__syncthreads();
	}
	// D layout of 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
	if (epi.bias) epi.bias += (long)blockIdx.z * bias_zoff;
#pragma unroll
	for (int ti = 0; ti < WM; ti++)
#pragma unroll
		for (int tj = 0; tj < WN; tj++) {
			const int n = n0 + wn * (32 * WN) + tj * 32 + li;
#pragma unroll
			for (int r = 0; r < 16; r++) {
				const int m = m0 + wm * (32 * WM) + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
				epi(m, n, acc[ti][tj][r]);
			}
		}
}

// Finish a split-K contraction: c = alpha * sum_s slab[s] (+ bias[n]) (+ old c). Fixed summation order => deterministic.
static __global__ void __launch_bounds__(256) splitk_reduce_kernel(const float* ws, const int splits, const long slab, float* c, const long ldm, const long ldn, const float* bias, const float alpha, const int accumulate, const int M, const int N)
{
	for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < slab; idx += (long)gridDim.x * blockDim.x) {
		const int m = (int)(idx / N), n = (int)(idx - (long)m * N);
		float v = 0.f;
		for (int s = 0; s < splits; s++) v += ws[(long)s * slab + idx];
		v *= alpha;
		if (bias) v += bias[n];
		const long o = (long)m * ldm + (long)n * ldn;
		if (accumulate) v += c[o];
		c[o] = v;
	}
}

} // namespace nnc
