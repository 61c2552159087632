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
// Geometry: 128x128 block tile, BK = 32, 256 threads = 4 waves (2x2), each wave owns 64x64 = 2x2 MFMA tiles
// (64 accumulator registers).  Operand tiles are staged HBM -> registers -> LDS (double buffered, one barrier per
// K-step; the global loads of tile t+1 are issued before the MFMAs of tile t and written to LDS after them).
// LDS images:
//   k-contiguous operand:   [128 rows][36]  (row stride 36 floats: ds_read_b128 by 16-lane groups is conflict free,
//                                            36*r mod 64 hits 16 distinct 4-bank slots for 16 distinct rows)
//   row-contiguous operand: [32 k][128]     (ds_read_b32, lanes 0-31 read 32 consecutive banks, the other half-wave
//                                            is a different k row: no conflicts)
// Both images feed the same k permutation: MFMA number (q,e) of a K-step consumes k = 8q+e (lanes 0-31) and
// k = 8q+4+e (lanes 32-63), so a k-contiguous operand needs ONE ds_read_b128 per four MFMAs.
// Workgroup -> tile mapping is XCD aware: consecutive tile ids (which share im2col halo rows / weight panels)
// are dispatched to the same XCD so its private 4 MiB L2 serves the re-reads.
#pragma once
#include <hip/hip_runtime.h>

namespace nnc {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int GEMM_BM = 128, GEMM_BN = 128, GEMM_BK = 32, GEMM_THREADS = 256;
constexpr int GEMM_LDK = 36;  // row stride of a k-contiguous LDS image
constexpr int GEMM_LDR = 128; // k-row stride of a row-contiguous LDS image
constexpr int GEMM_TILE_FLOATS = GEMM_BM * GEMM_LDK; // 4608 >= 32 * 128

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// ---------------------------------------------------------------------------------------------- loaders
// Concept:  static constexpr bool KCONTIG;  const float* p;  Ctx make(int r) const;  float4 load(const Ctx&, int k) const;
//   KCONTIG:  load() returns elements (r, k..k+3)          (k is a multiple of 4)
//   !KCONTIG: load() returns elements (r..r+3, k)          (r is a multiple of 4)
// Out-of-range rows / k and im2col padding read as zero.  VEC = one 16-byte global load per chunk (needs the
// alignment / divisibility the host checks before picking it); !VEC = four guarded scalar loads.

// Plain matrix: element(r, k) = p[r * ldr + k * ldk].  KC => ldk == 1, otherwise ldr == 1.
template <bool KC, bool VEC>
struct MatLoader {
	static constexpr bool KCONTIG = KC;
	const float* p;
	long ldr, ldk;
	int R, K;
	struct Ctx { const float* base; int r; };
	__device__ __forceinline__ Ctx make(int r) const
	{
		Ctx c;
		c.r = r;
		c.base = p + (KC ? (long)r * ldr : (long)r);
		return c;
	}
	__device__ __forceinline__ float4 load(const Ctx& c, int k) const
	{
		if (KC) {
			if (c.r >= R || k >= K) return f4zero();
			if (VEC) return *(const float4*)(c.base + k);
			float4 v = f4zero();
			v.x = c.base[k];
			if (k + 1 < K) v.y = c.base[k + 1];
			if (k + 2 < K) v.z = c.base[k + 2];
			if (k + 3 < K) v.w = c.base[k + 3];
			return v;
		} else {
			if (c.r >= R || k >= K) return f4zero();
			const float* q = c.base + (long)k * ldk;
			if (VEC) return *(const float4*)q;
			float4 v = f4zero();
			v.x = q[0];
			if (c.r + 1 < R) v.y = q[1];
			if (c.r + 2 < R) v.z = q[2];
			if (c.r + 3 < R) v.w = q[3];
			return v;
		}
	}
};

// im2col gather with the reduction index running (tap_y, tap_x, channel), channel fastest: rows are output
// pixels m = (n, oy, ox).  Serves conv forward (source = a) and conv dgrad (source = g, taps walked backwards).
//   t_y = oy * my + oy_off + i * ty ;  source y = t_y / dv_y, valid iff t_y >= 0, t_y % dv_y == 0, y < H   (same for x)
//   forward: my = stride, oy_off = -border, ty = +dilation, dv = 1
//   dgrad:   my = 1, oy_off = +border, ty = -dilation, dv = stride
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
	struct Ctx { const float* base; int iy0, ix0; };
	__device__ __forceinline__ Ctx make(int m) const
	{
		Ctx c;
		if (m >= M) { c.base = 0; c.iy0 = 0; c.ix0 = 0; return c; }
		const int n = m / OHW;
		const int rem = m - n * OHW;
		const int oy = rem / OW;
		const int ox = rem - oy * OW;
		c.base = p + (long)n * s_n;
		c.iy0 = oy * my + oy_off;
		c.ix0 = ox * mx + ox_off;
		return c;
	}
	__device__ __forceinline__ float elem(const Ctx& c, int k) const
	{
		if (k >= K) return 0.f;
		const int i = k / KWC;
		const int r = k - i * KWC;
		const int j = r / C;
		const int ch = r - j * C;
		int y = c.iy0 + i * ty, x = c.ix0 + j * tx;
		if (y < 0 || x < 0) return 0.f;
		if (dv_y != 1) { if (y % dv_y) return 0.f; y /= dv_y; }
		if (dv_x != 1) { if (x % dv_x) return 0.f; x /= dv_x; }
		if (y >= H || x >= W) return 0.f;
		return c.base[(long)y * s_h + (long)x * s_w + ch];
	}
	__device__ __forceinline__ float4 load(const Ctx& c, int k) const
	{
		if (!c.base) return f4zero();
		if (VEC) {
			if (k >= K) return f4zero();
			const int i = k / KWC;
			const int r = k - i * KWC;
			const int j = r / C;
			const int ch = r - j * C;
			int y = c.iy0 + i * ty, x = c.ix0 + j * tx;
			if (y < 0 || x < 0) return f4zero();
			if (dv_y != 1) { if (y % dv_y) return f4zero(); y /= dv_y; }
			if (dv_x != 1) { if (x % dv_x) return f4zero(); x /= dv_x; }
			if (y >= H || x >= W) return f4zero();
			return *(const float4*)(c.base + (long)y * s_h + (long)x * s_w + ch);
		}
		return make_float4(elem(c, k), elem(c, k + 1), elem(c, k + 2), elem(c, k + 3));
	}
};

// conv dgrad weights: B(k = (tap, ko), n = c) = w[ko][tap][c]  (n contiguous).
template <bool VEC>
struct WgtDgradNC {
	static constexpr bool KCONTIG = false;
	const float* p;
	long ko_stride; // kh*kw*C
	int C, Ko, K;   // K = kh*kw*Ko
	struct Ctx { int c; };
	__device__ __forceinline__ Ctx make(int c) const { Ctx x; x.c = c; return x; }
	__device__ __forceinline__ float4 load(const Ctx& c, int k) const
	{
		if (k >= K || c.c >= C) return f4zero();
		const int tap = k / Ko;
		const int ko = k - tap * Ko;
		const float* q = p + (long)ko * ko_stride + (long)tap * C + c.c;
		if (VEC) return *(const float4*)q;
		float4 v = f4zero();
		v.x = q[0];
		if (c.c + 1 < C) v.y = q[1];
		if (c.c + 2 < C) v.z = q[2];
		if (c.c + 3 < C) v.w = q[3];
		return v;
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
	struct Ctx { int nn; int off_y[4], off_x[4], ch[4]; }; // per column (i*dy - py, j*dx - px, c)
	__device__ __forceinline__ Ctx make(int nn) const
	{
		Ctx c;
		c.nn = nn;
#pragma unroll
		for (int e = 0; e < 4; e++) {
			const int q = nn + e;
			const int i = q / KWC;
			const int r = q - i * KWC;
			const int j = r / C;
			c.off_y[e] = i * dy - py;
			c.off_x[e] = j * dx - px;
			c.ch[e] = r - j * C;
			if (VEC) break; // VEC: the four columns share a tap and are channel-consecutive
		}
		return c;
	}
	__device__ __forceinline__ float4 load(const Ctx& c, int k) const
	{
		if (k >= K || c.nn >= NN) return f4zero();
		const int n = k / OHW;
		const int rem = k - n * OHW;
		const int oy = rem / OW;
		const int ox = rem - oy * OW;
		const float* base = p + (long)n * s_n;
		const int by = oy * sy, bx = ox * sx;
		if (VEC) {
			const int y = by + c.off_y[0], x = bx + c.off_x[0];
			if (y < 0 || x < 0 || y >= H || x >= W) return f4zero();
			return *(const float4*)(base + (long)y * s_h + (long)x * s_w + c.ch[0]);
		}
		float v[4];
#pragma unroll
		for (int e = 0; e < 4; e++) {
			const int y = by + c.off_y[e], x = bx + c.off_x[e];
			v[e] = (c.nn + e < NN && y >= 0 && x >= 0 && y < H && x < W) ? base[(long)y * s_h + (long)x * s_w + c.ch[e]] : 0.f;
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
template <class L>
__device__ __forceinline__ void gemm_stage_store(float* lds, const float4 (&r)[4], int t)
{
#pragma unroll
	for (int jj = 0; jj < 4; jj++) {
		const int id = t + GEMM_THREADS * jj;
		if (L::KCONTIG) {
			const int row = id >> 3, kc = (id & 7) << 2;
			*(float4*)(lds + row * GEMM_LDK + kc) = r[jj];
		} else {
			const int k = id >> 5, rc = (id & 31) << 2;
			*(float4*)(lds + k * GEMM_LDR + rc) = r[jj];
		}
	}
}

// grid: x = tiles (XCD-swizzled), y = split-K slices, z = batch / conv group.
template <class LA, class LB, class EPI>
__global__ void __launch_bounds__(GEMM_THREADS) mfma_gemm_f32_kernel(LA la, LB lb, EPI epi, const int tiles_m, const int tiles_n, const int K, const int k_per_split, const long a_zoff, const long b_zoff, const long c_zoff, const long bias_zoff)
{
	__shared__ __attribute__((aligned(16))) float lds[2][2][GEMM_TILE_FLOATS]; // [buffer][A|B]
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
	const int m0 = tile_m * GEMM_BM, n0 = tile_n * GEMM_BN;
	la.p += (long)blockIdx.z * a_zoff;
	lb.p += (long)blockIdx.z * b_zoff;
	epi.c += (long)blockIdx.z * c_zoff;
	const int k_begin = blockIdx.y * k_per_split;
	const int k_end = (k_begin + k_per_split < K) ? k_begin + k_per_split : K;
	const int nk = (k_end - k_begin + GEMM_BK - 1) / GEMM_BK;

	// Per-thread gather contexts. KCONTIG: 4 rows (id>>3), one k chunk (id&7). !KCONTIG: one row chunk (id&31), 4 k's (id>>5).
	constexpr int NCA = LA::KCONTIG ? 4 : 1, NCB = LB::KCONTIG ? 4 : 1;
	typename LA::Ctx ca[NCA];
	typename LB::Ctx cb[NCB];
	int ka[4], kb[4];
#pragma unroll
	for (int jj = 0; jj < 4; jj++) {
		const int id = t + GEMM_THREADS * jj;
		if (LA::KCONTIG) { ca[jj % NCA] = la.make(m0 + (id >> 3)); ka[jj] = (id & 7) << 2; }
		else { if (jj == 0) ca[0] = la.make(m0 + ((id & 31) << 2)); ka[jj] = id >> 5; }
		if (LB::KCONTIG) { cb[jj % NCB] = lb.make(n0 + (id >> 3)); kb[jj] = (id & 7) << 2; }
		else { if (jj == 0) cb[0] = lb.make(n0 + ((id & 31) << 2)); kb[jj] = id >> 5; }
	}
	floatx16 acc[2][2];
#pragma unroll
	for (int i = 0; i < 2; i++)
#pragma unroll
		for (int j = 0; j < 2; j++)
#pragma unroll
			for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

	float4 ra[4], rb[4];
	// k beyond this split's range must read as zero: loaders only know the global K, so clamp here.
#define NNC_GEMM_FETCH(kt) \
	do { \
		const int kbase = k_begin + (kt) * GEMM_BK; \
		_Pragma("unroll") for (int jj = 0; jj < 4; jj++) { \
			const int kk_a = kbase + ka[jj]; \
			const int kk_b = kbase + kb[jj]; \
			ra[jj] = (kk_a < k_end) ? la.load(ca[jj % NCA], kk_a) : f4zero(); \
			rb[jj] = (kk_b < k_end) ? lb.load(cb[jj % NCB], kk_b) : f4zero(); \
		} \
	} while (0)

	if (nk > 0) {
		NNC_GEMM_FETCH(0);
		gemm_stage_store<LA>(lds[0][0], ra, t);
		gemm_stage_store<LB>(lds[0][1], rb, t);
	}
	__syncthreads();
	for (int kt = 0; kt < nk; kt++) {
		const int cur = kt & 1;
		if (kt + 1 < nk) NNC_GEMM_FETCH(kt + 1);
		const float* sa = lds[cur][0];
		const float* sb = lds[cur][1];
#pragma unroll
		for (int q = 0; q < 4; q++) {
			float fa[2][4], fb[2][4];
#pragma unroll
			for (int ti = 0; ti < 2; ti++) {
				const int row = wm * 64 + ti * 32 + li;
				if (LA::KCONTIG) {
					const float4 v = *(const float4*)(sa + row * GEMM_LDK + 8 * q + 4 * lh);
					fa[ti][0] = v.x; fa[ti][1] = v.y; fa[ti][2] = v.z; fa[ti][3] = v.w;
				} else {
#pragma unroll
					for (int e = 0; e < 4; e++) fa[ti][e] = sa[(8 * q + 4 * lh + e) * GEMM_LDR + row];
				}
				const int col = wn * 64 + ti * 32 + li;
				if (LB::KCONTIG) {
					const float4 v = *(const float4*)(sb + col * GEMM_LDK + 8 * q + 4 * lh);
					fb[ti][0] = v.x; fb[ti][1] = v.y; fb[ti][2] = v.z; fb[ti][3] = v.w;
				} else {
#pragma unroll
					for (int e = 0; e < 4; e++) fb[ti][e] = sb[(8 * q + 4 * lh + e) * GEMM_LDR + col];
				}
			}
#pragma unroll
			for (int e = 0; e < 4; e++)
#pragma unroll
				for (int ti = 0; ti < 2; ti++)
#pragma unroll
					for (int tj = 0; tj < 2; tj++)
						acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ti][e], fb[tj][e], acc[ti][tj], 0, 0, 0);
		}
		if (kt + 1 < nk) {
			gemm_stage_store<LA>(lds[cur ^ 1][0], ra, t);
			gemm_stage_store<LB>(lds[cur ^ 1][1], rb, t);
		}
		__syncthreads();
	}
#undef NNC_GEMM_FETCH
	// D layout of 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
	if (epi.bias) epi.bias += (long)blockIdx.z * bias_zoff;
#pragma unroll
	for (int ti = 0; ti < 2; ti++)
#pragma unroll
		for (int tj = 0; tj < 2; tj++) {
			const int n = n0 + wn * 64 + tj * 32 + li;
#pragma unroll
			for (int r = 0; r < 16; r++) {
				const int m = m0 + wm * 64 + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
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
