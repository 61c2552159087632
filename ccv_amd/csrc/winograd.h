// Winograd F(4x4, 3x3) convolution for gfx950, fp32 -- the algorithm the reference's optimised CPU backend uses for
// 3x3 stride-1 convolutions (lib/nnc/cmd/convolution/cpu_opt/_ccv_nnc_conv_cpu_4x4_3x3_winograd.c:25-124 transforms the
// weights with G, :126- the tiles with B^T / A^T), re-designed for the MI355X as four stages over HBM-resident images:
//
//   U[z][K][C]  = G w G^T               weights, z = 6 * zy + zx in [0, 36)          wino_weight_kernel      (tiny)
//   V[z][T][C]  = B^T d B               one 6x6 input tile per 4x4 output tile t     wino_input_kernel       (HBM bound)
//   M[z][T][K]  = V[z] * U[z]^T         36 independent GEMMs, one launch (grid z)    mfma_gemm_f32_kernel    (MFMA bound)
//   b           = A^T M A + bias        4x4 outputs per tile, edge tiles clipped     wino_output_kernel      (HBM bound)
//
// 2.25x the activations travel through HBM twice (V, M) in exchange for 4x fewer MFMA FLOPs: pays once the direct
// contraction is long enough (C >= 256 on VGG-D).  The same four stages serve dgrad: the "input" is the
// output gradient, the weights are read with flipped taps and swapped roles (U[z][C][K]), the padding is 2 - p.
// Transform matrices: Lavin & Gray, "Fast Algorithms for Convolutional Neural Networks" (interpolation points 0, +-1, +-2).
#pragma once
#include "mfma_gemm.h"

namespace nnc {

// y = B^T x (6 -> 6), y = G x (3 -> 6), y = A^T x (6 -> 4), element type T (float or float4-like with + - *)
template <class T>
__device__ __forceinline__ void wino_bt(const T (&x)[6], T (&y)[6])
{
	const T a = x[4] - 4.f * x[2], b = x[3] - 4.f * x[1];
	const T c = x[4] - x[2], d = 2.f * (x[3] - x[1]);
	y[0] = 4.f * x[0] - 5.f * x[2] + x[4];
	y[1] = a + b;
	y[2] = a - b;
	y[3] = c + d;
	y[4] = c - d;
	y[5] = 4.f * x[1] - 5.f * x[3] + x[5];
}
template <class T>
__device__ __forceinline__ void wino_g(const T (&x)[3], T (&y)[6])
{
	const T s = x[0] + x[2];
	const T e = x[0] * (1.f / 24.f) + x[2] * (1.f / 6.f), o = x[1] * (1.f / 12.f);
	y[0] = x[0] * 0.25f;
	y[1] = (s + x[1]) * (-1.f / 6.f);
	y[2] = (s - x[1]) * (-1.f / 6.f);
	y[3] = e + o;
	y[4] = e - o;
	y[5] = x[2];
}
template <class T>
__device__ __forceinline__ void wino_at(const T (&x)[6], T (&y)[4])
{
	const T s1 = x[1] + x[2], d1 = x[1] - x[2], s2 = x[3] + x[4], d2 = x[3] - x[4];
	y[0] = x[0] + s1 + s2;
	y[1] = d1 + 2.f * d2;
	y[2] = s1 + 4.f * s2;
	y[3] = d1 + 8.f * d2 + x[5];
}

// F(3x3, 4x4) (filter gradient: 3x3 outputs from a 6x6 input tile and a 4x4 output-gradient tile; same points, same B^T):
// y = G' x (4 -> 6) on the output-gradient tile, y = A'^T x (6 -> 3) on the products.
template <class T>
__device__ __forceinline__ void wino_g4(const T (&x)[4], T (&y)[6])
{
	const T e = x[0] + x[2], o = x[1] + x[3];
	const T e2 = x[0] * (1.f / 24.f) + x[2] * (1.f / 6.f), o2 = x[1] * (1.f / 12.f) + x[3] * (1.f / 3.f);
	y[0] = x[0] * 0.25f;
	y[1] = (e + o) * (-1.f / 6.f);
	y[2] = (e - o) * (-1.f / 6.f);
	y[3] = e2 + o2;
	y[4] = e2 - o2;
	y[5] = x[3];
}
template <class T>
__device__ __forceinline__ void wino_at3(const T (&x)[6], T (&y)[3])
{
	const T s1 = x[1] + x[2], d1 = x[1] - x[2], s2 = x[3] + x[4], d2 = x[3] - x[4];
	y[0] = x[0] + s1 + s2;
	y[1] = d1 + 2.f * d2;
	y[2] = s1 + 4.f * s2 + x[5];
}

struct f4 {
	float x, y, z, w;
	__device__ __forceinline__ f4() {}
	__device__ __forceinline__ f4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
	__device__ __forceinline__ f4(const float4 v) : x(v.x), y(v.y), z(v.z), w(v.w) {}
	__device__ __forceinline__ operator float4() const { return make_float4(x, y, z, w); }
};
__device__ __forceinline__ f4 operator+(const f4 a, const f4 b) { return f4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ f4 operator-(const f4 a, const f4 b) { return f4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ f4 operator*(const float s, const f4 a) { return f4(s * a.x, s * a.y, s * a.z, s * a.w); }
__device__ __forceinline__ f4 operator*(const f4 a, const float s) { return f4(s * a.x, s * a.y, s * a.z, s * a.w); }

// U[z][r][q] = (G w G^T)[zy][zx] with, for forward, r = output channel k, q = input channel c  (w[k][i][j][c]);
// for dgrad (FLIP), r = input channel c, q = output channel k and the taps mirrored: w'[c][i][j][k] = w[k][2-i][2-j][c].
// One thread per (r, q); q fastest so the 36 stores of a wave are contiguous runs.
template <bool FLIP>
static __global__ void __launch_bounds__(256) wino_weight_kernel(const float* __restrict__ w, float* __restrict__ u, const int K, const int C)
{
	const int R = FLIP ? C : K, Q = FLIP ? K : C;
	const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (idx >= (long)R * Q) return;
	const int r = (int)(idx / Q), q = (int)(idx - (long)r * Q);
	const int k = FLIP ? q : r, c = FLIP ? r : q;
	float g[3][3];
#pragma unroll
	for (int i = 0; i < 3; i++)
#pragma unroll
		for (int j = 0; j < 3; j++)
			g[i][j] = w[((long)k * 9 + (FLIP ? (2 - i) * 3 + (2 - j) : i * 3 + j)) * C + c];
	float t[6][3]; // G g
#pragma unroll
	for (int j = 0; j < 3; j++) {
		const float col[3] = { g[0][j], g[1][j], g[2][j] };
		float y[6];
		wino_g(col, y);
#pragma unroll
		for (int i = 0; i < 6; i++) t[i][j] = y[i];
	}
	const long plane = (long)R * Q;
#pragma unroll
	for (int i = 0; i < 6; i++) {
		float y[6];
		wino_g(t[i], y);
#pragma unroll
		for (int j = 0; j < 6; j++) u[(long)(i * 6 + j) * plane + idx] = y[j];
	}
}

// Geometry shared by the tile kernels: source image (n, h, w, c) strides in floats, channels dense.
struct WinoTiles {
	int TH, TW, T;   // tiles per image column / row, total
	int H, W;        // source extent (input transform) or destination extent (output transform)
	long sn, sh, sw; // its strides
	int oy, ox;      // input transform: source row / column of tile (0, 0)'s first element = -padding
	int C4;          // channels / 4
	int relu;        // output transform: write max(0, .) (NNC_MI355X_CONV_ALGO_FUSE_RELU)
	const float* mask; long m_sn, m_sh, m_sw; // output transform of a data gradient: zero where mask <= 0 (the ReLU backward of the map the gradient belongs to), or null
	FastDiv d_c4, d_tw, d_th;
};

// V[z][t][c] = (B^T d B)[zy][zx], d = the 6x6 source patch of tile t (zero outside the image).  One thread per (t, c4):
// 36 16-byte loads, 12 six-point transforms on float4, 36 16-byte stores (the c4 threads of a tile write contiguous runs).
static __global__ void __launch_bounds__(256) wino_input_kernel(const float* __restrict__ a, float* __restrict__ v, const WinoTiles g)
{
	const long idx = (long)nnc_xcd_block(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x; // neighbouring tiles share 2 of their 6 rows / columns: same XCD
	if (idx >= (long)g.T * g.C4) return;
	const int t = g.d_c4.div((int)idx), c4 = (int)(idx - (long)t * g.C4);
	const int tn = g.d_tw.div(t), tx = t - tn * g.TW;
	const int n = g.d_th.div(tn), ty = tn - n * g.TH;
	const int y0 = ty * 4 + g.oy, x0 = tx * 4 + g.ox;
	const float* const src = a + (long)n * g.sn + (long)c4 * 4;
	f4 s[6][6]; // rows transformed horizontally: s[r] = d[r] B
#pragma unroll
	for (int r = 0; r < 6; r++) {
		const int y = y0 + r;
		const bool yok = (y >= 0) & (y < g.H);
		f4 d[6];
#pragma unroll
		for (int q = 0; q < 6; q++) {
			const int x = x0 + q;
			const bool ok = yok & (x >= 0) & (x < g.W);
			d[q] = ok ? f4(*(const float4*)(src + (long)y * g.sh + (long)x * g.sw)) : f4(0.f, 0.f, 0.f, 0.f);
		}
		wino_bt(d, s[r]);
	}
	const long plane = (long)g.T * g.C4 * 4;
	float* const dst = v + idx * 4;
#pragma unroll
	for (int q = 0; q < 6; q++) {
		const f4 col[6] = { s[0][q], s[1][q], s[2][q], s[3][q], s[4][q], s[5][q] };
		f4 y[6];
		wino_bt(col, y);
#pragma unroll
		for (int r = 0; r < 6; r++) *(float4*)(dst + (long)(r * 6 + q) * plane) = y[r];
	}
}

// b[n, 4 ty + i, 4 tx + j, k] = (A^T m A)[i][j] (+ bias[k]), m = M[.][t][k]; rows / columns past the image are dropped.
static __global__ void __launch_bounds__(256) wino_output_kernel(const float* __restrict__ m, const float* __restrict__ bias, float* __restrict__ b, const WinoTiles g)
{
	const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
	if (idx >= (long)g.T * g.C4) return;
	const int t = g.d_c4.div((int)idx), k4 = (int)(idx - (long)t * g.C4);
	const int tn = g.d_tw.div(t), tx = t - tn * g.TW;
	const int n = g.d_th.div(tn), ty = tn - n * g.TH;
	const long plane = (long)g.T * g.C4 * 4;
	const float* const src = m + idx * 4;
	f4 s[4][6]; // columns transformed vertically: s[.][q] = A^T m[.][q]
#pragma unroll
	for (int q = 0; q < 6; q++) {
		f4 col[6];
#pragma unroll
		for (int r = 0; r < 6; r++) col[r] = f4(*(const float4*)(src + (long)(r * 6 + q) * plane));
		f4 y[4];
		wino_at(col, y);
#pragma unroll
		for (int i = 0; i < 4; i++) s[i][q] = y[i];
	}
	const f4 bv = bias ? f4(*(const float4*)(bias + k4 * 4)) : f4(0.f, 0.f, 0.f, 0.f);
	float* const dst = b + (long)n * g.sn + (long)k4 * 4;
	const float* const msk = g.mask ? g.mask + (long)n * g.m_sn + (long)k4 * 4 : 0;
#pragma unroll
	for (int i = 0; i < 4; i++) {
		f4 y[4];
		wino_at(s[i], y);
		const int oy = ty * 4 + i;
#pragma unroll
		for (int j = 0; j < 4; j++) {
			const int ox = tx * 4 + j;
			if ((oy < g.H) & (ox < g.W)) {
				f4 o = y[j] + bv;
				if (g.relu) o = f4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
				if (msk) {
					const float4 mv = *(const float4*)(msk + (long)oy * g.m_sh + (long)ox * g.m_sw);
					o = f4(mv.x > 0.f ? o.x : 0.f, mv.y > 0.f ? o.y : 0.f, mv.z > 0.f ? o.z : 0.f, mv.w > 0.f ? o.w : 0.f);
				}
				*(float4*)(dst + (long)oy * g.sh + (long)ox * g.sw) = (float4)o;
			}
		}
	}
}

// W[z][t][k] = (G' dy G'^T)[zy][zx], dy = the 4x4 output-gradient tile t (zero past the image edge).  One thread per (t, k4).
// COLSUM: the kernel reads every element of the output gradient exactly once, so the bias gradient's column sums ride
// along: each thread adds up its 16 pixels, the block folds the tiles it covers through LDS (fixed order) and writes one
// row of partial sums per block (blockpart[block][K]); colsum_f32 over those rows finishes -- the separate pass over g
// (3.4 ms of the VGG-D step) is gone.  Needs a whole number of tiles per block: (K / 4) divides 256.
template <bool COLSUM>
static __global__ void __launch_bounds__(256) wino_outgrad_kernel(const float* __restrict__ gr, float* __restrict__ wt, const WinoTiles g, float* __restrict__ blockpart)
{
	__shared__ float4 red[COLSUM ? 256 : 1];
	const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
	const bool live = idx < (long)g.T * g.C4;
	f4 sum(0.f, 0.f, 0.f, 0.f);
	if (live) {
		const int t = g.d_c4.div((int)idx), k4 = (int)(idx - (long)t * g.C4);
		const int tn = g.d_tw.div(t), tx = t - tn * g.TW;
		const int n = g.d_th.div(tn), ty = tn - n * g.TH;
		const float* const src = gr + (long)n * g.sn + (long)k4 * 4;
		f4 s[4][6]; // rows transformed horizontally: s[r] = dy[r] G'^T
#pragma unroll
		for (int r = 0; r < 4; r++) {
			const int y = ty * 4 + r;
			f4 d[4];
#pragma unroll
			for (int q = 0; q < 4; q++) {
				const int x = tx * 4 + q;
				const bool ok = (y < g.H) & (x < g.W);
				d[q] = ok ? f4(*(const float4*)(src + (long)y * g.sh + (long)x * g.sw)) : f4(0.f, 0.f, 0.f, 0.f);
				if (COLSUM) sum = sum + d[q];
			}
			wino_g4(d, s[r]);
		}
		const long plane = (long)g.T * g.C4 * 4;
		float* const dst = wt + idx * 4;
#pragma unroll
		for (int q = 0; q < 6; q++) {
			const f4 col[4] = { s[0][q], s[1][q], s[2][q], s[3][q] };
			f4 y[6];
			wino_g4(col, y);
#pragma unroll
			for (int r = 0; r < 6; r++) *(float4*)(dst + (long)(r * 6 + q) * plane) = y[r];
		}
	}
	if (COLSUM) {
		red[threadIdx.x] = (float4)sum;
		__syncthreads();
		if ((int)threadIdx.x < g.C4) { // 256 % C4 == 0 and block bases are multiples of 256: thread k4 owns channel group k4
			f4 acc(0.f, 0.f, 0.f, 0.f);
			for (int j = threadIdx.x; j < 256; j += g.C4) acc = acc + f4(red[j]);
			*(float4*)(blockpart + ((long)blockIdx.x * g.C4 + threadIdx.x) * 4) = (float4)acc;
		}
	}
}

// dw[k][i][j][c] (+)= (A'^T du A')[i][j], du = dU[.][k][c].  One thread per (k, c), c fastest.
static __global__ void __launch_bounds__(256) wino_wgrad_final_kernel(const float* __restrict__ du, float* __restrict__ dw, const int K, const int C, const int accumulate)
{
	const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
	const long plane = (long)K * C;
	if (idx >= plane) return;
	const int k = (int)(idx / C), c = (int)(idx - (long)k * C);
	float s[3][6];
#pragma unroll
	for (int q = 0; q < 6; q++) {
		float col[6];
#pragma unroll
		for (int r = 0; r < 6; r++) col[r] = du[(long)(r * 6 + q) * plane + idx];
		float y[3];
		wino_at3(col, y);
#pragma unroll
		for (int i = 0; i < 3; i++) s[i][q] = y[i];
	}
#pragma unroll
	for (int i = 0; i < 3; i++) {
		float y[3];
		wino_at3(s[i], y);
#pragma unroll
		for (int j = 0; j < 3; j++) {
			float* const o = dw + ((long)k * 9 + i * 3 + j) * C + c;
			*o = accumulate ? *o + y[j] : y[j];
		}
	}
}

} // namespace nnc
