// CCV_NNC_MAX_POOL / AVERAGE_POOL FORWARD + BACKWARD on gfx950.  HBM-bound (algorithmic bytes |a| + |b| forward,
// |g| + |a| + |b| + |h| max-pool backward, |g| + |h| avg-pool backward); one lane per element with the innermost
// (memory-contiguous) index on consecutive lanes, so every wave touches whole 256-byte rows.
// The results are BIT-EXACT with the reference CPU backend because each lane replays the oracle's float operation
// order for its element:
//   max fwd   lib/nnc/cmd/pool/ccv_nnc_max_pool_cpu_ref.c:13-61    running max over the border-clipped window, raster order
//   max bwd   lib/nnc/cmd/pool/ccv_nnc_max_pool_cpu_ref.c:63-141   h[p] += g[o] for EVERY p in the window with a[p] == b[o]
//             (the oracle scatters over outputs in raster order; we gather per input position over the same outputs in
//              the same raster order, starting from 0 -- the identical addition sequence)
//   avg fwd   lib/nnc/cmd/pool/ccv_nnc_avg_pool_cpu_ref.c:13-60    sum / (clipped window element count)
//   avg bwd   lib/nnc/cmd/pool/ccv_nnc_avg_pool_cpu_ref.c:62-109   h[p] += g[o] / count(o)
// (The CPU oracle only walks image 0 of a batch; on device every image is processed the same way.)
// Replaces cudnnPoolingForward/Backward of lib/nnc/cmd/pool/gpu/ccv_nnc_{max,avg}_pool_gpu_cudnn.cu.
#include "common.h"
#include "mfma_gemm.h" // FastDiv

using namespace nnc;

namespace {

typedef _Float16 half_t;
struct pool_geom_t {
	int N, H, W, C, OH, OW;
	int kh, kw, sy, sx, pby, pbx;
	long a_sn, a_sh, a_sw, a_sc; // input-shaped tensors (a, h)
	long b_sn, b_sh, b_sw, b_sc; // output-shaped tensors (b, g)
	FastDiv d_c, d_w, d_h, d_ow, d_oh; // index decomposition without hardware division
	FastDiv d_c4, d_sy, d_sx;          // float4 kernels: C / 4 channel groups; strides for the window bounds
	int relu_mask;                     // max backward: a is a ReLU's output, h is masked by a > 0 as well (NNC_MI355X_POOL_ALGO_FUSE_RELU_BACKWARD)
};

// idx -> (n, y, x, c) with the memory-contiguous index fastest; idx < 2^31 (the host splits larger tensors per image).
template <bool NHWC>
__device__ __forceinline__ void unflatten(int idx, const FastDiv& d1, const FastDiv& d2, const FastDiv& dc, int& n, int& y, int& x, int& c)
{
	if (NHWC) {
		int q = dc.div(idx); c = idx - q * dc.d; idx = q;
		q = d2.div(idx); x = idx - q * d2.d; idx = q;
		q = d1.div(idx); y = idx - q * d1.d; n = q;
	} else {
		int q = d2.div(idx); x = idx - q * d2.d; idx = q;
		q = d1.div(idx); y = idx - q * d1.d; idx = q;
		q = dc.div(idx); c = idx - q * dc.d; n = q;
	}
}

// T = float, or _Float16 for the half-precision trainers' tensors (values compared / summed as fp32: exact for the maximum)
template <bool NHWC, bool IS_MAX, class T>
__global__ void __launch_bounds__(256) pool_forw_kernel(const pool_geom_t g, const T* a, T* b, const size_t total)
{
	for (size_t idx64 = (size_t)nnc_xcd_block(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x; idx64 < total; idx64 += (size_t)gridDim.x * blockDim.x) {
		int n, oy, ox, c;
		unflatten<NHWC>((int)idx64, g.d_oh, g.d_ow, g.d_c, n, oy, ox, c);
		int y0 = oy * g.sy - g.pby, x0 = ox * g.sx - g.pbx;
		int y1 = y0 + g.kh, x1 = x0 + g.kw;
		if (y0 < 0) y0 = 0;
		if (x0 < 0) x0 = 0;
		if (y1 > g.H) y1 = g.H;
		if (x1 > g.W) x1 = g.W;
		const T* ap = a + n * g.a_sn + c * g.a_sc;
		float v;
		if (IS_MAX) {
			v = (float)ap[y0 * g.a_sh + x0 * g.a_sw];
			for (int y = y0; y < y1; y++)
				for (int x = x0; x < x1; x++) {
					const float u = (float)ap[y * g.a_sh + x * g.a_sw];
					if (u > v) v = u;
				}
		} else {
			v = 0.f;
			for (int y = y0; y < y1; y++)
				for (int x = x0; x < x1; x++) v += (float)ap[y * g.a_sh + x * g.a_sw];
			v = v / (float)((y1 - y0) * (x1 - x0));
		}
		b[n * g.b_sn + oy * g.b_sh + ox * g.b_sw + c * g.b_sc] = (T)v;
	}
}

// ceil(a / b) for b > 0 and any sign of a
__device__ __forceinline__ int ceil_div(int a, int b) { return a >= 0 ? (a + b - 1) / b : -((-a) / b); }
__device__ __forceinline__ int floor_div(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

template <bool NHWC, bool IS_MAX, class T>
__global__ void __launch_bounds__(256) pool_back_kernel(const pool_geom_t g, const T* gr, const T* a, const T* b, T* h, const size_t total)
{
	for (size_t idx64 = (size_t)nnc_xcd_block(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x; idx64 < total; idx64 += (size_t)gridDim.x * blockDim.x) {
		int n, y, x, c;
		unflatten<NHWC>((int)idx64, g.d_h, g.d_w, g.d_c, n, y, x, c);
		// windows that contain (y, x): ceil((t - k + 1) / s) .. floor(t / s) with t = y + pb >= 0; the ceiling as a floor of a
		// non-negative dividend (+ k * s, - k afterwards), both by multiply-shift (FastDiv) instead of hardware division
		const int ty = y + g.pby, tx = x + g.pbx;
		int oy0 = g.d_sy.div(ty - g.kh + g.kh * g.sy + g.sy) - g.kh, oy1 = g.d_sy.div(ty);
		int ox0 = g.d_sx.div(tx - g.kw + g.kw * g.sx + g.sx) - g.kw, ox1 = g.d_sx.div(tx);
		if (oy0 < 0) oy0 = 0;
		if (ox0 < 0) ox0 = 0;
		if (oy1 > g.OH - 1) oy1 = g.OH - 1;
		if (ox1 > g.OW - 1) ox1 = g.OW - 1;
		const long ob = n * g.b_sn + c * g.b_sc;
		float acc = 0.f;
		float av = 0.f;
		if (IS_MAX) av = (float)a[n * g.a_sn + y * g.a_sh + x * g.a_sw + c * g.a_sc];
		for (int oy = oy0; oy <= oy1; oy++)
			for (int ox = ox0; ox <= ox1; ox++) {
				const long o = ob + oy * g.b_sh + ox * g.b_sw;
				if (IS_MAX) {
					if (av == (float)b[o]) acc += (float)gr[o];
				} else {
					int wy0 = oy * g.sy - g.pby, wx0 = ox * g.sx - g.pbx;
					int wy1 = wy0 + g.kh, wx1 = wx0 + g.kw;
					if (wy0 < 0) wy0 = 0;
					if (wx0 < 0) wx0 = 0;
					if (wy1 > g.H) wy1 = g.H;
					if (wx1 > g.W) wx1 = g.W;
					acc += (float)gr[o] / (float)((wy1 - wy0) * (wx1 - wx0));
				}
			}
		if (IS_MAX && g.relu_mask && !(av > 0.f)) acc = 0.f;
		h[n * g.a_sn + y * g.a_sh + x * g.a_sw + c * g.a_sc] = (T)acc;
	}
}

// ---- NHWC, C % 4 == 0, dense channels: one lane per FOUR channels (16-byte loads / stores, a quarter of the index
// arithmetic per byte), window bounds by multiply-shift instead of hardware division.  Same per-element operation order
// as the scalar kernels above, so the result is bit-identical.
__device__ __forceinline__ float4 ld4(const float* p) { return *(const float4*)p; }
__device__ __forceinline__ void st4(float* p, const float4 v) { *(float4*)p = v; }

template <bool IS_MAX>
__global__ void __launch_bounds__(256) pool_forw_v4_kernel(const pool_geom_t g, const float* a, float* b, const size_t total)
{
	for (size_t idx64 = (size_t)nnc_xcd_block(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x; idx64 < total; idx64 += (size_t)gridDim.x * blockDim.x) {
		int n, oy, ox, c4;
		unflatten<true>((int)idx64, g.d_oh, g.d_ow, g.d_c4, n, oy, ox, c4);
		int y0 = oy * g.sy - g.pby, x0 = ox * g.sx - g.pbx;
		int y1 = y0 + g.kh, x1 = x0 + g.kw;
		if (y0 < 0) y0 = 0;
		if (x0 < 0) x0 = 0;
		if (y1 > g.H) y1 = g.H;
		if (x1 > g.W) x1 = g.W;
		const float* ap = a + n * g.a_sn + c4 * 4;
		float4 v;
		if (IS_MAX) {
			v = ld4(ap + y0 * g.a_sh + x0 * g.a_sw);
			for (int y = y0; y < y1; y++)
				for (int x = x0; x < x1; x++) {
					const float4 u = ld4(ap + y * g.a_sh + x * g.a_sw);
					if (u.x > v.x) v.x = u.x;
					if (u.y > v.y) v.y = u.y;
					if (u.z > v.z) v.z = u.z;
					if (u.w > v.w) v.w = u.w;
				}
		} else {
			v = make_float4(0.f, 0.f, 0.f, 0.f);
			for (int y = y0; y < y1; y++)
				for (int x = x0; x < x1; x++) {
					const float4 u = ld4(ap + y * g.a_sh + x * g.a_sw);
					v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
				}
			const float cnt = (float)((y1 - y0) * (x1 - x0));
			v.x = v.x / cnt; v.y = v.y / cnt; v.z = v.z / cnt; v.w = v.w / cnt;
		}
		st4(b + n * g.b_sn + oy * g.b_sh + ox * g.b_sw + c4 * 4, v);
	}
}

template <bool IS_MAX>
__global__ void __launch_bounds__(256) pool_back_v4_kernel(const pool_geom_t g, const float* gr, const float* a, const float* b, float* h, const size_t total)
{
	for (size_t idx64 = (size_t)nnc_xcd_block(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x; idx64 < total; idx64 += (size_t)gridDim.x * blockDim.x) {
		int n, y, x, c4;
		unflatten<true>((int)idx64, g.d_h, g.d_w, g.d_c4, n, y, x, c4);
		// outputs whose window covers (y, x): oy in [ceil((y + p - k + 1) / s), floor((y + p) / s)].  The numerators are
		// shifted by k * s to stay non-negative for the multiply-shift division.
		const int ty = y + g.pby, tx = x + g.pbx;
		int oy0 = g.d_sy.div(ty - g.kh + g.kh * g.sy + g.sy) - g.kh, oy1 = g.d_sy.div(ty);   // ceil(a / s) = floor((a + s - 1) / s)
		int ox0 = g.d_sx.div(tx - g.kw + g.kw * g.sx + g.sx) - g.kw, ox1 = g.d_sx.div(tx);
		if (oy0 < 0) oy0 = 0;
		if (ox0 < 0) ox0 = 0;
		if (oy1 > g.OH - 1) oy1 = g.OH - 1;
		if (ox1 > g.OW - 1) ox1 = g.OW - 1;
		const long ob = n * g.b_sn + c4 * 4;
		float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
		float4 av = acc;
		if (IS_MAX) av = ld4(a + n * g.a_sn + y * g.a_sh + x * g.a_sw + c4 * 4);
		for (int oy = oy0; oy <= oy1; oy++)
			for (int ox = ox0; ox <= ox1; ox++) {
				const long o = ob + oy * g.b_sh + ox * g.b_sw;
				const float4 gv = ld4(gr + o);
				if (IS_MAX) {
					const float4 bv = ld4(b + o);
					if (av.x == bv.x) acc.x += gv.x;
					if (av.y == bv.y) acc.y += gv.y;
					if (av.z == bv.z) acc.z += gv.z;
					if (av.w == bv.w) acc.w += gv.w;
				} else {
					int wy0 = oy * g.sy - g.pby, wx0 = ox * g.sx - g.pbx;
					int wy1 = wy0 + g.kh, wx1 = wx0 + g.kw;
					if (wy0 < 0) wy0 = 0;
					if (wx0 < 0) wx0 = 0;
					if (wy1 > g.H) wy1 = g.H;
					if (wx1 > g.W) wx1 = g.W;
					const float cnt = (float)((wy1 - wy0) * (wx1 - wx0));
					acc.x += gv.x / cnt; acc.y += gv.y / cnt; acc.z += gv.z / cnt; acc.w += gv.w / cnt;
				}
			}
		if (IS_MAX && g.relu_mask) {
			if (!(av.x > 0.f)) acc.x = 0.f;
			if (!(av.y > 0.f)) acc.y = 0.f;
			if (!(av.z > 0.f)) acc.z = 0.f;
			if (!(av.w > 0.f)) acc.w = 0.f;
		}
		st4(h + n * g.a_sn + y * g.a_sh + x * g.a_sw + c4 * 4, acc);
	}
}

// ---- windows that TILE the map (stride == window, no border, H = OH * kh, W = OW * kw: the 2 x 2 / 2 pools of VGG-D and the
// CIFAR-10 nets): every input position belongs to exactly one window, so the gradient needs no window search.  A thread per
// window x VEC contiguous channels (NHWC, 16-byte accesses) or per window (NCHW: lanes run along the row): y and g are read once,
// x once, h written once; no divisions per element.  (The general kernels spent 155 us on the DawnNet's 134 MB pools in either
// precision -- instruction-bound -- and ran VGG-D's at 3.4 TB/s.)
template <class T, int VEC> struct packv { typedef T type __attribute__((ext_vector_type(VEC))); };
template <class T> struct packv<T, 1> { typedef T type; };
template <bool NHWC, bool IS_MAX, class T, int VEC>
__global__ void __launch_bounds__(256) pool_back_tiled_kernel(const pool_geom_t g, const FastDiv d_cv, const T* __restrict__ gr, const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ h, const size_t total)
{
	typedef typename packv<T, VEC>::type V;
	const float cnt = (float)(g.kh * g.kw); // (divided by, like the general kernels: bit-identical averages)
	for (size_t idx64 = (size_t)nnc_xcd_block(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x; idx64 < total; idx64 += (size_t)gridDim.x * blockDim.x) {
		int n, oy, ox, c;
		unflatten<NHWC>((int)idx64, g.d_oh, g.d_ow, d_cv, n, oy, ox, c);
		c *= VEC;
		const long o = n * g.b_sn + oy * g.b_sh + ox * g.b_sw + c * g.b_sc;
		float gv[VEC], bv[VEC];
		if constexpr (VEC > 1) {
			const V t = *(const V*)(gr + o);
			for (int e = 0; e < VEC; e++) gv[e] = (float)t[e];
			if (IS_MAX) { const V u = *(const V*)(b + o); for (int e = 0; e < VEC; e++) bv[e] = (float)u[e]; }
		} else {
			gv[0] = (float)gr[o];
			if (IS_MAX) bv[0] = (float)b[o];
		}
		const long i0 = n * g.a_sn + oy * g.kh * g.a_sh + ox * g.kw * g.a_sw + c * g.a_sc;
		for (int dy = 0; dy < g.kh; dy++)
			for (int dx = 0; dx < g.kw; dx++) {
				const long i = i0 + dy * g.a_sh + dx * g.a_sw;
				if constexpr (VEC > 1) {
					V r;
					if (IS_MAX) {
						const V av = *(const V*)(a + i);
						for (int e = 0; e < VEC; e++) r[e] = (T)(((float)av[e] == bv[e]) & (!g.relu_mask | ((float)av[e] > 0.f)) ? gv[e] : 0.f);
					} else
						for (int e = 0; e < VEC; e++) r[e] = (T)(gv[e] / cnt);
					*(V*)(h + i) = r;
				} else {
					if (IS_MAX) { const float av = (float)a[i]; h[i] = (T)((av == bv[0]) & (!g.relu_mask | (av > 0.f)) ? gv[0] : 0.f); }
					else h[i] = (T)(gv[0] / cnt);
				}
			}
		// rows / columns past the last window (VGG-D's 225 x 225 maps under 2 x 2 / 2: one of each) belong to no window: zero gradient.
		// The threads of the last window column / row write them.
		const int ex = g.OW * g.kw, ey = g.OH * g.kh;
		const long z0 = n * g.a_sn + c * g.a_sc;
		if (ox == g.OW - 1 && ex < g.W)
			for (int dy = 0; dy < g.kh; dy++)
				for (int x = ex; x < g.W; x++) {
					const long i = z0 + (oy * g.kh + dy) * g.a_sh + x * g.a_sw;
					if constexpr (VEC > 1) *(V*)(h + i) = V{}; else h[i] = (T)0.f;
				}
		if (oy == g.OH - 1 && ey < g.H)
			for (int y = ey; y < g.H; y++) {
				const int x1 = ox == g.OW - 1 ? g.W : (ox + 1) * g.kw;
				for (int x = ox * g.kw; x < x1; x++) {
					const long i = z0 + y * g.a_sh + x * g.a_sw;
					if constexpr (VEC > 1) *(V*)(h + i) = V{}; else h[i] = (T)0.f;
				}
			}
	}
}
static bool pool_tiles(const pool_geom_t& g)
{
	// windows side by side from the origin; what is left over at the far edges (less than one window) is handled in the kernel
	return g.kh == g.sy && g.kw == g.sx && g.pby == 0 && g.pbx == 0 && g.OH >= 1 && g.OW >= 1 && g.OH == g.H / g.kh && g.OW == g.W / g.kw;
}
template <bool IS_MAX, class T>
static bool pool_back_tiled(const pool_geom_t& g, const bool nhwc, const T* gp, const T* ap, const T* bp, T* hp, const int nn, hipStream_t stream)
{
	if (!pool_tiles(g)) return false;
	constexpr int W = 16 / sizeof(T);
	FastDiv d_cv;
	if (nhwc) {
		const bool vec = g.C % W == 0 && g.a_sc == 1 && g.b_sc == 1 && ((g.a_sn | g.a_sh | g.a_sw | g.b_sn | g.b_sh | g.b_sw) % W) == 0 && aligned16(gp) && aligned16(hp) && (!ap || aligned16(ap)) && (!bp || aligned16(bp));
		if (vec) {
			d_cv.init(g.C / W);
			const size_t total = (size_t)nn * g.OH * g.OW * (g.C / W);
			hipLaunchKernelGGL(HIP_KERNEL_NAME(pool_back_tiled_kernel<true, IS_MAX, T, W>), dim3(grid_for(total, 256)), dim3(256), 0, stream, g, d_cv, gp, ap, bp, hp, total);
		} else {
			d_cv.init(g.C);
			const size_t total = (size_t)nn * g.OH * g.OW * g.C;
			hipLaunchKernelGGL(HIP_KERNEL_NAME(pool_back_tiled_kernel<true, IS_MAX, T, 1>), dim3(grid_for(total, 256)), dim3(256), 0, stream, g, d_cv, gp, ap, bp, hp, total);
		}
	} else {
		d_cv.init(g.C);
		const size_t total = (size_t)nn * g.OH * g.OW * g.C;
		hipLaunchKernelGGL(HIP_KERNEL_NAME(pool_back_tiled_kernel<false, IS_MAX, T, 1>), dim3(grid_for(total, 256)), dim3(256), 0, stream, g, d_cv, gp, ap, bp, hp, total);
	}
	return true;
}

// ---- NCHW rows (round 4): the trainers' tensors are NCHW, where the lanes of the kernels above run along x one element each -- 4 (2) bytes loaded per
// load instruction and lane, the index arithmetic of a whole element (three multiply-shift divisions) per 4 bytes.  Measured in ResNet-50's step at batch 256:
// the stem's 3 x 3 / 2 max-pool gradient took 1.5 ms for 2.05 GB in fp32 AND 1.3 ms for half of that in f16 (instruction-bound).
// In the GRADIENT kernel below a lane owns 4 consecutive x of a row: one 16-byte (8-byte) load of a and store of h, and every output (b, g) that any of the
// four needs is loaded ONCE and offered to all four.  Per element the operations and their order are exactly the kernels' above (each element still sees the
// windows that concern it in raster order), so the results stay bit-identical with the reference.  Measured in the steps (profiles/r04_v6_pool_rows.txt): the stem's
// max-pool gradient 1526 -> 509 us (fp32), 1304 -> 426 us (f16); the DawnNet's 2 x 2 max-pool gradients 60 -> 42 us.  It is NOT taken for average pools whose
// windows tile the map (the window-per-thread kernel above is faster there: 281 vs 636 us), and a forward twin -- four outputs per lane over the union of
// their windows' columns -- was written and removed: slower than a lane per output on three of four shapes (the DawnNet's 2 x 2: 87 vs 35 us).
constexpr int PR = 4; // elements per lane along x
template <class T> struct row4 { typedef T type __attribute__((ext_vector_type(PR))); };

// backward: a lane computes h (n, c, y, x0 .. x0 + 3); W % 4 == 0, a / h rows 4-element aligned.  Every output (oy, ox) whose window covers any of the four is
// visited once, in raster order, and added to the elements it covers.
template <bool IS_MAX, class T>
__global__ void __launch_bounds__(256) pool_back_rows_kernel(const pool_geom_t g, const FastDiv d_w4, const T* __restrict__ gr, const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ h, const size_t total)
{
	typedef typename row4<T>::type V;
	for (size_t idx64 = (size_t)nnc_xcd_block(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x; idx64 < total; idx64 += (size_t)gridDim.x * blockDim.x) {
		int n, y, xq, c;
		unflatten<false>((int)idx64, g.d_h, d_w4, g.d_c, n, y, xq, c);
		const int x0 = xq * PR;
		const int ty = y + g.pby;
		int oy0 = g.d_sy.div(ty - g.kh + g.kh * g.sy + g.sy) - g.kh, oy1 = g.d_sy.div(ty);
		// columns: the union over x0 .. x0 + 3
		int ox0 = g.d_sx.div(x0 + g.pbx - g.kw + g.kw * g.sx + g.sx) - g.kw, ox1 = g.d_sx.div(x0 + PR - 1 + g.pbx);
		if (oy0 < 0) oy0 = 0;
		if (ox0 < 0) ox0 = 0;
		if (oy1 > g.OH - 1) oy1 = g.OH - 1;
		if (ox1 > g.OW - 1) ox1 = g.OW - 1;
		const long ib = n * g.a_sn + c * g.a_sc + y * g.a_sh + x0;
		float av[PR], acc[PR];
		if (IS_MAX) {
			const V t = *(const V*)(a + ib);
#pragma unroll
			for (int e = 0; e < PR; e++) av[e] = (float)t[e];
		}
#pragma unroll
		for (int e = 0; e < PR; e++) acc[e] = 0.f;
		const long ob = n * g.b_sn + c * g.b_sc;
		for (int oy = oy0; oy <= oy1; oy++) {
			float cy = 0.f;
			if (!IS_MAX) { // rows of the clipped window (the divisor of this output's average)
				int wy0 = oy * g.sy - g.pby, wy1 = wy0 + g.kh;
				if (wy0 < 0) wy0 = 0;
				if (wy1 > g.H) wy1 = g.H;
				cy = (float)(wy1 - wy0);
			}
			for (int ox = ox0; ox <= ox1; ox++) {
				const long o = ob + oy * g.b_sh + ox;
				const float gv = (float)gr[o];
				const int wx0 = ox * g.sx - g.pbx, wx1 = wx0 + g.kw; // unclipped: coverage test against in-range x
				float term = gv, bv = 0.f;
				if (IS_MAX) bv = (float)b[o];
				else {
					const int cx0 = wx0 < 0 ? 0 : wx0, cx1 = wx1 > g.W ? g.W : wx1;
					term = gv / (cy * (float)(cx1 - cx0)); // (float)((wy1 - wy0) * (wx1 - wx0)) of the kernels above: a product of two small integers, exact either way
				}
#pragma unroll
				for (int e = 0; e < PR; e++) {
					const bool in = (x0 + e >= wx0) & (x0 + e < wx1);
					if (IS_MAX) acc[e] = (in & (av[e] == bv)) ? acc[e] + gv : acc[e];
					else acc[e] = in ? acc[e] + term : acc[e];
				}
			}
		}
		V r;
#pragma unroll
		for (int e = 0; e < PR; e++) {
			if (IS_MAX && g.relu_mask && !(av[e] > 0.f)) acc[e] = 0.f;
			r[e] = (T)acc[e];
		}
		*(V*)(h + ib) = r;
	}
}
// dense NCHW planes whose rows are whole groups of four (of the walked tensor), 8- / 16-byte aligned
template <class T>
static bool pool_rows_ok(const pool_geom_t& g, const bool nhwc, const int walked_w, const long s_h, const long s_c, const long s_n, const void* vp, const long o_sw, const long i_sw)
{
	return tune(TUNE_POOL_ROWS) && !nhwc && o_sw == 1 && i_sw == 1 && walked_w % PR == 0 && s_h % PR == 0 && s_c % PR == 0 && s_n % PR == 0 && (((uintptr_t)vp) & (PR * sizeof(T) - 1)) == 0;
}

static bool pool_vec4_ok(const pool_geom_t& g, bool nhwc, const void* p0, const void* p1, const void* p2, const void* p3)
{
	if (!nhwc || g.C % 4 || g.a_sc != 1 || g.b_sc != 1) return false;
	if ((g.a_sn | g.a_sh | g.a_sw | g.b_sn | g.b_sh | g.b_sw) % 4) return false;
	return aligned16(p0) && aligned16(p1) && (!p2 || aligned16(p2)) && (!p3 || aligned16(p3));
}

static bool pool_geometry(const ccv_nnc_cmd_t& cmd, const ccv_nnc_hint_t& hint, const ccv_nnc_tensor_t* in_like, const ccv_nnc_tensor_t* out_like, pool_geom_t* g, bool* nhwc)
{
	Image4 a, b;
	if (!image4(in_like, &a) || !image4(out_like, &b)) return false;
	if (in_like->info.format != out_like->info.format) return false;
	if (a.n != b.n || a.c != b.c) return false;
	*nhwc = in_like->info.format == CCV_TENSOR_FORMAT_NHWC;
	g->N = a.n; g->H = a.h; g->W = a.w; g->C = a.c; g->OH = b.h; g->OW = b.w;
	// a window size of 0 means "the whole map" (global pooling, bin/nnc/imagenet.c:92)
	g->kh = cmd.info.size.dim[0] > 0 ? cmd.info.size.dim[0] : a.h;
	g->kw = cmd.info.size.dim[1] > 0 ? cmd.info.size.dim[1] : a.w;
	g->sy = hint.stride.dim[0] > 0 ? hint.stride.dim[0] : 1;
	g->sx = hint.stride.dim[1] > 0 ? hint.stride.dim[1] : 1;
	g->pby = hint.border.begin[0]; g->pbx = hint.border.begin[1];
	g->a_sn = a.sn; g->a_sh = a.sh; g->a_sw = a.sw; g->a_sc = a.sc;
	g->b_sn = b.sn; g->b_sh = b.sh; g->b_sw = b.sw; g->b_sc = b.sc;
	g->d_c.init(g->C); g->d_w.init(g->W); g->d_h.init(g->H); g->d_ow.init(g->OW); g->d_oh.init(g->OH);
	g->d_c4.init(g->C / 4 > 0 ? g->C / 4 : 1); g->d_sy.init(g->sy); g->d_sx.init(g->sx);
	g->relu_mask = 0;
	return true;
}

static bool same_layout(const ccv_nnc_tensor_t* x, const ccv_nnc_tensor_t* like)
{ // dims and strides equal
	Image4 a, b;
	if (!image4(x, &a) || !image4(like, &b) || x->info.format != like->info.format) return false;
	return a.n == b.n && a.h == b.h && a.w == b.w && a.c == b.c && (a.n == 1 || a.sn == b.sn) && a.sh == b.sh && a.sw == b.sw && a.sc == b.sc;
}

template <bool IS_MAX>
static int pool_forw(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	if (input_size < 1 || output_size < 1 || !inputs[0] || !outputs[0]) return CCV_NNC_EXEC_INVALID;
	pool_geom_t g;
	bool nhwc;
	if (!pool_geometry(cmd, hint, inputs[0], outputs[0], &g, &nhwc)) return CCV_NNC_EXEC_INVALID;
	const bool half = CCV_GET_DATA_TYPE(inputs[0]->info.datatype) == CCV_16F;
	if (inputs[0]->info.datatype != outputs[0]->info.datatype) return CCV_NNC_EXEC_INVALID;
	const size_t per_image = (size_t)g.OH * g.OW * g.C;
	if (per_image == 0 || g.N == 0) return CCV_NNC_EXEC_SUCCESS;
	if (per_image >= 0x7fffffffUL) return CCV_NNC_EXEC_INVALID;
	hipStream_t stream = stream_of(stream_context);
	const int nchunk = (int)(0x7fffffffUL / per_image) < g.N ? (int)(0x7fffffffUL / per_image) : g.N; // images per launch (index < 2^31)
	for (int n0 = 0; n0 < g.N; n0 += nchunk) {
		const int nn = g.N - n0 < nchunk ? g.N - n0 : nchunk;
		const size_t total = per_image * nn;
		if (half) {
			const half_t* ap = (const half_t*)inputs[0]->data.f16 + (long)n0 * g.a_sn;
			half_t* bp = (half_t*)outputs[0]->data.f16 + (long)n0 * g.b_sn;
			if (nhwc) hipLaunchKernelGGL(HIP_KERNEL_NAME(pool_forw_kernel<true, IS_MAX, half_t>), dim3(grid_for(total, 256)), dim3(256), 0, stream, g, ap, bp, total);
			else hipLaunchKernelGGL(HIP_KERNEL_NAME(pool_forw_kernel<false, IS_MAX, half_t>), dim3(grid_for(total, 256)), dim3(256), 0, stream, g, ap, bp, total);
			HIP_ENFORCE(hipGetLastError());
			continue;
		}
		const float* ap = inputs[0]->data.f32 + (long)n0 * g.a_sn;
		float* bp = outputs[0]->data.f32 + (long)n0 * g.b_sn;
		if (pool_vec4_ok(g, nhwc, ap, bp, 0, 0)) hipLaunchKernelGGL(HIP_KERNEL_NAME(pool_forw_v4_kernel<IS_MAX>), dim3(grid_for(total / 4, 256)), dim3(256), 0, stream, g, ap, bp, total / 4);
		else if (nhwc) hipLaunchKernelGGL(HIP_KERNEL_NAME(pool_forw_kernel<true, IS_MAX, float>), dim3(grid_for(total, 256)), dim3(256), 0, stream, g, ap, bp, total);
		else hipLaunchKernelGGL(HIP_KERNEL_NAME(pool_forw_kernel<false, IS_MAX, float>), dim3(grid_for(total, 256)), dim3(256), 0, stream, g, ap, bp, total);
		HIP_ENFORCE(hipGetLastError());
	}
	return CCV_NNC_EXEC_SUCCESS;
}

template <bool IS_MAX>
static int pool_back(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	// max: inputs (g, a, b) -> h ; avg: inputs (g, ...) -> h
	if (input_size < 1 || output_size < 1 || !inputs[0] || !outputs[0]) return CCV_NNC_EXEC_INVALID;
	const ccv_nnc_tensor_t* gt = inputs[0];
	ccv_nnc_tensor_t* h = outputs[0];
	const ccv_nnc_tensor_t* a = 0;
	const ccv_nnc_tensor_t* b = 0;
	if (IS_MAX) {
		if (input_size < 3 || !inputs[1] || !inputs[2]) return CCV_NNC_EXEC_INVALID;
		a = inputs[1]; b = inputs[2];
		if (!same_layout(a, h) || !same_layout(b, gt)) return CCV_NNC_EXEC_INVALID;
	}
	pool_geom_t g;
	bool nhwc;
	if (!pool_geometry(cmd, hint, h, gt, &g, &nhwc)) return CCV_NNC_EXEC_INVALID;
	// opt-in: the RELU_BACKWARD that would follow (h = a > 0 ? h : 0, a being that ReLU's output) folded into this command
	const bool relu_mask = IS_MAX && cmd.algorithm > 0 && (cmd.algorithm & NNC_MI355X_POOL_ALGO_FUSE_RELU_BACKWARD);
	if (relu_mask && (flags & CCV_NNC_ACCUMULATE_OUTPUT)) return CCV_NNC_EXEC_INVALID;
	g.relu_mask = relu_mask ? 1 : 0; // every max kernel reads a anyway and masks as it writes: no extra traffic
	const bool half = CCV_GET_DATA_TYPE(h->info.datatype) == CCV_16F;
	if (gt->info.datatype != h->info.datatype || (a && a->info.datatype != h->info.datatype) || (b && b->info.datatype != h->info.datatype)) return CCV_NNC_EXEC_INVALID;
	const size_t per_image = (size_t)g.H * g.W * g.C;
	if (per_image == 0 || g.N == 0) return CCV_NNC_EXEC_SUCCESS;
	if (per_image >= 0x7fffffffUL) return CCV_NNC_EXEC_INVALID;
	hipStream_t stream = stream_of(stream_context);
	const int nchunk = (int)(0x7fffffffUL / per_image) < g.N ? (int)(0x7fffffffUL / per_image) : g.N;
	for (int n0 = 0; n0 < g.N; n0 += nchunk) {
		const int nn = g.N - n0 < nchunk ? g.N - n0 : nchunk;
		const size_t total = per_image * nn;
		if (half) {
			const half_t* gp = (const half_t*)gt->data.f16 + (long)n0 * g.b_sn;
			const half_t* ap = a ? (const half_t*)a->data.f16 + (long)n0 * g.a_sn : 0;
			const half_t* bp = b ? (const half_t*)b->data.f16 + (long)n0 * g.b_sn : 0;
			half_t* hp = (half_t*)h->data.f16 + (long)n0 * g.a_sn;
			if ((IS_MAX || !pool_tiles(g)) && pool_rows_ok<half_t>(g, nhwc, g.W, g.a_sh, g.a_sc, g.a_sn, hp, g.a_sw, g.b_sw) && (!ap || (((uintptr_t)ap) & (PR * sizeof(half_t) - 1)) == 0)) {
				FastDiv d4; d4.init(g.W / PR);
				hipLaunchKernelGGL(HIP_KERNEL_NAME(pool_back_rows_kernel<IS_MAX, half_t>), dim3(grid_for(total / PR, 256)), dim3(256), 0, stream, g, d4, gp, ap, bp, hp, total / PR);
				HIP_ENFORCE(hipGetLastError());
				continue;
			}
			if (pool_back_tiled<IS_MAX, half_t>(g, nhwc, gp, ap, bp, hp, nn, stream)) { HIP_ENFORCE(hipGetLastError()); continue; }
			if (nhwc) hipLaunchKernelGGL(HIP_KERNEL_NAME(pool_back_kernel<true, IS_MAX, half_t>), dim3(grid_for(total, 256)), dim3(256), 0, stream, g, gp, ap, bp, hp, total);
			else hipLaunchKernelGGL(HIP_KERNEL_NAME(pool_back_kernel<false, IS_MAX, half_t>), dim3(grid_for(total, 256)), dim3(256), 0, stream, g, gp, ap, bp, hp, total);
			HIP_ENFORCE(hipGetLastError());
			continue;
		}
		const float* gp = gt->data.f32 + (long)n0 * g.b_sn;
		const float* ap = a ? a->data.f32 + (long)n0 * g.a_sn : 0;
		const float* bp = b ? b->data.f32 + (long)n0 * g.b_sn : 0;
		float* hp = h->data.f32 + (long)n0 * g.a_sn;
		if ((IS_MAX || !pool_tiles(g)) && pool_rows_ok<float>(g, nhwc, g.W, g.a_sh, g.a_sc, g.a_sn, hp, g.a_sw, g.b_sw) && (!ap || aligned16(ap))) {
			FastDiv d4; d4.init(g.W / PR);
			hipLaunchKernelGGL(HIP_KERNEL_NAME(pool_back_rows_kernel<IS_MAX, float>), dim3(grid_for(total / PR, 256)), dim3(256), 0, stream, g, d4, gp, ap, bp, hp, total / PR);
			HIP_ENFORCE(hipGetLastError());
			continue;
		}
		if (pool_back_tiled<IS_MAX, float>(g, nhwc, gp, ap, bp, hp, nn, stream)) { HIP_ENFORCE(hipGetLastError()); continue; }
		if (pool_vec4_ok(g, nhwc, gp, hp, ap, bp)) hipLaunchKernelGGL(HIP_KERNEL_NAME(pool_back_v4_kernel<IS_MAX>), dim3(grid_for(total / 4, 256)), dim3(256), 0, stream, g, gp, ap, bp, hp, total / 4);
		else if (nhwc) hipLaunchKernelGGL(HIP_KERNEL_NAME(pool_back_kernel<true, IS_MAX, float>), dim3(grid_for(total, 256)), dim3(256), 0, stream, g, gp, ap, bp, hp, total);
		else hipLaunchKernelGGL(HIP_KERNEL_NAME(pool_back_kernel<false, IS_MAX, float>), dim3(grid_for(total, 256)), dim3(256), 0, stream, g, gp, ap, bp, hp, total);
		HIP_ENFORCE(hipGetLastError());
	}
	return CCV_NNC_EXEC_SUCCESS;
}

static int _max_pool_forw(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{ return pool_forw<true>(cmd, hint, flags, inputs, input_size, outputs, output_size, stream_context); }
static int _max_pool_back(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	uint64_t sig; // recorded: the RELU_BACKWARD of the pooled map may follow on the gradient this writes (peephole.cpp)
	if (const int e = deferred_take_error(stream_context)) return e; // a recorded command failed when a flush launched it (peephole.cpp)
	if (deferred_try(_max_pool_back, DEFER_POOL_BACKWARD, cmd, hint, flags, inputs, input_size, outputs, output_size, stream_context, &sig)) return CCV_NNC_EXEC_SUCCESS;
	const int r = pool_back<true>(cmd, hint, flags, inputs, input_size, outputs, output_size, stream_context);
	if (r == CCV_NNC_EXEC_SUCCESS) deferred_mark_good(sig);
	return r;
}
static int _avg_pool_forw(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{ return pool_forw<false>(cmd, hint, flags, inputs, input_size, outputs, output_size, stream_context); }
static int _avg_pool_back(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{ return pool_back<false>(cmd, hint, flags, inputs, input_size, outputs, output_size, stream_context); }

} // namespace

#define NNC_REG(CMD, BACKEND, EXEC) \
	extern "C" void _register_command_##CMD##_backend_##BACKEND(ccv_nnc_cmd_backend_registry_t* const registry) \
	{ registry->tensor_formats = CCV_TENSOR_FORMAT_NCHW | CCV_TENSOR_FORMAT_NHWC; registry->tensor_datatypes = CCV_32F; registry->tensor_memory = CCV_TENSOR_GPU_MEMORY; registry->algorithms = 1; registry->exec = EXEC; NNC_HALF_STAGED(registry, EXEC); }
NNC_REG(CCV_NNC_MAX_POOL_FORWARD, CCV_NNC_BACKEND_GPU_CUDNN, _max_pool_forw)
NNC_REG(CCV_NNC_MAX_POOL_BACKWARD, CCV_NNC_BACKEND_GPU_CUDNN, _max_pool_back)
NNC_REG(CCV_NNC_AVERAGE_POOL_FORWARD, CCV_NNC_BACKEND_GPU_CUDNN, _avg_pool_forw)
NNC_REG(CCV_NNC_AVERAGE_POOL_BACKWARD, CCV_NNC_BACKEND_GPU_CUDNN, _avg_pool_back)
