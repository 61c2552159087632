// The classic image pre-process loops that feed the trainer, as batch HIP kernels behind a C-ABI:
//   nnc_mi355x_resample_batch  <- ccv_resample   lib/ccv_resample.c:433-478 (dispatch), :11-133 (area, 8-bit fixed point),
//                                                :135-248 (area, float), :266-431 (bicubic, float / integer)
//   nnc_mi355x_filter_batch    <- ccv_filter     lib/ccv_numeric.c:1036-1061 (dispatch), :960-1034 (direct 8-bit path)
// A whole batch of same-sized images resident in HBM is processed by ONE launch (the reference walks one image at a time on
// a CPU worker thread of the dataframe); both are HBM-bound byte / float streaming kernels: one lane per output element,
// consecutive lanes on consecutive output elements (channel fastest), source taps gathered through L2.
//
// Parity: the reference's row-sequential state machines are restated as per-output-row / per-output-column TAP TABLES built
// on the host with the reference's own double -> fixed-point formulas, so that
//   * area 8u -> 8u is integer-only on the device and BIT-EXACT (uint32 sums are order-independent);
//   * float area / bicubic replay the reference's accumulation order per element (horizontal taps first, then rows);
//   * the direct 8-bit filter is integer-only and bit-exact (replicated border, 2^14 fixed-point coefficients).
// The float filter path of the reference goes through a TILED FFT (_ccv_filter_kissfft, ccv_numeric.c:771-): the kernel is flipped into a tile, the tile is
// filled with an image window (zeros where it leaves the image), convolved circularly, and a block of the result is copied out at an offset of size / 2 --
//   d[y][x] = sum_{i,j} a[y + i - (kh - 1) / 2][x + j - (kw - 1) / 2] * b[i][j],   a = 0 outside the image
// (centre tap (size - 1) / 2 for even AND odd windows: test/unit/numeric.tests.c:112-150) wherever window + kernel fit the tile; at the borders of an image that
// needs several tiles the far rows / columns of a tile's window wrap in.  Round 5: filter_f32_kernel reproduces that tiling (per-row / per-column maps of window
// origin, extent and read position, filter_axis_map), so the whole image agrees with the reference, borders included -- the same sums, added directly.
#include "common.h"
#include "isa.h"
#include <math.h>
#include <pthread.h>
#include <vector>

using namespace nnc;

namespace {

struct tap_u32_t { int si; unsigned w; };  // source index (element offset within a row, or a row number) and weight
struct tap_f32_t { int si; float w; };

// ------------------------------------------------------------------------------------------------ area, 8u -> 8u
// out[dy][dx*ch + c] = min(255, (sum_{(sy, wy) in Y[dy]} wy * (sum_{(sx, wx) in X[dx]} wx * a[sy][sx + c])) / inv_scale_256)
__global__ void __launch_bounds__(256) area_8u_kernel(const unsigned char* a, unsigned char* b, const long a_step, const long a_image, const long b_step, const long b_image,
	const int b_rows, const int b_cols_ch, const int ch, const int* xstart, const tap_u32_t* xtaps, const int* ystart, const tap_u32_t* ytaps, const unsigned inv_scale_256, const size_t total)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
		const int e = (int)(idx % b_cols_ch);
		const size_t r = idx / b_cols_ch;
		const int dy = (int)(r % b_rows);
		const size_t img = r / b_rows;
		const int dx = e / ch, c = e - dx * ch;
		const unsigned char* ai = a + img * a_image;
		unsigned acc = 0;
		for (int ky = ystart[dy]; ky < ystart[dy + 1]; ky++) {
			const unsigned char* row = ai + (long)ytaps[ky].si * a_step + c;
			unsigned h = 0;
			for (int kx = xstart[dx]; kx < xstart[dx + 1]; kx++) h += row[xtaps[kx].si] * xtaps[kx].w;
			acc += h * ytaps[ky].w;
		}
		const unsigned v = acc / inv_scale_256;
		b[img * b_image + (long)dy * b_step + e] = (unsigned char)(v > 255 ? 255 : v);
	}
}

// The same arithmetic, ONE WORKGROUP PER GROUP OF OUTPUT ROWS (round 5; VERDICT round 4 item 8: the kernel above is one lane per output byte with scalar byte gathers,
// 0.45 TB/s on 256 x 480^2 -> 224^2).  uint32 sums are a ring: sum_y wy (sum_x wx a) = sum_x wx (sum_y wy a) bit for bit, overflow included, so the rows are
// reduced FIRST -- every lane loads VEC bytes (16 when base and pitches allow, else 4) of each of the output row's source rows, coalesced, and leaves VEC
// column sums in LDS -- and the horizontal taps then read LDS instead of gathering bytes through L2; four output bytes leave a lane as one dword.  A source row
// is fetched once per output row that taps it (neighbouring rows' workgroups find it in L2).  The quotient acc / inv_scale_256 is at most a few hundred: a float
// estimate corrected by one exact multiply-compare replaces the 32-bit division.  Needs: 4-byte aligned source rows and destination rows, channels 1 / 3 / 4.
// where column sum x of a row lives in LDS: a lane's sixteen sums are four 16-byte pieces; piece k of lane u sits at piece slot k ^ ((u >> 1) & 3)
template <int VEC> __host__ __device__ __forceinline__ int area_col(const int x) { return VEC == 16 ? (x & ~12) | ((((x >> 2) ^ (x >> 5)) & 3) << 2) : x; }
// R consecutive output rows per workgroup: the kernel is bound by the LATENCY of its dependent loads (tap tables -> source rows -> LDS -> taps again), not by
// bytes or arithmetic -- one row per workgroup ran at 1.3 TB/s with every pipe idle most of the time -- so a workgroup issues the source loads of R rows back
// to back and applies a lane's horizontal taps (loaded once) to all R rows' column sums.  (Measured and not adopted: dealing the (row, unit) and (element
// group, row pair) tasks over ALL lanes -- a third of them idle in both stages at 480 -> 224 -- is slower, 0.105 / 0.092 vs 0.087 ms: what a lane has in flight
// counts, not how many lanes work.)
struct etap_t { unsigned off, w; };
template <int VEC, int R>
__global__ void __launch_bounds__(128) area_8u_rows_kernel(const unsigned char* __restrict__ a, unsigned char* __restrict__ b, const long a_step, const long a_image, const long b_step, const long b_image,
	const int b_rows, const int b_cols_ch, const int a_cols_ch, const int row_groups, const etap_t* __restrict__ etab, const int e_pitch, const int maxt, const int* __restrict__ ystart, const tap_u32_t* __restrict__ ytaps, const double inv_scale_rcp, const double inv_scale_half)
{
	HIP_DYNAMIC_SHARED(unsigned, area_v) // column sums of the R output rows: R x pitch (pitch = a_cols_ch rounded up to whole lanes' worth)
	const int dy0 = (int)(blockIdx.x % (unsigned)row_groups) * R;
	const size_t img = blockIdx.x / (unsigned)row_groups;
	const unsigned char* const ai = a + img * a_image;
	const int units = (a_cols_ch + VEC - 1) / VEC, pitch = units * VEC;
	int ky0[R], ky1[R];
#pragma unroll
	for (int r = 0; r < R; r++) { const int dy = dy0 + r < b_rows ? dy0 + r : b_rows - 1; ky0[r] = ystart[dy]; ky1[r] = ystart[dy + 1]; }
	for (int u = (int)threadIdx.x; u < units; u += 128) {
#pragma unroll
		for (int r = 0; r < R; r++) {
			unsigned acc[VEC];
#pragma unroll
			for (int j = 0; j < VEC; j++) acc[j] = 0;
			for (int ky = ky0[r]; ky < ky1[r]; ky++) {
				const unsigned char* const row = ai + (long)ytaps[ky].si * a_step + (long)u * VEC;
				const unsigned w = ytaps[ky].w;
				if (VEC == 16) {
					const uint4 q = *(const uint4*)row; // (a row's last unit stays inside the row pitch: pitches are multiples of 16 here)
					const unsigned d[4] = { q.x, q.y, q.z, q.w };
#pragma unroll
					for (int j = 0; j < 16; j++) acc[j] += nnc_mul24((d[j >> 2] >> (8 * (j & 3))) & 0xffu, w); // (weights are at most 256: the 24-bit multiply-add, not the quarter-rate 32-bit one)
				} else {
					const unsigned q = *(const unsigned*)row;
#pragma unroll
					for (int j = 0; j < 4; j++) acc[j] += nnc_mul24((q >> (8 * j)) & 0xffu, w);
				}
			}
			// (16 sums per lane are four 16-byte pieces 64 bytes apart from the neighbouring lane's: unswizzled, the eight lanes of a ds_write_b128 group land on two
			// bank quads -- the counters showed 85 % of the LDS cycles of this kernel were conflicts; area_col() spreads the pieces over all eight quads)
#pragma unroll
			for (int j = 0; j < VEC; j += 4) *(uint4*)(area_v + r * pitch + area_col<VEC>(u * VEC + j)) = uint4{ acc[j], acc[j + 1], acc[j + 2], acc[j + 3] };
		}
	}
	__syncthreads();
	// horizontal taps per OUTPUT ELEMENT, structure of arrays (etab[k][e] = {LDS column -- swizzled for this VEC --, weight}; missing taps have weight 0): a
	// lane's four consecutive elements are 32 contiguous bytes per tap index -- two 16-byte loads, coalesced across the wave --, read once and applied to all R rows
	for (int t4 = (int)threadIdx.x * 4; t4 < b_cols_ch; t4 += 128 * 4) {
		unsigned h[R][4];
#pragma unroll
		for (int r = 0; r < R; r++)
#pragma unroll
			for (int j = 0; j < 4; j++) h[r][j] = 0;
		for (int k = 0; k < maxt; k++) {
			const uint4* const tp = (const uint4*)(etab + (size_t)k * e_pitch + t4); // (e_pitch is a multiple of 4 elements: whole 32-byte groups, in range)
			const uint4 t01 = tp[0], t23 = tp[1];
			const unsigned off[4] = { t01.x, t01.z, t23.x, t23.z }, w[4] = { t01.y, t01.w, t23.y, t23.w };
#pragma unroll
			for (int j = 0; j < 4; j++)
#pragma unroll
				for (int r = 0; r < R; r++) h[r][j] += nnc_mul24(area_v[r * pitch + off[j]], w[j]); // (a column sum is < 2^24: 255 x the row weights, whose sum is at most 2^16)
		}
#pragma unroll
		for (int r = 0; r < R; r++) {
			if (dy0 + r >= b_rows) break;
			unsigned packed = 0;
#pragma unroll
			for (int j = 0; j < 4; j++) {
				// floor(h / d) exactly, in three fp64 instructions: (h + 1/2) / d is never an integer and sits at least 1 / (2 d) > 2^-25 away from one, the double
				// product's error is below 2^-52 of the quotient (< 2^16): truncation cannot land on the wrong side.  (The 32-bit division is ~40 instructions, a float
				// estimate needs a multiply-compare correction: 11.)
				const unsigned q = (unsigned)__builtin_fma((double)h[r][j], inv_scale_rcp, inv_scale_half);
				packed |= (q > 255 ? 255u : q) << (8 * j);
			}
			unsigned char* const brow = b + img * b_image + (long)(dy0 + r) * b_step;
			if (t4 + 4 <= b_cols_ch) *(unsigned*)(brow + t4) = packed;
			else for (int j = 0; t4 + j < b_cols_ch; j++) brow[t4 + j] = (unsigned char)(packed >> (8 * j));
		}
	}
}

template <typename T> __device__ __forceinline__ float ld_as_float(const void* p, long i) { return (float)((const T*)p)[i]; }
__device__ __forceinline__ void st_from_float(unsigned char* p, long i, float v) { int t = (int)v; p[i] = (unsigned char)(t < 0 ? 0 : t > 255 ? 255 : t); } // _ccv_set_8u_value: truncate, clamp
__device__ __forceinline__ void st_from_float(float* p, long i, float v) { p[i] = v; }

// ------------------------------------------------------------------------------------------------ area, float accumulate
template <typename TA, typename TB>
__global__ void __launch_bounds__(256) area_f32_kernel(const unsigned char* a, unsigned char* b, const long a_step, const long a_image, const long b_step, const long b_image,
	const int b_rows, const int b_cols_ch, const int ch, const int* xstart, const tap_f32_t* xtaps, const int* ystart, const tap_f32_t* ytaps, const size_t total)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
		const int e = (int)(idx % b_cols_ch);
		const size_t r = idx / b_cols_ch;
		const int dy = (int)(r % b_rows);
		const size_t img = r / b_rows;
		const int dx = e / ch, c = e - dx * ch;
		const unsigned char* ai = a + img * a_image;
		float acc = 0.f;
		for (int ky = ystart[dy]; ky < ystart[dy + 1]; ky++) {
			const unsigned char* row = ai + (long)ytaps[ky].si * a_step;
			float h = 0.f;
			for (int kx = xstart[dx]; kx < xstart[dx + 1]; kx++) h += ld_as_float<TA>(row, xtaps[kx].si + c) * xtaps[kx].w;
			acc += h * ytaps[ky].w;
		}
		st_from_float((TB*)(b + img * b_image + (long)dy * b_step), e, acc);
	}
}

// ------------------------------------------------------------------------------------------------ bicubic
struct cubic_f_t { int si[4]; float w[4]; };
struct cubic_i_t { int si[4]; int w[4]; };
// float-only variant (output 32F): rows are first reduced horizontally into the output type, then combined vertically
template <typename TA>
__global__ void __launch_bounds__(256) cubic_f32_kernel(const unsigned char* a, unsigned char* b, const long a_step, const long a_image, const long b_step, const long b_image,
	const int b_rows, const int b_cols_ch, const int ch, const cubic_f_t* xofs, const cubic_f_t* yofs, const size_t total)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
		const int e = (int)(idx % b_cols_ch);
		const size_t r = idx / b_cols_ch;
		const int dy = (int)(r % b_rows);
		const size_t img = r / b_rows;
		const int dx = e / ch, c = e - dx * ch;
		const cubic_f_t xo = xofs[dx], yo = yofs[dy];
		const unsigned char* ai = a + img * a_image;
		float rows[4];
#pragma unroll
		for (int k = 0; k < 4; k++) {
			const unsigned char* row = ai + (long)yo.si[k] * a_step;
			rows[k] = ld_as_float<TA>(row, xo.si[0] * ch + c) * xo.w[0] + ld_as_float<TA>(row, xo.si[1] * ch + c) * xo.w[1] + ld_as_float<TA>(row, xo.si[2] * ch + c) * xo.w[2] + ld_as_float<TA>(row, xo.si[3] * ch + c) * xo.w[3];
		}
		((float*)(b + img * b_image + (long)dy * b_step))[e] = rows[0] * yo.w[0] + rows[1] * yo.w[1] + rows[2] * yo.w[2] + rows[3] * yo.w[3];
	}
}
// integer-only variant (8u source, 8u output): 6-bit coefficients both ways, descale by 12, clamp
__global__ void __launch_bounds__(256) cubic_8u_kernel(const unsigned char* a, unsigned char* b, const long a_step, const long a_image, const long b_step, const long b_image,
	const int b_rows, const int b_cols_ch, const int ch, const cubic_i_t* xofs, const cubic_i_t* yofs, const size_t total)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
		const int e = (int)(idx % b_cols_ch);
		const size_t r = idx / b_cols_ch;
		const int dy = (int)(r % b_rows);
		const size_t img = r / b_rows;
		const int dx = e / ch, c = e - dx * ch;
		const cubic_i_t xo = xofs[dx], yo = yofs[dy];
		const unsigned char* ai = a + img * a_image;
		int v = 0;
#pragma unroll
		for (int k = 0; k < 4; k++) {
			const unsigned char* row = ai + (long)yo.si[k] * a_step + c;
			const int h = row[xo.si[0] * ch] * xo.w[0] + row[xo.si[1] * ch] * xo.w[1] + row[xo.si[2] * ch] * xo.w[2] + row[xo.si[3] * ch] * xo.w[3];
			v += h * yo.w[k];
		}
		v = (v + (1 << 11)) >> 12; // ccv_descale(x, 12)
		b[img * b_image + (long)dy * b_step + e] = (unsigned char)(v < 0 ? 0 : v > 255 ? 255 : v);
	}
}

// ------------------------------------------------------------------------------------------------ direct 8-bit filter
// d[y][x] = clamp((sum_{i,j} a[clamp(y + i - kh/2)][clamp(x + j - kw/2)] * coeff[i][j]) >> 14), single channel
__global__ void __launch_bounds__(256) filter_8u_kernel(const unsigned char* a, unsigned char* d, const long a_step, const long a_image, const long d_step, const long d_image,
	const int rows, const int cols, const int* coeff, const int kh, const int kw, const size_t total)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
		const int x = (int)(idx % cols);
		const size_t r = idx / cols;
		const int y = (int)(r % rows);
		const size_t img = r / rows;
		const unsigned char* ai = a + img * a_image;
		int z = 0;
		for (int i = 0; i < kh; i++) {
			int sy = y + i - kh / 2;
			sy = sy < 0 ? 0 : sy > rows - 1 ? rows - 1 : sy;
			const unsigned char* row = ai + (long)sy * a_step;
			for (int j = 0; j < kw; j++) {
				int sx = x + j - kw / 2;
				sx = sx < 0 ? 0 : sx > cols - 1 ? cols - 1 : sx;
				z += row[sx] * coeff[i * kw + j];
			}
		}
		z >>= 14;
		d[img * d_image + (long)y * d_step + x] = (unsigned char)(z < 0 ? 0 : z > 255 ? 255 : z);
	}
}
// float correlation with the reference's TILED circular semantics (round 5; _ccv_filter_kissfft, lib/ccv_numeric.c:771-958).  The reference cuts the image into
// tiles of `trows x tcols`, fills a tile with the image window at (oy, ox) -- zeros where the window leaves the image --, convolves it CIRCULARLY with the
// flipped kernel and copies a block of the result to the output.  Which tile serves an output row and where in the tile's result the row is read are functions
// of the row alone (the same for columns), so the host hands over two small maps -- {window origin, rows of the window inside the image, row of the circular
// result} per output row and per output column (filter_axis_map below replays the reference's loops, later tiles overwriting earlier ones as they do) -- and
// an output element is   sum_{i, j} b[i][j] * T[(p_y - (kh - 1) + i) mod trows][(p_x - (kw - 1) + j) mod tcols],   T = the zero-filled window.
// Where a window plus the kernel fits the tile this is the plain correlation with a zero border and centre tap (size - 1) / 2; at the borders of an image that
// needs several tiles the window's far rows / columns wrap in, exactly as in the reference.  Same sums, added directly instead of through an FFT (fp32 rounding apart).
struct filter_map_t { int o, end, p; }; // window origin, window extent inside the image, index into the circular result; p < 0: the reference writes nothing there
__global__ void __launch_bounds__(256) filter_f32_kernel(const float* a, float* d, const long a_step, const long a_image, const long d_step, const long d_image,
	const int rows, const int cols, const int ch, const float* coeff, const int kh, const int kw, const int kch, const filter_map_t* rowmap, const filter_map_t* colmap, const int trows, const int tcols, const size_t total)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
		const int e = (int)(idx % ((size_t)cols * ch));
		const size_t r = idx / ((size_t)cols * ch);
		const int y = (int)(r % rows);
		const size_t img = r / rows;
		const int x = e / ch, c = e - x * ch;
		const float* ai = (const float*)((const char*)a + img * a_image);
		const filter_map_t rm = rowmap[y], cm = colmap[x];
		float z = 0.f;
		if (rm.p >= 0 && cm.p >= 0)
			for (int i = 0; i < kh; i++) {
				int ty = rm.p - (kh - 1) + i;
				if (ty < 0) ty += trows; // (p < trows and kh <= trows: one wrap at most)
				if (ty >= rm.end) continue;
				const float* row = (const float*)((const char*)ai + (long)(rm.o + ty) * a_step);
				for (int j = 0; j < kw; j++) {
					int tx = cm.p - (kw - 1) + j;
					if (tx < 0) tx += tcols;
					if (tx >= cm.end) continue;
					z += row[(cm.o + tx) * ch + c] * coeff[(i * kw + j) * kch + (kch > 1 ? c : 0)];
				}
			}
		((float*)((char*)d + img * d_image + (long)y * d_step))[e] = z;
	}
}

// ------------------------------------------------------------------------------------------------ host: tap tables
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

// Horizontal taps of the area resample (lib/ccv_resample.c:35-63 / :157-184): left partial, full columns, right partial.
template <class TAP, class W>
static void area_x_taps(int a_cols, int b_cols, int ch, double scale_x, W full, double unit, std::vector<int>& start, std::vector<TAP>& taps)
{
	start.assign(b_cols + 1, 0);
	for (int dx = 0; dx < b_cols; dx++) {
		start[dx] = (int)taps.size();
		const double fsx1 = dx * scale_x, fsx2 = fsx1 + scale_x;
		const int sx1 = (int)(fsx1 + 1.0 - 1e-6), sx2 = (int)(fsx2);
		if (sx1 > fsx1) { TAP t; t.si = imin(sx1 - 1, a_cols - 1) * ch; t.w = (W)((sx1 - fsx1) * unit); taps.push_back(t); }
		for (int sx = sx1; sx < sx2; sx++) { TAP t; t.si = imin(sx, a_cols - 1) * ch; t.w = full; taps.push_back(t); }
		if (fsx2 - sx2 > 1e-3) { TAP t; t.si = imin(sx2, a_cols - 1) * ch; t.w = (W)((fsx2 - sx2) * unit); taps.push_back(t); }
	}
	start[b_cols] = (int)taps.size();
}

// Vertical taps of the 8-bit area resample: the reference's row-sequential state machine (lib/ccv_resample.c:68-131) replayed
// symbolically -- `sum` is kept as a list of (source row, weight) instead of a value.
static void area_y_taps_8u(int a_rows, int b_rows, double scale_y, std::vector<int>& start, std::vector<tap_u32_t>& taps)
{
	// (output rows are finished in increasing order, so a row's taps -- the carried list `sum` plus the closing tap -- are appended as the row closes: no
	// per-row lists; round 5: the vector of vectors this replaces was 224 allocations per image in the jitter path)
	tap_u32_t sum[2 + 256];
	int nsum = 0;
	start.assign(b_rows + 1, 0);
	int dy = 0, dy_weight_256 = 0;
	for (int sy = 0; sy < a_rows; sy++) {
		if (dy < b_rows && (dy + 1) * scale_y <= sy + 1) {
			unsigned beta = (unsigned)(int)(fmax(sy + 1 - (dy + 1) * scale_y, 0.f) * 256);
			const unsigned beta1 = 256 - beta;
			if (sy == a_rows - 1) beta = (unsigned)(int)(scale_y * 256);
			else dy_weight_256 = (int)beta;
			start[dy] = (int)taps.size();
			taps.insert(taps.end(), sum, sum + nsum);
			tap_u32_t t; t.si = sy;
			nsum = 0;
			if ((int)beta <= 0) { t.w = 256; taps.push_back(t); }
			else { t.w = beta1; taps.push_back(t); t.w = beta; sum[nsum++] = t; }
			dy++;
		} else if (dy >= b_rows) {
			// the reference would write past b here; cannot happen for the scales it is called with (rows_scale = b/a)
			break;
		} else {
			tap_u32_t t; t.si = sy;
			if (sy == a_rows - 1) { dy_weight_256 = (int)(scale_y * 256) - dy_weight_256; t.w = (unsigned)dy_weight_256; }
			else { dy_weight_256 += 256; t.w = 256; }
			if (nsum < (int)(sizeof(sum) / sizeof(sum[0]))) sum[nsum++] = t;
		}
	}
	for (; dy < b_rows; dy++) { start[dy] = (int)taps.size(); taps.insert(taps.end(), sum, sum + nsum); } // :125-130
	start[b_rows] = (int)taps.size();
}
// The float variant (lib/ccv_resample.c:186-243): weights 1 for full rows, beta carried into the next output row.
static void area_y_taps_f32(int a_rows, int b_rows, double scale_y, std::vector<int>& start, std::vector<tap_f32_t>& taps)
{
	std::vector<tap_f32_t>& sum = *([]() { static thread_local std::vector<tap_f32_t> v; return &v; })();
	sum.clear();
	start.assign(b_rows + 1, 0);
	int dy = 0;
	float dy_weight = 0;
	for (int sy = 0; sy < a_rows; sy++) {
		if (dy < b_rows && (dy + 1) * scale_y <= sy + 1) {
			float beta = (float)fmax(sy + 1 - (dy + 1) * scale_y, 0.f);
			const float beta1 = 1 - beta;
			if (sy == a_rows - 1) beta = (float)scale_y;
			else dy_weight = beta;
			start[dy] = (int)taps.size();
			taps.insert(taps.end(), sum.begin(), sum.end());
			tap_f32_t t; t.si = sy;
			sum.clear();
			if (fabsf(beta) < 1e-3f) { t.w = 1.f; taps.push_back(t); }
			else { t.w = beta1; taps.push_back(t); t.w = beta; sum.push_back(t); }
			dy++;
		} else if (dy >= b_rows) break;
		else {
			tap_f32_t t; t.si = sy;
			if (sy == a_rows - 1) { dy_weight = (float)(scale_y - dy_weight); t.w = dy_weight; }
			else { dy_weight += 1; t.w = 1.f; }
			sum.push_back(t);
		}
	}
	for (; dy < b_rows; dy++) { start[dy] = (int)taps.size(); taps.insert(taps.end(), sum.begin(), sum.end()); }
	start[b_rows] = (int)taps.size();
}

// Bicubic coefficients, A = -0.75 (lib/ccv_resample.c:266-279 float, :343-357 6-bit integer).
static void cubic_f(int si, int sz, float s, cubic_f_t* c)
{
	const float A = -0.75f;
	c->si[0] = imin(imax(si - 1, 0), sz - 1); c->si[1] = imin(imax(si, 0), sz - 1); c->si[2] = imin(imax(si + 1, 0), sz - 1); c->si[3] = imin(imax(si + 2, 0), sz - 1);
	const float x = s - si;
	c->w[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
	c->w[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
	c->w[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
	c->w[3] = 1.f - c->w[0] - c->w[1] - c->w[2];
}
static void cubic_i(int si, int sz, float s, cubic_i_t* c)
{
	const float A = -0.75f;
	c->si[0] = imin(imax(si - 1, 0), sz - 1); c->si[1] = imin(imax(si, 0), sz - 1); c->si[2] = imin(imax(si + 1, 0), sz - 1); c->si[3] = imin(imax(si + 2, 0), sz - 1);
	const float x = s - si;
	const int W_BITS = 1 << 6;
	c->w[0] = (int)((((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A) * W_BITS + 0.5);
	c->w[1] = (int)((((A + 2) * x - (A + 3)) * x * x + 1) * W_BITS + 0.5);
	c->w[2] = (int)((((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1) * W_BITS + 0.5);
	c->w[3] = W_BITS - c->w[0] - c->w[1] - c->w[2];
}

// Tables are tiny (a few KB): packed behind each other in the stream workspace and uploaded with ONE stream-ordered copy
// (ordered behind the kernels still reading the previous tables).  The source is ordinary host memory kept alive in a small
// ring until long after the runtime has staged it -- no host callback, no free from a HIP callback thread (HIP API calls
// from stream callbacks deadlock).
struct upload_t {
	std::vector<char> host;
	size_t add(const void* p, size_t bytes) { const size_t off = (host.size() + 15) & ~(size_t)15; host.resize(off + bytes); memcpy(host.data() + off, p, bytes); return off; }
};
static char* upload(upload_t& u, ccv_nnc_stream_context_t* ctx)
{
	static thread_local std::vector<char> ring[16];
	static thread_local unsigned ring_at = 0;
	char* dev = (char*)workspace_of(ctx, u.host.size());
	if (!dev) return 0;
	std::vector<char>& keep = ring[ring_at++ & 15];
	keep.swap(u.host);
	HIP_ENFORCE(hipMemcpyAsync(dev, keep.data(), keep.size(), hipMemcpyHostToDevice, stream_of(ctx)));
	return dev;
}

// ------------------------------------------------------------------------------------------------ random-jitter batch
// The pixel half of _ccv_cnnp_random_jitter (lib/nnc/ccv_cnnp_dataframe_addons.c:265-366) for a whole batch in ONE kernel: per
// image the host has decided (the reference's own integer arithmetic on its generator's draws, :276-330) the source slice, the
// size it is resampled to, the mirror flag and the crop window; the device resamples (area when shrinking, bicubic otherwise:
// :335-344, as separable weighted taps built per image by the tables above), mirrors (ccv_flip, :347-348), normalises
// ((v - mean) * inv_std, :351-354: BEFORE the late crop, so the window's overhang stays 0 as ccv_slice leaves it, :357-361)
// and writes straight into the batch tensor the trainer feeds (NHWC or NCHW, CCV_32F or CCV_16F: dataframe combine + the
// trainer's datatype conversion, bin/nnc/imagenet.c:390-397) -- no per-image intermediate ever exists.
struct jitter_dev_t {
	long src;                 // byte offset of the slice's first pixel in the source buffer
	int step;                 // source row pitch, bytes
	int xs, xt, ys, yt;       // offsets (elements) of this image's tap-start / tap arrays in the packed tables
	int res_rows, res_cols;   // size after resampling
	int crop_x, crop_y, flip; // window origin in the resampled image, mirror in x
	int nops, mean_slot;      // colour operations; index of this image's contrast mean in the means array (-1: no contrast)
	int kind[4];
	float v[4][3];
};
// resampled value of channel c at (ry, rx) of the resampled image (before the mirror)
__device__ __forceinline__ float jitter_sample(const unsigned char* __restrict__ src, const jitter_dev_t& im, const int* __restrict__ starts, const tap_f32_t* __restrict__ taps, const int ry, const int rx, const int c)
{
	const int* xs = starts + im.xs;
	const int* ys = starts + im.ys;
	const tap_f32_t* xt = taps + im.xt;
	const tap_f32_t* yt = taps + im.yt;
	const unsigned char* base = src + im.src;
	float acc = 0.f;
	for (int ky = ys[ry]; ky < ys[ry + 1]; ky++) {
		const unsigned char* row = base + (long)yt[ky].si * im.step + c;
		float h = 0.f;
		for (int kx = xs[rx]; kx < xs[rx + 1]; kx++) h += (float)row[xt[kx].si] * xt[kx].w;
		acc += h * yt[ky].w;
	}
	return acc;
}
// operations [0, upto) of the image's colour list on one pixel; every operation stores floats, like the reference's in-place ccv_* calls
__device__ __forceinline__ void jitter_color(float p[3], const jitter_dev_t& im, const int upto, const double* __restrict__ mean)
{
	for (int o = 0; o < upto; o++) {
		const double ds = (double)im.v[o][0];
		switch (im.kind[o]) {
			case NNC_MI355X_COLOR_BRIGHTNESS: // ccv_scale (ccv_algebra.c:272-): p * ds
				for (int k = 0; k < 3; k++) p[k] = (float)((double)p[k] * ds);
				break;
			case NNC_MI355X_COLOR_SATURATION: { // ccv_saturation (ccv_image_processing.c:46-71)
				const double gs = (double)p[0] * 0.299 + (double)p[1] * 0.587 + (double)p[2] * 0.114;
				for (int k = 0; k < 3; k++) p[k] = (float)(((double)p[k] - gs) * ds + gs);
				break;
			}
			case NNC_MI355X_COLOR_CONTRAST: // ccv_contrast (:73-125): about the image's per-channel mean
				for (int k = 0; k < 3; k++) p[k] = (float)(((double)p[k] - mean[k]) * ds + mean[k]);
				break;
			case NNC_MI355X_COLOR_LIGHTING: // _ccv_cnnp_image_lighting (dataframe_addons.c:187-198): float add, clamp to [0, 255]
				for (int k = 0; k < 3; k++) { const float q = p[k] + im.v[o][k]; p[k] = q < 0.f ? 0.f : (q > 255.f ? 255.f : q); }
				break;
		}
	}
}
// Contrast needs the mean of the WHOLE resampled image as it stands when the operation is reached: the operations in front of it
// applied to every resampled pixel, summed per channel in double.  grid (chunks, images with a contrast operation); partial sums
// per chunk, folded by jitter_mean_fold_kernel in chunk order.
constexpr int JM_CHUNKS = 32;
__global__ void __launch_bounds__(256) jitter_mean_kernel(const unsigned char* __restrict__ src, const jitter_dev_t* __restrict__ imgs, const int* __restrict__ list, const int* __restrict__ starts, const tap_f32_t* __restrict__ taps, double* __restrict__ partial)
{
	__shared__ double red[4][3];
	const jitter_dev_t im = imgs[list[blockIdx.y]];
	int upto = 0;
	while (upto < im.nops && im.kind[upto] != NNC_MI355X_COLOR_CONTRAST) upto++;
	const long npix = (long)im.res_rows * im.res_cols;
	const long per = (npix + JM_CHUNKS - 1) / JM_CHUNKS, p0 = blockIdx.x * per, p1 = p0 + per < npix ? p0 + per : npix;
	double s[3] = { 0, 0, 0 };
	for (long i = p0 + threadIdx.x; i < p1; i += 256) {
		const int ry = (int)(i / im.res_cols), rx = (int)(i - (long)ry * im.res_cols);
		float p[3];
		for (int c = 0; c < 3; c++) p[c] = jitter_sample(src, im, starts, taps, ry, rx, c);
		jitter_color(p, im, upto, 0);
		for (int c = 0; c < 3; c++) s[c] += (double)p[c];
	}
	for (int c = 0; c < 3; c++) {
		double v = s[c];
		for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
		if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][c] = v;
	}
	__syncthreads();
	if (threadIdx.x < 3) partial[((long)blockIdx.y * JM_CHUNKS + blockIdx.x) * 3 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ void jitter_mean_fold_kernel(const double* __restrict__ partial, const jitter_dev_t* __restrict__ imgs, const int* __restrict__ list, double* __restrict__ means, const int n)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n * 3) return;
	const int slot = i / 3, c = i - slot * 3;
	double s = 0;
	for (int k = 0; k < JM_CHUNKS; k++) s += partial[((long)slot * JM_CHUNKS + k) * 3 + c];
	const jitter_dev_t& im = imgs[list[slot]];
	means[i] = s / ((double)im.res_rows * im.res_cols);
}
template <typename TO>
__global__ void __launch_bounds__(256) jitter_kernel(const unsigned char* src, const jitter_dev_t* imgs, const int* starts, const tap_f32_t* taps, const double* means, TO* out,
	const int out_rows, const int out_cols, const int ch, const int nchw, const float m0, const float m1, const float m2, const float s0, const float s1, const float s2, const size_t total)
{ // one thread per output PIXEL (all its channels: saturation mixes them); lanes run along the row
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	const float mean_c[3] = { m0, m1, m2 }, inv_c[3] = { s0, s1, s2 };
	for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
		size_t r = idx;
		const int ox = (int)(r % out_cols); r /= out_cols;
		const int oy = (int)(r % out_rows); r /= out_rows;
		const jitter_dev_t im = imgs[r];
		const int ry = oy + im.crop_y;
		int rx = ox + im.crop_x;
		float p[3] = { 0.f, 0.f, 0.f };
		if (ry >= 0 && ry < im.res_rows && rx >= 0 && rx < im.res_cols) {
			if (im.flip) rx = im.res_cols - 1 - rx;
			for (int c = 0; c < ch; c++) p[c] = jitter_sample(src, im, starts, taps, ry, rx, c);
			if (im.nops) jitter_color(p, im, im.nops, im.mean_slot >= 0 ? means + 3 * im.mean_slot : 0);
			for (int c = 0; c < ch; c++) p[c] = (p[c] - mean_c[c]) * inv_c[c];
		}
		const size_t plane = (size_t)out_rows * out_cols, pix = (size_t)oy * out_cols + ox;
		for (int c = 0; c < ch; c++) out[nchw ? (r * ch + c) * plane + pix : (r * plane + pix) * ch + c] = (TO)p[c];
	}
}
template <typename TO>
__global__ void __launch_bounds__(256) one_hot_kernel(const int* labels, TO* out, const int range, const float onval, const float offval, const size_t total)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) out[idx] = (TO)((int)(idx % range) == labels[idx / range] ? onval : offval);
}

} // namespace

// One axis of _ccv_filter_kissfft's tiling (lib/ccv_numeric.c:775-776 tile size, :836-839 tile count, :846-925 the copy-out blocks): n = image extent, k = kernel
// extent.  Returns the tile extent (0: the reference itself would divide by zero -- a one-pixel axis under an even kernel; the caller refuses) and fills map[0 .. n - 1].
static int kiss_next_fast(int n)
{ // kiss_fft_next_fast_size (lib/3rdparty/kissfft/kiss_fft.c:396-408)
	for (;;) {
		int m = n;
		while (m % 2 == 0) m /= 2;
		while (m % 3 == 0) m /= 3;
		while (m % 5 == 0) m /= 5;
		if (m <= 1) return n;
		n++;
	}
}
static int filter_axis_map(const int n, const int k, std::vector<filter_map_t>& map)
{
	const int fast = kiss_next_fast((k * 3 + 1) >> 1) << 1; // kiss_fftr_next_fast_size_real(k * 3)
	const int t = ((imin(n + k - 1, fast) + 1) >> 1) << 1;
	const int ke = k & ~1, k2 = k / 2;
	map.assign(n, filter_map_t{ 0, 0, -1 });
	if (t - ke <= 0) return 0;
	const int tiles = imax(1, (n + t - 2 * ke) / (t - ke));
	for (int i = 0; i < tiles; i++) {
		const int o = imin(i * (t - ke), imax(n - t, 0));
		const int end_in = imin(t, n - o);
		const int out0 = o + (i > 0 ? k2 : 0);
		const int end = imin(n - out0, (t - ke) + (i == 0 ? k2 : 0));
		for (int y = 0; y < end; y++) map[out0 + y] = filter_map_t{ o, end_in, (1 + (i > 0 ? 1 : 0)) * k2 + y };
		if (i + 1 == tiles && end + out0 < n) { // the last tile's edge block: read from the top of the circular result
			const int end_tile = imin(k2, n - (out0 + end));
			for (int y = 0; y < end_tile; y++) map[out0 + end + y] = filter_map_t{ o, end_in, y };
		}
	}
	return t;
}

// ---- the 8-bit area path's tap tables, per geometry, resident on the device (a few KB each; the least recently used of 16 gives way)
namespace {
struct area_tables_t { int a_rows, a_cols, b_rows, b_cols, ch, device, live, pins; double sx, sy; char* dev; size_t oxs, oxt, oys, oyt, oe16, oe4; int e_pitch, maxt; unsigned inv; unsigned long stamp; };
area_tables_t g_area_tables[16];
unsigned long g_area_stamp = 0;
pthread_mutex_t g_area_mutex = PTHREAD_MUTEX_INITIALIZER;
// The caller holds the returned table PINNED until its kernel has been enqueued (area_8u_unpin): a concurrent thread that needs a slot takes the least recently
// used UNPINNED one, and frees its table outside the mutex (the table could be evicted and freed between this function's return and the launch;
// the free -- stream-ordered since round 6 -- covers only what has been enqueued).  All 16 slots pinned (17 threads resampling 17 geometries at once): the new
// table is not cached, *one_shot tells the caller to free it behind its launch.
void area_8u_unpin(char* const dev, const bool one_shot, const int device)
{
	if (one_shot) { nnc_mi355x_free(device, dev); return; }
	pthread_mutex_lock(&g_area_mutex);
	for (area_tables_t& e : g_area_tables) if (e.live && e.dev == dev) { if (e.pins > 0) e.pins--; break; }
	pthread_mutex_unlock(&g_area_mutex);
}
char* area_8u_tables(bool* const one_shot, const int a_rows, const int a_cols, const int b_rows, const int b_cols, const int ch, const double sx, const double sy, size_t* oxs, size_t* oxt, size_t* oys, size_t* oyt, unsigned* inv, size_t* oe16, size_t* oe4, int* e_pitch, int* maxt)
{
	int device = 0;
	HIP_ENFORCE(hipGetDevice(&device));
	pthread_mutex_lock(&g_area_mutex);
	*one_shot = false;
	area_tables_t* hit = 0;
	area_tables_t* lru = 0;
	for (area_tables_t& e : g_area_tables) {
		if (e.live && e.a_rows == a_rows && e.a_cols == a_cols && e.b_rows == b_rows && e.b_cols == b_cols && e.ch == ch && e.device == device && e.sx == sx && e.sy == sy) { hit = &e; break; }
		if (e.live && e.pins > 0) continue;
		if (!lru || (lru->live && (!e.live || e.stamp < lru->stamp))) lru = &e;
	}
	char* evicted = 0;
	int evicted_device = 0;
	if (!hit) {
		std::vector<int> xs, ys;
		std::vector<tap_u32_t> xt, yt;
		area_x_taps<tap_u32_t, unsigned>(a_cols, b_cols, ch, sx, 256u, 256.0, xs, xt); // NB the reference's left tap uses 0x100 and the right one 256: the same number
		area_y_taps_8u(a_rows, b_rows, sy, ys, yt);
		*inv = (unsigned)(int)(sx * sy * 0x10000);
		if (*inv == 0) { pthread_mutex_unlock(&g_area_mutex); return 0; }
		upload_t u;
		area_tables_t e;
		e.oxs = u.add(xs.data(), xs.size() * sizeof(int)); e.oxt = u.add(xt.data(), xt.size() * sizeof(tap_u32_t)); e.oys = u.add(ys.data(), ys.size() * sizeof(int)); e.oyt = u.add(yt.data(), yt.size() * sizeof(tap_u32_t));
		// the rows kernel's per-element tables (see area_8u_rows_kernel): [tap index][output element], the LDS column pre-swizzled for the 16-byte and the 4-byte variant
		e.maxt = 1;
		for (int dx = 0; dx < b_cols; dx++) e.maxt = imax(e.maxt, xs[dx + 1] - xs[dx]);
		e.e_pitch = (b_cols * ch + 3) & ~3;
		std::vector<etap_t> e16((size_t)e.maxt * e.e_pitch, etap_t{ 0, 0 }), e4((size_t)e.maxt * e.e_pitch, etap_t{ 0, 0 });
		for (int dx = 0; dx < b_cols; dx++)
			for (int c = 0; c < ch; c++)
				for (int kx = xs[dx]; kx < xs[dx + 1]; kx++) {
					const size_t at = (size_t)(kx - xs[dx]) * e.e_pitch + dx * ch + c;
					e16[at] = etap_t{ (unsigned)area_col<16>(xt[kx].si + c), xt[kx].w };
					e4[at] = etap_t{ (unsigned)area_col<4>(xt[kx].si + c), xt[kx].w };
				}
		e.oe16 = u.add(e16.data(), e16.size() * sizeof(etap_t)); e.oe4 = u.add(e4.data(), e4.size() * sizeof(etap_t));
		e.dev = (char*)nnc_mi355x_malloc(device, u.host.size());
		if (!e.dev) { *inv = 1; pthread_mutex_unlock(&g_area_mutex); return 0; }
		HIP_ENFORCE(hipMemcpy(e.dev, u.host.data(), u.host.size(), hipMemcpyHostToDevice)); // (once per geometry)
		e.a_rows = a_rows; e.a_cols = a_cols; e.b_rows = b_rows; e.b_cols = b_cols; e.ch = ch; e.device = device; e.live = 1; e.pins = 0; e.sx = sx; e.sy = sy; e.inv = *inv;
		if (!lru) { // every slot is pinned by a launch in progress: hand the table out uncached
			*one_shot = true;
			*oxs = e.oxs; *oxt = e.oxt; *oys = e.oys; *oyt = e.oyt; *oe16 = e.oe16; *oe4 = e.oe4; *e_pitch = e.e_pitch; *maxt = e.maxt;
			pthread_mutex_unlock(&g_area_mutex);
			return e.dev;
		}
		if (lru->live) { evicted = lru->dev; evicted_device = lru->device; } // queued kernels may still read it: freed below, stream-ordered behind them
		*lru = e;
		hit = lru;
	}
	hit->stamp = ++g_area_stamp;
	hit->pins++;
	*oxs = hit->oxs; *oxt = hit->oxt; *oys = hit->oys; *oyt = hit->oyt; *inv = hit->inv; *oe16 = hit->oe16; *oe4 = hit->oe4; *e_pitch = hit->e_pitch; *maxt = hit->maxt;
	char* const dev = hit->dev;
	pthread_mutex_unlock(&g_area_mutex);
	if (evicted) { // (outside the mutex: the free may flush recorded commands and take other locks)
		nnc_mi355x_free(evicted_device, evicted);
		HIP_ENFORCE(hipSetDevice(device));
	}
	return dev;
}
}

extern "C" {

int nnc_mi355x_resample_batch(const void* a, const nnc_mi355x_image_batch_t ad, void* b, const nnc_mi355x_image_batch_t bd, const int count, const double rows_scale, const double cols_scale, const int type, ccv_nnc_stream_context_t* const stream_context)
{
	if (!a || !b || count < 0 || rows_scale <= 0 || cols_scale <= 0) return CCV_NNC_EXEC_INVALID;
	if (ad.channels != bd.channels || ad.channels < 1 || ad.rows < 1 || ad.cols < 1 || bd.rows < 1 || bd.cols < 1) return CCV_NNC_EXEC_INVALID;
	const int adt = CCV_GET_DATA_TYPE(ad.datatype), bdt = CCV_GET_DATA_TYPE(bd.datatype);
	if ((adt != CCV_8U && adt != CCV_32F) || (bdt != CCV_8U && bdt != CCV_32F)) return CCV_NNC_EXEC_INVALID;
	if (count == 0) return CCV_NNC_EXEC_SUCCESS;
	const int ch = ad.channels;
	hipStream_t stream = stream_of(stream_context);
	const size_t total = (size_t)count * bd.rows * bd.cols * ch;
	const int grid = grid_for(total, 256);
	const double scale_x = 1.0 / cols_scale, scale_y = 1.0 / rows_scale;
	if (ad.rows == bd.rows && ad.cols == bd.cols && adt == bdt) { // same size: plain copy (ccv_resample.c:450-458)
		for (int i = 0; i < count; i++)
			HIP_ENFORCE(hipMemcpy2DAsync((char*)b + i * bd.image_stride, bd.step, (const char*)a + i * ad.image_stride, ad.step, (size_t)bd.cols * ch * datatype_size(bdt), (size_t)bd.rows, hipMemcpyDeviceToDevice, stream));
		return CCV_NNC_EXEC_SUCCESS;
	}
	upload_t u;
	if ((type & 0x01 /* CCV_INTER_AREA */) && ad.rows >= bd.rows && ad.cols >= bd.cols) {
		if (adt == CCV_8U && bdt == CCV_8U && (long)ad.rows * ad.cols / ((long)bd.rows * bd.cols) < 0x100) {
			if (ch > 4) return CCV_NNC_EXEC_INVALID; // the reference clamps the channel count to 4 on this path (:14)
			// the tap tables depend on the geometry alone: built once per (sizes, channels, scales, device) and kept on the device -- a loader resamples
			// thousands of batches to the same size, and the host-side building + upload used to be part of every call
			size_t oxs, oxt, oys, oyt, oe16, oe4;
			int e_pitch, maxt;
			unsigned inv_scale_256;
			bool one_shot = false;
			char* const dev = area_8u_tables(&one_shot, ad.rows, ad.cols, bd.rows, bd.cols, ch, scale_x, scale_y, &oxs, &oxt, &oys, &oyt, &inv_scale_256, &oe16, &oe4, &e_pitch, &maxt);
			if (!dev) return inv_scale_256 == 0 ? CCV_NNC_EXEC_INVALID : CCV_NNC_EXEC_OOM;
			int tables_device = 0;
			HIP_ENFORCE(hipGetDevice(&tables_device));
			// one workgroup per output row (area_8u_rows_kernel) when rows are dword-aligned on both sides and the row's column sums fit the LDS
			const long a_cols_ch = (long)ad.cols * ch;
			const bool al4 = (((uintptr_t)a | (uintptr_t)b | (uintptr_t)ad.step | (uintptr_t)ad.image_stride | (uintptr_t)bd.step | (uintptr_t)bd.image_stride) & 3) == 0;
			const bool al16 = (((uintptr_t)a | (uintptr_t)ad.step | (uintptr_t)ad.image_stride) & 15) == 0;
			constexpr int AREA_R = 4; // output rows per workgroup
			const size_t lds = (size_t)((a_cols_ch + 15) / 16 * 16) * sizeof(unsigned) * AREA_R;
			static const int rows_kernel = getenv("NNC_MI355X_RESAMPLE_ROWS") ? atoi(getenv("NNC_MI355X_RESAMPLE_ROWS")) : 1;
			const int row_groups = (bd.rows + AREA_R - 1) / AREA_R;
			if (rows_kernel && al4 && lds <= 64 * 1024 && (long)count * row_groups < 0x7fffffffL && ad.step >= a_cols_ch) {
				const double rcp = 1.0 / (double)inv_scale_256, half = 0.5 * rcp;
				const dim3 g((unsigned)((long)count * row_groups));
#define AREA_ROWS(VEC, OE) hipLaunchKernelGGL(HIP_KERNEL_NAME(area_8u_rows_kernel<VEC, AREA_R>), g, dim3(128), lds, stream, (const unsigned char*)a, (unsigned char*)b, ad.step, ad.image_stride, bd.step, bd.image_stride, bd.rows, bd.cols * ch, (int)a_cols_ch, row_groups, \
					(const etap_t*)(dev + OE), e_pitch, maxt, (const int*)(dev + oys), (const tap_u32_t*)(dev + oyt), rcp, half)
				if (al16) AREA_ROWS(16, oe16); else AREA_ROWS(4, oe4);
#undef AREA_ROWS
			} else
			hipLaunchKernelGGL(area_8u_kernel, dim3(grid), dim3(256), 0, stream, (const unsigned char*)a, (unsigned char*)b, ad.step, ad.image_stride, bd.step, bd.image_stride, bd.rows, bd.cols * ch, ch,
				(const int*)(dev + oxs), (const tap_u32_t*)(dev + oxt), (const int*)(dev + oys), (const tap_u32_t*)(dev + oyt), inv_scale_256, total);
			area_8u_unpin(dev, one_shot, tables_device); // the kernel is in the stream: the table may be evicted (its free is ordered behind the kernel)
		} else {
			std::vector<int> xs, ys;
			std::vector<tap_f32_t> xt, yt;
			const double scale = 1.f / (scale_x * scale_y);
			area_x_taps<tap_f32_t, float>(ad.cols, bd.cols, ch, scale_x, (float)scale, scale, xs, xt);
			area_y_taps_f32(ad.rows, bd.rows, scale_y, ys, yt);
			const size_t oxs = u.add(xs.data(), xs.size() * sizeof(int)), oxt = u.add(xt.data(), xt.size() * sizeof(tap_f32_t)), oys = u.add(ys.data(), ys.size() * sizeof(int)), oyt = u.add(yt.data(), yt.size() * sizeof(tap_f32_t));
			char* dev = upload(u, stream_context);
			if (!dev) return CCV_NNC_EXEC_OOM;
#define AREA_F32(TA, TB) hipLaunchKernelGGL(HIP_KERNEL_NAME(area_f32_kernel<TA, TB>), dim3(grid), dim3(256), 0, stream, (const unsigned char*)a, (unsigned char*)b, ad.step, ad.image_stride, bd.step, bd.image_stride, bd.rows, bd.cols * ch, ch, \
				(const int*)(dev + oxs), (const tap_f32_t*)(dev + oxt), (const int*)(dev + oys), (const tap_f32_t*)(dev + oyt), total)
			if (adt == CCV_8U && bdt == CCV_8U) AREA_F32(unsigned char, unsigned char);
			else if (adt == CCV_8U) AREA_F32(unsigned char, float);
			else if (bdt == CCV_8U) AREA_F32(float, unsigned char);
			else AREA_F32(float, float);
#undef AREA_F32
		}
	} else if (type & 0x04 /* CCV_INTER_CUBIC */) {
		if (bdt == CCV_32F) {
			std::vector<cubic_f_t> xo(bd.cols), yo(bd.rows);
			for (int i = 0; i < bd.cols; i++) { const double sx = (i + 0.5) * scale_x - 0.5; cubic_f((int)sx, ad.cols, (float)sx, &xo[i]); }
			for (int i = 0; i < bd.rows; i++) { const double sy = (i + 0.5) * scale_y - 0.5; cubic_f((int)sy, ad.rows, (float)sy, &yo[i]); }
			const size_t ox = u.add(xo.data(), xo.size() * sizeof(cubic_f_t)), oy = u.add(yo.data(), yo.size() * sizeof(cubic_f_t));
			char* dev = upload(u, stream_context);
			if (!dev) return CCV_NNC_EXEC_OOM;
			if (adt == CCV_8U) hipLaunchKernelGGL(HIP_KERNEL_NAME(cubic_f32_kernel<unsigned char>), dim3(grid), dim3(256), 0, stream, (const unsigned char*)a, (unsigned char*)b, ad.step, ad.image_stride, bd.step, bd.image_stride, bd.rows, bd.cols * ch, ch, (const cubic_f_t*)(dev + ox), (const cubic_f_t*)(dev + oy), total);
			else hipLaunchKernelGGL(HIP_KERNEL_NAME(cubic_f32_kernel<float>), dim3(grid), dim3(256), 0, stream, (const unsigned char*)a, (unsigned char*)b, ad.step, ad.image_stride, bd.step, bd.image_stride, bd.rows, bd.cols * ch, ch, (const cubic_f_t*)(dev + ox), (const cubic_f_t*)(dev + oy), total);
		} else {
			if (adt != CCV_8U) return CCV_NNC_EXEC_INVALID;
			std::vector<cubic_i_t> xo(bd.cols), yo(bd.rows);
			for (int i = 0; i < bd.cols; i++) { const double sx = (i + 0.5) * scale_x - 0.5; cubic_i((int)sx, ad.cols, (float)sx, &xo[i]); }
			for (int i = 0; i < bd.rows; i++) { const double sy = (i + 0.5) * scale_y - 0.5; cubic_i((int)sy, ad.rows, (float)sy, &yo[i]); }
			const size_t ox = u.add(xo.data(), xo.size() * sizeof(cubic_i_t)), oy = u.add(yo.data(), yo.size() * sizeof(cubic_i_t));
			char* dev = upload(u, stream_context);
			if (!dev) return CCV_NNC_EXEC_OOM;
			hipLaunchKernelGGL(cubic_8u_kernel, dim3(grid), dim3(256), 0, stream, (const unsigned char*)a, (unsigned char*)b, ad.step, ad.image_stride, bd.step, bd.image_stride, bd.rows, bd.cols * ch, ch, (const cubic_i_t*)(dev + ox), (const cubic_i_t*)(dev + oy), total);
		}
	} else
		return CCV_NNC_EXEC_INVALID; // the reference asserts: LINEAR / LANCZOS are not implemented there either (:470-476)
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}

int nnc_mi355x_jitter_batch(const void* src, const nnc_mi355x_jitter_image_t* images, const int count, const nnc_mi355x_jitter_params_t params, void* out, ccv_nnc_stream_context_t* const stream_context)
{
	if (!src || !images || !out || count < 0 || params.out_rows < 1 || params.out_cols < 1 || params.channels < 1 || params.channels > 3) return CCV_NNC_EXEC_INVALID;
	const int odt = CCV_GET_DATA_TYPE(params.datatype);
	if ((odt != CCV_32F && odt != CCV_16F) || (params.format != CCV_TENSOR_FORMAT_NHWC && params.format != CCV_TENSOR_FORMAT_NCHW)) return CCV_NNC_EXEC_INVALID;
	if (count == 0) return CCV_NNC_EXEC_SUCCESS;
	const int ch = params.channels;
	// (round 5: the per-image tables are built into buffers that live for the thread -- four vectors used to be constructed, grown and destroyed per image,
	// a thousand allocations per batch of 256, most of the host's 4 ms per batch)
	static thread_local std::vector<jitter_dev_t> descs;
	static thread_local std::vector<int> mean_list; // images with a contrast operation
	static thread_local std::vector<int> starts;
	static thread_local std::vector<tap_f32_t> taps;
	static thread_local std::vector<int> xs, ys;
	static thread_local std::vector<tap_f32_t> xt, yt;
	descs.assign(count, jitter_dev_t());
	mean_list.clear(); starts.clear(); taps.clear();
	for (int i = 0; i < count; i++) {
		const nnc_mi355x_jitter_image_t& im = images[i];
		if (im.slice_rows < 1 || im.slice_cols < 1 || im.resize_rows < 1 || im.resize_cols < 1 || im.slice_x < 0 || im.slice_y < 0 || im.slice_x + im.slice_cols > im.cols || im.slice_y + im.slice_rows > im.rows) return CCV_NNC_EXEC_INVALID;
		jitter_dev_t& d = descs[i];
		d.src = (long)im.offset + (long)im.slice_y * im.step + (long)im.slice_x * ch;
		d.step = im.step; d.res_rows = im.resize_rows; d.res_cols = im.resize_cols; d.crop_x = im.crop_x; d.crop_y = im.crop_y; d.flip = im.flip ? 1 : 0;
		if (im.color_ops < 0 || im.color_ops > 4 || (im.color_ops && ch != 3)) return CCV_NNC_EXEC_INVALID;
		d.nops = im.color_ops; d.mean_slot = -1;
		int contrasts = 0;
		for (int o = 0; o < 4; o++) {
			d.kind[o] = o < im.color_ops ? im.color[o].kind : 0;
			for (int k = 0; k < 3; k++) d.v[o][k] = o < im.color_ops ? im.color[o].v[k] : 0.f;
			if (o < im.color_ops && (d.kind[o] < NNC_MI355X_COLOR_BRIGHTNESS || d.kind[o] > NNC_MI355X_COLOR_LIGHTING)) return CCV_NNC_EXEC_INVALID;
			if (d.kind[o] == NNC_MI355X_COLOR_CONTRAST) contrasts++;
		}
		if (contrasts > 1) return CCV_NNC_EXEC_INVALID; // (the reference applies each operation at most once)
		if (contrasts) { d.mean_slot = (int)mean_list.size(); mean_list.push_back(i); }
		xs.clear(); ys.clear(); xt.clear(); yt.clear();
		const double scale_x = (double)im.slice_cols / im.resize_cols, scale_y = (double)im.slice_rows / im.resize_rows;
		if (im.slice_rows >= im.resize_rows && im.slice_cols >= im.resize_cols && (im.slice_rows != im.resize_rows || im.slice_cols != im.resize_cols)) { // CCV_INTER_AREA (:335-338)
			const double scale = 1.f / (scale_x * scale_y);
			area_x_taps<tap_f32_t, float>(im.slice_cols, im.resize_cols, ch, scale_x, (float)scale, scale, xs, xt);
			area_y_taps_f32(im.slice_rows, im.resize_rows, scale_y, ys, yt);
		} else if (im.slice_rows != im.resize_rows || im.slice_cols != im.resize_cols) { // CCV_INTER_CUBIC (:339-340): four taps per output
			xs.resize(im.resize_cols + 1); ys.resize(im.resize_rows + 1);
			for (int j = 0; j < im.resize_cols; j++) {
				const double sx = (j + 0.5) * scale_x - 0.5;
				cubic_f_t c; cubic_f((int)sx, im.slice_cols, (float)sx, &c);
				xs[j] = (int)xt.size();
				for (int k = 0; k < 4; k++) { tap_f32_t t; t.si = c.si[k] * ch; t.w = c.w[k]; xt.push_back(t); }
			}
			xs[im.resize_cols] = (int)xt.size();
			for (int j = 0; j < im.resize_rows; j++) {
				const double sy = (j + 0.5) * scale_y - 0.5;
				cubic_f_t c; cubic_f((int)sy, im.slice_rows, (float)sy, &c);
				ys[j] = (int)yt.size();
				for (int k = 0; k < 4; k++) { tap_f32_t t; t.si = c.si[k]; t.w = c.w[k]; yt.push_back(t); }
			}
			ys[im.resize_rows] = (int)yt.size();
		} else { // same size: ccv_shift to 32F (:342): one tap of weight 1
			xs.resize(im.resize_cols + 1); ys.resize(im.resize_rows + 1);
			for (int j = 0; j <= im.resize_cols; j++) xs[j] = j;
			for (int j = 0; j <= im.resize_rows; j++) ys[j] = j;
			for (int j = 0; j < im.resize_cols; j++) { tap_f32_t t; t.si = j * ch; t.w = 1.f; xt.push_back(t); }
			for (int j = 0; j < im.resize_rows; j++) { tap_f32_t t; t.si = j; t.w = 1.f; yt.push_back(t); }
		}
		// pack: starts are made absolute into the shared tap array
		d.xs = (int)starts.size();
		for (size_t j = 0; j < xs.size(); j++) starts.push_back(xs[j] + (int)taps.size());
		d.xt = 0;
		taps.insert(taps.end(), xt.begin(), xt.end());
		d.ys = (int)starts.size();
		for (size_t j = 0; j < ys.size(); j++) starts.push_back(ys[j] + (int)taps.size());
		d.yt = 0;
		taps.insert(taps.end(), yt.begin(), yt.end());
	}
	upload_t u;
	const size_t od = u.add(descs.data(), descs.size() * sizeof(jitter_dev_t)), os = u.add(starts.data(), starts.size() * sizeof(int)), ot = u.add(taps.data(), taps.size() * sizeof(tap_f32_t));
	const size_t nm = mean_list.size();
	const int zero = 0;
	const size_t ol = u.add(nm ? (const void*)mean_list.data() : (const void*)&zero, sizeof(int) * (nm ? nm : 1));
	// behind the tables: the contrast means and their per-chunk partial sums (device scratch, not uploaded)
	const size_t tables = (u.host.size() + 15) & ~(size_t)15;
	const size_t omean = tables, opart = omean + sizeof(double) * 3 * (nm ? nm : 1);
	u.host.resize(opart + sizeof(double) * 3 * JM_CHUNKS * (nm ? nm : 1));
	char* dev = upload(u, stream_context);
	if (!dev) return CCV_NNC_EXEC_OOM;
	const size_t total = (size_t)count * params.out_rows * params.out_cols; // pixels
	hipStream_t stream = stream_of(stream_context);
	if (nm) {
		hipLaunchKernelGGL(jitter_mean_kernel, dim3(JM_CHUNKS, (unsigned)nm), dim3(256), 0, stream, (const unsigned char*)src, (const jitter_dev_t*)(dev + od), (const int*)(dev + ol), (const int*)(dev + os), (const tap_f32_t*)(dev + ot), (double*)(dev + opart));
		hipLaunchKernelGGL(jitter_mean_fold_kernel, dim3((unsigned)((nm * 3 + 63) / 64)), dim3(64), 0, stream, (const double*)(dev + opart), (const jitter_dev_t*)(dev + od), (const int*)(dev + ol), (double*)(dev + omean), (int)nm);
		HIP_ENFORCE(hipGetLastError());
	}
	const int nchw = params.format == CCV_TENSOR_FORMAT_NCHW;
#define JITTER(TO) hipLaunchKernelGGL(HIP_KERNEL_NAME(jitter_kernel<TO>), dim3(grid_for(total, 256)), dim3(256), 0, stream, (const unsigned char*)src, (const jitter_dev_t*)(dev + od), (const int*)(dev + os), (const tap_f32_t*)(dev + ot), (const double*)(dev + omean), (TO*)out, \
		params.out_rows, params.out_cols, ch, nchw, params.mean[0], params.mean[1], params.mean[2], params.inv_std[0], params.inv_std[1], params.inv_std[2], total)
	if (odt == CCV_32F) JITTER(float); else JITTER(_Float16);
#undef JITTER
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}

int nnc_mi355x_one_hot_batch(const int* labels_host, const int count, const int range, const float onval, const float offval, const int datatype, void* out, ccv_nnc_stream_context_t* const stream_context)
{ // _ccv_cnnp_one_hot (lib/nnc/ccv_cnnp_dataframe_addons.c:378-): one row of `range` values per label, onval at the label, offval elsewhere
	if (!labels_host || !out || count < 0 || range < 1) return CCV_NNC_EXEC_INVALID;
	const int odt = CCV_GET_DATA_TYPE(datatype);
	if (odt != CCV_32F && odt != CCV_16F) return CCV_NNC_EXEC_INVALID;
	if (count == 0) return CCV_NNC_EXEC_SUCCESS;
	for (int i = 0; i < count; i++) if (labels_host[i] < 0 || labels_host[i] >= range) return CCV_NNC_EXEC_INVALID;
	upload_t u;
	const size_t ol = u.add(labels_host, sizeof(int) * (size_t)count);
	char* dev = upload(u, stream_context);
	if (!dev) return CCV_NNC_EXEC_OOM;
	const size_t total = (size_t)count * range;
	hipStream_t stream = stream_of(stream_context);
	if (odt == CCV_32F) hipLaunchKernelGGL(HIP_KERNEL_NAME(one_hot_kernel<float>), dim3(grid_for(total, 256)), dim3(256), 0, stream, (const int*)(dev + ol), (float*)out, range, onval, offval, total);
	else hipLaunchKernelGGL(HIP_KERNEL_NAME(one_hot_kernel<_Float16>), dim3(grid_for(total, 256)), dim3(256), 0, stream, (const int*)(dev + ol), (_Float16*)out, range, onval, offval, total);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}

int nnc_mi355x_filter_batch(const void* a, const nnc_mi355x_image_batch_t ad, const void* kernel_host, const int kernel_rows, const int kernel_cols, const int kernel_channels, void* d, const nnc_mi355x_image_batch_t dd, const int count, ccv_nnc_stream_context_t* const stream_context)
{
	if (!a || !d || !kernel_host || count < 0 || kernel_rows < 1 || kernel_cols < 1) return CCV_NNC_EXEC_INVALID;
	if (ad.rows != dd.rows || ad.cols != dd.cols || ad.channels != dd.channels) return CCV_NNC_EXEC_INVALID;
	const int adt = CCV_GET_DATA_TYPE(ad.datatype), ddt = CCV_GET_DATA_TYPE(dd.datatype);
	if (count == 0) return CCV_NNC_EXEC_SUCCESS;
	hipStream_t stream = stream_of(stream_context);
	const float* kf = (const float*)kernel_host; // the kernel is a small host-side 32F matrix, as in the reference's callers
	upload_t u;
	if (adt == CCV_8U && ddt == CCV_8U) { // _ccv_filter_direct_8u (ccv_numeric.c:960-1034): single channel, 2^14 fixed point
		if (ad.channels != 1 || kernel_channels != 1) return CCV_NNC_EXEC_INVALID;
		std::vector<int> coeff(kernel_rows * kernel_cols);
		for (int i = 0; i < kernel_rows * kernel_cols; i++) coeff[i] = (int)(kf[i] * (1 << 14) + 0.5);
		const size_t oc = u.add(coeff.data(), coeff.size() * sizeof(int));
		char* dev = upload(u, stream_context);
		if (!dev) return CCV_NNC_EXEC_OOM;
		const size_t total = (size_t)count * ad.rows * ad.cols;
		hipLaunchKernelGGL(filter_8u_kernel, dim3(grid_for(total, 256)), dim3(256), 0, stream, (const unsigned char*)a, (unsigned char*)d, ad.step, ad.image_stride, dd.step, dd.image_stride, ad.rows, ad.cols, (const int*)(dev + oc), kernel_rows, kernel_cols, total);
	} else if (adt == CCV_32F && ddt == CCV_32F) {
		if (kernel_channels != 1 && kernel_channels != ad.channels) return CCV_NNC_EXEC_INVALID;
		const size_t oc = u.add(kf, sizeof(float) * kernel_rows * kernel_cols * kernel_channels);
		std::vector<filter_map_t> rowmap, colmap;
		const int trows = filter_axis_map(ad.rows, kernel_rows, rowmap), tcols = filter_axis_map(ad.cols, kernel_cols, colmap);
		if (trows <= 0 || tcols <= 0) return CCV_NNC_EXEC_INVALID; // (the reference divides by zero there)
		const size_t orm = u.add(rowmap.data(), rowmap.size() * sizeof(filter_map_t)), ocm = u.add(colmap.data(), colmap.size() * sizeof(filter_map_t));
		char* dev = upload(u, stream_context);
		if (!dev) return CCV_NNC_EXEC_OOM;
		const size_t total = (size_t)count * ad.rows * ad.cols * ad.channels;
		hipLaunchKernelGGL(filter_f32_kernel, dim3(grid_for(total, 256)), dim3(256), 0, stream, (const float*)a, (float*)d, ad.step, ad.image_stride, dd.step, dd.image_stride, ad.rows, ad.cols, ad.channels, (const float*)(dev + oc), kernel_rows, kernel_cols, kernel_channels,
			(const filter_map_t*)(dev + orm), (const filter_map_t*)(dev + ocm), trows, tcols, total);
	} else
		return CCV_NNC_EXEC_INVALID;
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}

} // extern "C"
