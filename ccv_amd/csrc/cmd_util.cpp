// Layout / type commands on gfx950: FORMAT_TRANSFORM (NCHW <-> NHWC, strided views), TRANSPOSE, DATATYPE_CONVERSION.
// All are pure data movement, bound by HBM (algorithmic bytes = |in| + |out|).
// Oracle semantics: lib/nnc/cmd/util/ccv_nnc_util_cpu_ref.c:996-1082 (format transform), :1102-1180 (transpose),
// :1200-1260 (datatype conversion).  Replaces util/gpu/ccv_nnc_util_gpu_cudnn.cu:13-60,160-222 and
// util/gpu/ccv_nnc_util_gpu_ref.cu (cudnnTransformTensor / element kernels).
//
// Two kernels do the work:
//   * permute4_kernel: generic 4-d gather, one element per lane in OUTPUT memory order (coalesced stores, strided loads);
//     serves views, small tensors and every odd case.
//   * transpose_tile_kernel: [batch][R][C] -> [batch][C][R] through a 64 x 65 LDS tile so that BOTH the global loads and
//     the global stores are 256-byte-per-wavefront coalesced; picked for dense NCHW <-> NHWC (R = channels, C = H*W or
//     the reverse), the layout change conv uses for NCHW tensors.
#include "common.h"
#include "isa.h"
#include <stdint.h>

using namespace nnc;

namespace {

typedef _Float16 half_t;

struct perm4_t {
	int dim[4];      // extents in output memory order (outer .. inner)
	long is[4], os[4]; // element strides of input / output for those axes
};

template <typename T>
__global__ void __launch_bounds__(256) permute4_kernel(const T* in, T* out, const perm4_t p, const size_t n)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += stride) {
		size_t r = idx;
		const int i3 = (int)(r % p.dim[3]); r /= p.dim[3];
		const int i2 = (int)(r % p.dim[2]); r /= p.dim[2];
		const int i1 = (int)(r % p.dim[1]); r /= p.dim[1];
		const int i0 = (int)r;
		out[i0 * p.os[0] + i1 * p.os[1] + i2 * p.os[2] + i3 * p.os[3]] = in[i0 * p.is[0] + i1 * p.is[1] + i2 * p.is[2] + i3 * p.is[3]];
	}
}

constexpr int TT = 64;
// in[b][r][c] (c contiguous) -> out[b][c][r] (r contiguous).  grid (ceil(C/64), ceil(R/64), batch), 256 threads.
template <typename T>
__global__ void __launch_bounds__(256) transpose_tile_kernel(const T* in, T* out, const int R, const int C)
{
	__shared__ T tile[TT][TT + 1];
	const int c0 = blockIdx.x * TT, r0 = blockIdx.y * TT;
	const size_t base = (size_t)blockIdx.z * R * C;
	const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
	for (int j = ty; j < TT; j += 4) {
		const int r = r0 + j, c = c0 + tx;
		if (r < R && c < C) tile[j][tx] = in[base + (size_t)r * C + c];
	}
	__syncthreads();
	for (int j = ty; j < TT; j += 4) {
		const int c = c0 + j, r = r0 + tx;
		if (r < R && c < C) out[base + (size_t)c * R + r] = tile[tx][j];
	}
}

// The same for 4-byte elements with 16-BYTE global accesses both ways (R % 4 == 0, C % 4 == 0, 16-byte aligned bases): a thread loads four consecutive c
// of a row as one dwordx4, writes them to the tile as four dwords, and after the barrier gathers four consecutive r of a column into one dwordx4 store.
// The one-dword-per-lane kernel above ran the ResNet-50 / DawnNet re-layouts at ~4 TB/s (profiles/r02_v10_transpose_bench.txt; 15 % of config 4's step);
// pitch 65 keeps both LDS phases at 2-way conflicts at most (bank = (4 r4 + e + lane row) mod 32).
template <typename T>
__global__ void __launch_bounds__(256) transpose_tile_vec4_kernel(const T* __restrict__ in, T* __restrict__ out, const int R, const int C)
{
	typedef T T4 __attribute__((ext_vector_type(4)));
	__shared__ T tile[TT][TT + 1];
	const int c0 = blockIdx.x * TT, r0 = blockIdx.y * TT;
	const size_t base = (size_t)blockIdx.z * R * C;
	const int q = threadIdx.x & 15, l = threadIdx.x >> 4;
#pragma unroll
	for (int i = 0; i < 4; i++) {
		const int r = r0 + l + 16 * i, c = c0 + 4 * q;
		if (r < R && c < C) {
			const T4 v = *(const T4*)(in + base + (size_t)r * C + c);
			tile[l + 16 * i][4 * q] = v[0]; tile[l + 16 * i][4 * q + 1] = v[1]; tile[l + 16 * i][4 * q + 2] = v[2]; tile[l + 16 * i][4 * q + 3] = v[3];
		}
	}
	__syncthreads();
#pragma unroll
	for (int i = 0; i < 4; i++) {
		const int c = c0 + l + 16 * i, r = r0 + 4 * q;
		if (c < C && r < R) {
			const T4 v = { tile[4 * q][l + 16 * i], tile[4 * q + 1][l + 16 * i], tile[4 * q + 2][l + 16 * i], tile[4 * q + 3][l + 16 * i] };
			*(T4*)(out + base + (size_t)c * R + r) = v;
		}
	}
}

// The same tile transpose with a type change on the way through LDS: half-precision NCHW tensors become the fp32 NHWC images the
// fp32 convolution kernels read (and back) in ONE pass each -- the layout change the NCHW path needs anyway carries the conversion,
// where converting first and transposing second moved every element twice more.
template <typename TI, typename TO>
__global__ void __launch_bounds__(256) transpose_convert_kernel(const TI* in, TO* out, const int R, const int C)
{
	__shared__ float tile[TT][TT + 1];
	const int c0 = blockIdx.x * TT, r0 = blockIdx.y * TT;
	const size_t base = (size_t)blockIdx.z * R * C;
	const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
	for (int j = ty; j < TT; j += 4) {
		const int r = r0 + j, c = c0 + tx;
		if (r < R && c < C) tile[j][tx] = (float)in[base + (size_t)r * C + c];
	}
	__syncthreads();
	for (int j = ty; j < TT; j += 4) {
		const int c = c0 + j, r = r0 + tx;
		if (r < R && c < C) out[base + (size_t)c * R + r] = (TO)tile[tx][j];
	}
}

// ... and its four-elements-per-access form (8-byte accesses on the half side, 16-byte on the float side; same conditions as transpose_tile_vec4_kernel)
template <typename TI, typename TO>
__global__ void __launch_bounds__(256) transpose_convert_vec4_kernel(const TI* __restrict__ in, TO* __restrict__ out, const int R, const int C)
{
	typedef TI TI4 __attribute__((ext_vector_type(4)));
	typedef TO TO4 __attribute__((ext_vector_type(4)));
	__shared__ float tile[TT][TT + 1];
	const int c0 = blockIdx.x * TT, r0 = blockIdx.y * TT;
	const size_t base = (size_t)blockIdx.z * R * C;
	const int q = threadIdx.x & 15, l = threadIdx.x >> 4;
#pragma unroll
	for (int i = 0; i < 4; i++) {
		const int r = r0 + l + 16 * i, c = c0 + 4 * q;
		if (r < R && c < C) {
			const TI4 v = *(const TI4*)(in + base + (size_t)r * C + c);
			tile[l + 16 * i][4 * q] = (float)v[0]; tile[l + 16 * i][4 * q + 1] = (float)v[1]; tile[l + 16 * i][4 * q + 2] = (float)v[2]; tile[l + 16 * i][4 * q + 3] = (float)v[3];
		}
	}
	__syncthreads();
#pragma unroll
	for (int i = 0; i < 4; i++) {
		const int c = c0 + l + 16 * i, r = r0 + 4 * q;
		if (c < C && r < R) {
			const TO4 v = { (TO)tile[4 * q][l + 16 * i], (TO)tile[4 * q + 1][l + 16 * i], (TO)tile[4 * q + 2][l + 16 * i], (TO)tile[4 * q + 3][l + 16 * i] };
			*(TO4*)(out + base + (size_t)c * R + r) = v;
		}
	}
}

// Halves to halves in 16-byte accesses (round 6): the NCHW <-> NHWC re-layouts around the half-precision 3 x 3 convolutions were 3.6 - 6.2 % of the f16 trainers'
// kernel time in 8-byte accesses through an fp32 tile (an 8-byte access moves 0.54 - 0.70 of the 16-byte rate on this chip).  A 64 x 64 tile of halves goes into
// LDS as it arrives (one ds_write_b128 per lane, rows 192 bytes apart) and comes back out transposed through gfx950's LDS transpose read (ds_read_b64_tr_b16,
// isa.h tr_read4: a 16-lane group reads a [4 rows][16 columns] block, lane i receives column i): two reads give a lane eight consecutive input rows of one
// column = 16 contiguous bytes of an output row; a wave writes 16 output rows x 64 contiguous bytes per store.  R, C multiples of 8, 16-byte aligned tensors.
constexpr int TH_PITCH = TT + 32; // halves per tile row: 192 bytes = 48 dwords (the four rows of a transpose-read block land 16 banks apart: mfma_gemm_f16.h gemm16_npitch)
// SUM: every input row's sum over the tile's columns goes to row_partial[(batch entry * tile columns + tile column) * R + row] (fp32; common.h transpose_half_rowsum)
template <bool SUM>
__global__ void __launch_bounds__(256) transpose_half8_kernel(const nnc::half_t* __restrict__ in, nnc::half_t* __restrict__ out, const int R, const int C, float* __restrict__ row_partial)
{
	using namespace nnc;
	typedef unsigned int u4 __attribute__((ext_vector_type(4)));
	__shared__ __attribute__((aligned(16))) half_t tile[TT * TH_PITCH];
	const int c0 = blockIdx.x * TT, r0 = blockIdx.y * TT;
	const size_t base = (size_t)blockIdx.z * R * C;
	const int t = threadIdx.x;
#pragma unroll
	for (int j = 0; j < 2; j++) {
		const int id = t + 256 * j;
		const int r = id >> 3, cc = (id & 7) << 3;
		u4 v = { 0, 0, 0, 0 };
		if (r0 + r < R && c0 + cc < C) v = *(const u4*)(in + base + (size_t)(r0 + r) * C + c0 + cc);
		*(u4*)(tile + r * TH_PITCH + cc) = v;
		if (SUM) { // the eight lanes id & 7 hold the 64 columns of row r (an out-of-range chunk is zeros)
			typedef half_t h8 __attribute__((ext_vector_type(8)));
			const h8 hv = __builtin_bit_cast(h8, v);
			float sum = (((float)hv[0] + (float)hv[1]) + ((float)hv[2] + (float)hv[3])) + (((float)hv[4] + (float)hv[5]) + ((float)hv[6] + (float)hv[7]));
			sum += __shfl_xor(sum, 1, 64);
			sum += __shfl_xor(sum, 2, 64);
			sum += __shfl_xor(sum, 4, 64);
			if ((id & 7) == 0 && r0 + r < R) row_partial[((size_t)blockIdx.z * gridDim.x + blockIdx.x) * R + r0 + r] = sum;
		}
	}
	__syncthreads();
	const int lane = t & 63, w = t >> 6, g = lane >> 4, i = lane & 15;
#pragma unroll
	for (int p = 0; p < 2; p++) {
		const int cblk = p * 2 + (w >> 1), rchunk = (w & 1) * 4 + g;
		const half_t* const blk = tile + (rchunk * 8) * TH_PITCH + cblk * 16;
		halfx4 lo = tr_read4(blk, TH_PITCH, i), hi = tr_read4(blk + 4 * TH_PITCH, TH_PITCH, i);
		NNC_WAIT_LGKM0(); // (the transpose reads are asm: hipcc does not count them; the pins tie the uses below behind the wait)
		NNC_PIN_VEC(lo); NNC_PIN_VEC(hi);
		const halfx8 v = { lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3] };
		const int c = c0 + cblk * 16 + i, r = r0 + rchunk * 8;
		if (c < C && r < R) *(halfx8*)(out + base + (size_t)c * R + r) = v;
	}
}

template <typename TI, typename TO>
__global__ void __launch_bounds__(256) convert_kernel(const TI* in, TO* out, const size_t n)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (TO)in[i];
}

// Logical (n, h, w, c) extents + element strides of an up-to-4-d tensor / view in any of the three formats.
struct logical4_t { int d[4]; long s[4]; };
static bool logical4(const ccv_nnc_tensor_t* t, logical4_t* o)
{
	const int nd = tensor_nd(t->info.dim);
	int st[CCV_NNC_MAX_DIM_ALLOC];
	tensor_strides(t, st);
	const int* dim = t->info.dim;
	for (int i = 0; i < 4; i++) { o->d[i] = 1; o->s[i] = 0; }
	if (nd > 4 || nd < 1) return false;
	// Which logical axis each stored dim is, by rank (ccv_nnc_tensor_get_n / _c, lib/nnc/ccv_nnc_easy.h:340-403):
	//   4-d: the full format;  3-d: no batch;  2-d: (n, c);  1-d: (c)
	int axis[4]; // logical axis (0 n, 1 h, 2 w, 3 c) of stored dim i
	if (nd == 1) axis[0] = 3;
	else if (nd == 2) { axis[0] = 0; axis[1] = 3; if (t->info.format == CCV_TENSOR_FORMAT_CHWN) { axis[0] = 3; axis[1] = 0; } }
	else {
		const int full[3][4] = { { 0, 1, 2, 3 } /* NHWC */, { 0, 3, 1, 2 } /* NCHW */, { 3, 1, 2, 0 } /* CHWN */ };
		const int* f = t->info.format == CCV_TENSOR_FORMAT_NHWC ? full[0] : t->info.format == CCV_TENSOR_FORMAT_NCHW ? full[1] : t->info.format == CCV_TENSOR_FORMAT_CHWN ? full[2] : 0;
		if (!f) return false;
		if (nd == 4) for (int i = 0; i < 4; i++) axis[i] = f[i];
		else { int k = 0; for (int i = 0; i < 4; i++) if (f[i] != 0) axis[k++] = f[i]; } // drop the batch axis
	}
	if (t->info.format != CCV_TENSOR_FORMAT_NHWC && t->info.format != CCV_TENSOR_FORMAT_NCHW && t->info.format != CCV_TENSOR_FORMAT_CHWN) return false;
	for (int i = 0; i < nd; i++) { o->d[axis[i]] = dim[i]; o->s[axis[i]] = st[i]; }
	return true;
}

template <typename T>
static int launch_permute(const void* in, void* out, const perm4_t& p, ccv_nnc_stream_context_t* ctx)
{
	const size_t n = (size_t)p.dim[0] * p.dim[1] * p.dim[2] * p.dim[3];
	if (n == 0) return CCV_NNC_EXEC_SUCCESS;
	hipLaunchKernelGGL(HIP_KERNEL_NAME(permute4_kernel<T>), dim3(grid_for(n, 256)), dim3(256), 0, stream_of(ctx), (const T*)in, (T*)out, p, n);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}
// Filters between [K][C][kh * kw] (the NCHW tensors' filter layout) and [K][kh * kw][C]: per output channel a [R][S] -> [S][R] transpose of a few KB whose one
// side is 9 (or 49) long.  Through the 64 x 64 tile kernel above that is a grid of K x ceil(C / 64) workgroups with 9 of every 64 lanes working: ~28 us per
// call on ResNet-50's filters, three calls per convolution per step (forward, data gradient, filter gradient back) -- 1.5 ms of config 4's step for 100 MB.
// Here: one workgroup per filter; the R * S contiguous words come in as they lie (coalesced), cross through LDS, and leave as they will lie (coalesced).
constexpr int WT_BYTES = 65536; // one matrix in LDS: 512 x 9 or 1024 x 9 floats, 128 x 49 floats, 512 x 49 halves ...
// Round 6, last session: the LDS a workgroup takes follows the matrix (8 / 16 / 32 / 64 KB instances; one static 64 KB kept two workgroups on a CU), the matrix comes in 16 bytes per lane
// where it is whole chunks, and leaves 16 bytes per lane as well: EIGHT (four for 4-byte elements) consecutive outputs gathered from LDS, one division per chunk instead of one per element.  ResNet-50's 97 (fp32: 143) launches per step ran 10.5 us each for 0.1 - 4.7 MB.
template <typename T, int BYTES>
__global__ void __launch_bounds__(256) filter_transpose_kernel(const T* __restrict__ in, T* __restrict__ out, const int R, const int S)
{
	__shared__ __attribute__((aligned(16))) T buf[BYTES / sizeof(T)];
	typedef unsigned int u4 __attribute__((ext_vector_type(4)));
	constexpr int PER = 16 / (int)sizeof(T);
	const int n = R * S;
	const size_t base = (size_t)blockIdx.x * n;
	const bool wide = (n % PER) == 0 && ((((uintptr_t)in) | ((uintptr_t)out)) & 15) == 0; // (whole 16-byte chunks per matrix: every matrix of the batch starts on one)
	if (wide) {
		for (int i = threadIdx.x; i < n / PER; i += 256) ((u4*)buf)[i] = ((const u4*)(in + base))[i];
		__syncthreads();
		for (int c = threadIdx.x; c < n / PER; c += 256) { // outputs o = s * R + r, PER consecutive ones per lane: out[o] <- buf[r * S + s]
			const int o0 = c * PER;
			int s_ = o0 / R, r = o0 - s_ * R;
			union { u4 q; T v[PER]; } u;
#pragma unroll
			for (int e = 0; e < PER; e++) {
				u.v[e] = buf[r * S + s_];
				if (++r == R) { r = 0; ++s_; }
			}
			*(u4*)(out + base + o0) = u.q;
		}
		return;
	}
	for (int i = threadIdx.x; i < n; i += 256) buf[i] = in[base + i];
	__syncthreads();
	for (int o = threadIdx.x; o < n; o += 256) { // o = s * R + r  <-  r * S + s
		const int s_ = o / R, r = o - s_ * R;
		out[base + o] = buf[r * S + s_];
	}
}
template <typename T>
static bool filter_transpose_fits(const int R, const int S) { return (R < 64 || S < 64) && (long)R * S * (long)sizeof(T) <= WT_BYTES; }
template <typename T>
static int filter_transpose(const void* in, void* out, int K, int R, int S, ccv_nnc_stream_context_t* ctx)
{
	const size_t bytes = (size_t)R * S * sizeof(T);
	hipStream_t stream = stream_of(ctx);
	if (bytes <= 8192) hipLaunchKernelGGL(HIP_KERNEL_NAME(filter_transpose_kernel<T, 8192>), dim3(K), dim3(256), 0, stream, (const T*)in, (T*)out, R, S);
	else if (bytes <= 16384) hipLaunchKernelGGL(HIP_KERNEL_NAME(filter_transpose_kernel<T, 16384>), dim3(K), dim3(256), 0, stream, (const T*)in, (T*)out, R, S);
	else if (bytes <= 32768) hipLaunchKernelGGL(HIP_KERNEL_NAME(filter_transpose_kernel<T, 32768>), dim3(K), dim3(256), 0, stream, (const T*)in, (T*)out, R, S);
	else hipLaunchKernelGGL(HIP_KERNEL_NAME(filter_transpose_kernel<T, WT_BYTES>), dim3(K), dim3(256), 0, stream, (const T*)in, (T*)out, R, S);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}

template <typename T>
static int launch_transpose(const void* in, void* out, int batch, int R, int C, ccv_nnc_stream_context_t* ctx)
{
	if (batch <= 0 || R <= 0 || C <= 0) return CCV_NNC_EXEC_SUCCESS;
	if (filter_transpose_fits<T>(R, C)) return filter_transpose<T>(in, out, batch, R, C, ctx); // (thin matrices: one workgroup each, see above)
	if constexpr (sizeof(T) == 4) {
		if (R % 4 == 0 && C % 4 == 0 && (((uintptr_t)in | (uintptr_t)out) & 15) == 0) {
			hipLaunchKernelGGL(HIP_KERNEL_NAME(transpose_tile_vec4_kernel<T>), dim3((C + TT - 1) / TT, (R + TT - 1) / TT, batch), dim3(256), 0, stream_of(ctx), (const T*)in, (T*)out, R, C);
			HIP_ENFORCE(hipGetLastError());
			return CCV_NNC_EXEC_SUCCESS;
		}
	}
	hipLaunchKernelGGL(HIP_KERNEL_NAME(transpose_tile_kernel<T>), dim3((C + TT - 1) / TT, (R + TT - 1) / TT, batch), dim3(256), 0, stream_of(ctx), (const T*)in, (T*)out, R, C);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}
// 2-byte elements that are halves (the only 2-byte tensor type the rows register): four per access through the float tile of the converting
// transpose -- half -> float -> half is exact
static int launch_transpose_half(const void* in, void* out, int batch, int R, int C, ccv_nnc_stream_context_t* ctx)
{
	static const int wide = !(getenv("NNC_MI355X_TRANSPOSE_HALF8") && *getenv("NNC_MI355X_TRANSPOSE_HALF8") == '0');
	if (wide && batch > 0 && R > 0 && C > 0 && R % 8 == 0 && C % 8 == 0 && (((uintptr_t)in | (uintptr_t)out) & 15) == 0) {
		hipLaunchKernelGGL(HIP_KERNEL_NAME(transpose_half8_kernel<false>), dim3((C + TT - 1) / TT, (R + TT - 1) / TT, batch), dim3(256), 0, stream_of(ctx), (const half_t*)in, (half_t*)out, R, C, (float*)0);
		HIP_ENFORCE(hipGetLastError());
		return CCV_NNC_EXEC_SUCCESS;
	}
	if (batch > 0 && R > 0 && C > 0 && R % 4 == 0 && C % 4 == 0 && (((uintptr_t)in | (uintptr_t)out) & 7) == 0) {
		hipLaunchKernelGGL(HIP_KERNEL_NAME(transpose_convert_vec4_kernel<half_t, half_t>), dim3((C + TT - 1) / TT, (R + TT - 1) / TT, batch), dim3(256), 0, stream_of(ctx), (const half_t*)in, (half_t*)out, R, C);
		HIP_ENFORCE(hipGetLastError());
		return CCV_NNC_EXEC_SUCCESS;
	}
	return launch_transpose<uint16_t>(in, out, batch, R, C, ctx);
}

static int by_size_permute(size_t es, const void* in, void* out, const perm4_t& p, ccv_nnc_stream_context_t* ctx)
{
	switch (es) {
		case 1: return launch_permute<uint8_t>(in, out, p, ctx);
		case 2: return launch_permute<uint16_t>(in, out, p, ctx);
		case 4: return launch_permute<uint32_t>(in, out, p, ctx);
		case 8: return launch_permute<uint64_t>(in, out, p, ctx);
	}
	return CCV_NNC_EXEC_INVALID;
}
static int by_size_transpose(size_t es, const void* in, void* out, int batch, int R, int C, ccv_nnc_stream_context_t* ctx)
{
	switch (es) {
		case 2: return launch_transpose_half(in, out, batch, R, C, ctx);
		case 4: return launch_transpose<uint32_t>(in, out, batch, R, C, ctx);
		case 8: return launch_transpose<uint64_t>(in, out, batch, R, C, ctx);
	}
	return CCV_NNC_EXEC_INVALID;
}

} // namespace

namespace nnc {

// b = a re-laid-out: same logical (n, h, w, c) contents, each tensor in its own format / strides.
int format_transform(const ccv_nnc_tensor_t* a, ccv_nnc_tensor_t* b, ccv_nnc_stream_context_t* ctx)
{
	logical4_t la, lb;
	if (!logical4(a, &la) || !logical4(b, &lb)) return CCV_NNC_EXEC_INVALID;
	for (int i = 0; i < 4; i++) if (la.d[i] != lb.d[i]) return CCV_NNC_EXEC_INVALID;
	const size_t es = datatype_size(a->info.datatype);
	if (es != datatype_size(b->info.datatype) || es == 0) return CCV_NNC_EXEC_INVALID;
	const bool dense = tensor_contiguous(a) && tensor_contiguous(b);
	const int N = la.d[0], HW = la.d[1] * la.d[2], C = la.d[3];
	if (dense && es >= 2 && a->info.format != b->info.format && HW > 1 && C > 1) {
		if (a->info.format == CCV_TENSOR_FORMAT_NCHW && b->info.format == CCV_TENSOR_FORMAT_NHWC) return by_size_transpose(es, a->data.u8, b->data.u8, N, C, HW, ctx);
		if (a->info.format == CCV_TENSOR_FORMAT_NHWC && b->info.format == CCV_TENSOR_FORMAT_NCHW) return by_size_transpose(es, a->data.u8, b->data.u8, N, HW, C, ctx);
	}
	// generic: iterate in b's memory order (axes sorted by b's stride, largest first)
	int order[4] = { 0, 1, 2, 3 };
	for (int i = 0; i < 4; i++)
		for (int j = i + 1; j < 4; j++) {
			const long si = lb.d[order[i]] == 1 ? -1 : lb.s[order[i]], sj = lb.d[order[j]] == 1 ? -1 : lb.s[order[j]];
			if (sj > si) { const int t = order[i]; order[i] = order[j]; order[j] = t; }
		}
	perm4_t p;
	int k = 0;
	for (int i = 0; i < 4; i++) if (lb.d[order[i]] != 1) { p.dim[k] = lb.d[order[i]]; p.is[k] = la.s[order[i]]; p.os[k] = lb.s[order[i]]; k++; }
	for (; k < 4; k++) { // pad on the OUTER side with unit axes
		for (int m = 3; m > 0; m--) { p.dim[m] = p.dim[m - 1]; p.is[m] = p.is[m - 1]; p.os[m] = p.os[m - 1]; }
		p.dim[0] = 1; p.is[0] = 0; p.os[0] = 0;
	}
	return by_size_permute(es, a->data.u8, b->data.u8, p, ctx);
}

// Dense weight layout change [K][C][kh][kw] (NCHW-format weights) -> [K][kh][kw][C] (the layout the kernels read).
// in[batch][R][C] -> out[batch][C][R], halves in / floats out and the reverse
int transpose_half(const void* in, void* out, int batch, int R, int C, ccv_nnc_stream_context_t* ctx) { return launch_transpose_half(in, out, batch, R, C, ctx); }
int transpose_half_rowsum(const void* in, void* out, int batch, int R, int C, float* row_partial, ccv_nnc_stream_context_t* ctx)
{
	static_assert(TT == 64, "transpose_half_rowsum_slices (common.h) counts 64-column tiles");
	if (!(batch > 0 && R > 0 && C > 0 && R % 8 == 0 && C % 8 == 0 && (((uintptr_t)in | (uintptr_t)out) & 15) == 0) || !row_partial) return CCV_NNC_EXEC_NO_KERNEL;
	hipLaunchKernelGGL(HIP_KERNEL_NAME(transpose_half8_kernel<true>), dim3((C + TT - 1) / TT, (R + TT - 1) / TT, batch), dim3(256), 0, stream_of(ctx), (const half_t*)in, (half_t*)out, R, C, row_partial);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}
int transpose_half_to_float(const void* in, float* out, int batch, int R, int C, ccv_nnc_stream_context_t* ctx)
{
	if (batch <= 0 || R <= 0 || C <= 0) return CCV_NNC_EXEC_SUCCESS;
	if (R % 4 == 0 && C % 4 == 0 && ((uintptr_t)in & 7) == 0 && ((uintptr_t)out & 15) == 0) {
		hipLaunchKernelGGL(HIP_KERNEL_NAME(transpose_convert_vec4_kernel<half_t, float>), dim3((C + TT - 1) / TT, (R + TT - 1) / TT, batch), dim3(256), 0, stream_of(ctx), (const half_t*)in, out, R, C);
		HIP_ENFORCE(hipGetLastError());
		return CCV_NNC_EXEC_SUCCESS;
	}
	hipLaunchKernelGGL(HIP_KERNEL_NAME(transpose_convert_kernel<half_t, float>), dim3((C + TT - 1) / TT, (R + TT - 1) / TT, batch), dim3(256), 0, stream_of(ctx), (const half_t*)in, out, R, C);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}
int transpose_float_to_half(const float* in, void* out, int batch, int R, int C, ccv_nnc_stream_context_t* ctx)
{
	if (batch <= 0 || R <= 0 || C <= 0) return CCV_NNC_EXEC_SUCCESS;
	if (R % 4 == 0 && C % 4 == 0 && ((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 7) == 0) {
		hipLaunchKernelGGL(HIP_KERNEL_NAME(transpose_convert_vec4_kernel<float, half_t>), dim3((C + TT - 1) / TT, (R + TT - 1) / TT, batch), dim3(256), 0, stream_of(ctx), in, (half_t*)out, R, C);
		HIP_ENFORCE(hipGetLastError());
		return CCV_NNC_EXEC_SUCCESS;
	}
	hipLaunchKernelGGL(HIP_KERNEL_NAME(transpose_convert_kernel<float, half_t>), dim3((C + TT - 1) / TT, (R + TT - 1) / TT, batch), dim3(256), 0, stream_of(ctx), in, (half_t*)out, R, C);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}
int weights_nchw_to_nhwc(const float* w, float* out, int K, int C, int khw, ccv_nnc_stream_context_t* ctx)
{
	return launch_transpose<uint32_t>(w, out, K, C, khw, ctx);
}
int weights_nhwc_to_nchw(const float* w, float* out, int K, int C, int khw, ccv_nnc_stream_context_t* ctx)
{
	return launch_transpose<uint32_t>(w, out, K, khw, C, ctx);
}

} // namespace nnc

namespace {

static int _format_transform(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	if (output_size > input_size) return CCV_NNC_EXEC_INVALID;
	for (int i = 0; i < output_size; i++) {
		const ccv_nnc_tensor_t* a = inputs[i];
		ccv_nnc_tensor_t* b = outputs[i];
		if (!a || !b || a == b) return CCV_NNC_EXEC_INVALID; // no in-place transform (util_cpu_ref.c:1005)
		if (a->info.dim[0] == 0 || b->info.dim[0] == 0) continue;
		const int r = format_transform(a, b, stream_context);
		if (r != CCV_NNC_EXEC_SUCCESS) return r;
	}
	return CCV_NNC_EXEC_SUCCESS;
}

static int _transpose(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	if (output_size > input_size) return CCV_NNC_EXEC_INVALID;
	const int ax0 = cmd.info.transpose.axis[0], ax1 = cmd.info.transpose.axis[1];
	for (int k = 0; k < output_size; k++) {
		const ccv_nnc_tensor_t* a = inputs[k];
		ccv_nnc_tensor_t* b = outputs[k];
		if (!a || !b) return CCV_NNC_EXEC_INVALID;
		const int nd = tensor_nd(a->info.dim);
		if (nd != tensor_nd(b->info.dim) || nd < 2 || nd > 4 || ax0 < 0 || ax1 < 0 || ax0 >= nd || ax1 >= nd) return CCV_NNC_EXEC_INVALID;
		int as[CCV_NNC_MAX_DIM_ALLOC], bs[CCV_NNC_MAX_DIM_ALLOC];
		tensor_strides(a, as);
		tensor_strides(b, bs);
		perm4_t p;
		for (int x = 0; x < 4; x++) { p.dim[x] = 1; p.is[x] = 0; p.os[x] = 0; }
		for (int x = 0; x < nd; x++) { // b's axis x reads a's axis (x swapped)
			const int ax = x == ax0 ? ax1 : x == ax1 ? ax0 : x;
			if (b->info.dim[x] != a->info.dim[ax]) return CCV_NNC_EXEC_INVALID;
			p.dim[4 - nd + x] = b->info.dim[x]; p.os[4 - nd + x] = bs[x]; p.is[4 - nd + x] = as[ax];
		}
		const size_t es = datatype_size(a->info.datatype);
		if (es != datatype_size(b->info.datatype)) return CCV_NNC_EXEC_INVALID;
		const int r = by_size_permute(es, a->data.u8, b->data.u8, p, stream_context);
		if (r != CCV_NNC_EXEC_SUCCESS) return r;
	}
	return CCV_NNC_EXEC_SUCCESS;
}

template <typename TI, typename TO>
static int convert(const void* in, void* out, size_t n, ccv_nnc_stream_context_t* ctx)
{
	if (n == 0) return CCV_NNC_EXEC_SUCCESS;
	hipLaunchKernelGGL(HIP_KERNEL_NAME(convert_kernel<TI, TO>), dim3(grid_for(n, 256)), dim3(256), 0, stream_of(ctx), (const TI*)in, (TO*)out, n);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}

static int _datatype_conversion(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	if (output_size > input_size) return CCV_NNC_EXEC_INVALID;
	for (int i = 0; i < output_size; i++) {
		const ccv_nnc_tensor_t* a = inputs[i];
		ccv_nnc_tensor_t* b = outputs[i];
		if (!a || !b || a == b) return CCV_NNC_EXEC_INVALID;
		const int da = CCV_GET_DATA_TYPE(a->info.datatype), db = CCV_GET_DATA_TYPE(b->info.datatype);
		if (da == db) { const int r = format_transform(a, b, stream_context); if (r) return r; continue; } // plain (possibly strided) copy
		if (!tensor_contiguous(a) || !tensor_contiguous(b)) return CCV_NNC_EXEC_INVALID;
		const size_t n = tensor_count(a->info);
		if (n != tensor_count(b->info)) return CCV_NNC_EXEC_INVALID;
		int r = CCV_NNC_EXEC_INVALID;
		if (da == CCV_32F && db == CCV_16F) r = convert<float, half_t>(a->data.u8, b->data.u8, n, stream_context);
		else if (da == CCV_16F && db == CCV_32F) r = convert<half_t, float>(a->data.u8, b->data.u8, n, stream_context);
		else if (da == CCV_64F && db == CCV_32F) r = convert<double, float>(a->data.u8, b->data.u8, n, stream_context);
		else if (da == CCV_32F && db == CCV_64F) r = convert<float, double>(a->data.u8, b->data.u8, n, stream_context);
		else if (da == CCV_64F && db == CCV_16F) r = convert<double, half_t>(a->data.u8, b->data.u8, n, stream_context);
		else if (da == CCV_16F && db == CCV_64F) r = convert<half_t, double>(a->data.u8, b->data.u8, n, stream_context);
		if (r != CCV_NNC_EXEC_SUCCESS) return r;
	}
	return CCV_NNC_EXEC_SUCCESS;
}

} // namespace

#define NNC_REG(CMD, BACKEND, FORMATS, DATATYPES, MEMORY, EXEC) \
	extern "C" void _register_command_##CMD##_backend_##BACKEND(ccv_nnc_cmd_backend_registry_t* const registry) \
	{ registry->tensor_formats = (FORMATS); registry->tensor_datatypes = (DATATYPES); registry->tensor_memory = (MEMORY); registry->algorithms = 1; registry->exec = EXEC; NNC_HALF_STAGED(registry, EXEC); }
#define ALL_FORMATS (CCV_TENSOR_FORMAT_NCHW | CCV_TENSOR_FORMAT_NHWC | CCV_TENSOR_FORMAT_CHWN)
#define MOVABLE_TYPES (CCV_64F | CCV_32F | CCV_16F | CCV_64S | CCV_32S | CCV_8U)

NNC_REG(CCV_NNC_FORMAT_TRANSFORM_FORWARD, CCV_NNC_BACKEND_GPU_CUDNN, ALL_FORMATS, MOVABLE_TYPES, CCV_TENSOR_GPU_MEMORY, _format_transform)
NNC_REG(CCV_NNC_FORMAT_TRANSFORM_BACKWARD, CCV_NNC_BACKEND_GPU_CUDNN, ALL_FORMATS, MOVABLE_TYPES, CCV_TENSOR_GPU_MEMORY, _format_transform)
NNC_REG(CCV_NNC_TRANSPOSE_FORWARD, CCV_NNC_BACKEND_GPU_CUDNN, ALL_FORMATS, MOVABLE_TYPES, CCV_TENSOR_GPU_MEMORY, _transpose)
NNC_REG(CCV_NNC_TRANSPOSE_BACKWARD, CCV_NNC_BACKEND_GPU_CUDNN, ALL_FORMATS, MOVABLE_TYPES, CCV_TENSOR_GPU_MEMORY, _transpose)
NNC_REG(CCV_NNC_DATATYPE_CONVERSION_FORWARD, CCV_NNC_BACKEND_GPU_REF, ALL_FORMATS, CCV_64F | CCV_32F | CCV_16F, CCV_TENSOR_GPU_MEMORY, _datatype_conversion)
NNC_REG(CCV_NNC_DATATYPE_CONVERSION_BACKWARD, CCV_NNC_BACKEND_GPU_REF, ALL_FORMATS, CCV_64F | CCV_32F | CCV_16F, CCV_TENSOR_GPU_MEMORY, _datatype_conversion)
