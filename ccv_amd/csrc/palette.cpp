// Palettized tensors (CCV_QX) on the MI355X: weights stored as 4 / 5 / 6 / 7 / 8-bit indices into a per-block palette of 16F / 32F / 64F values.
//
// Replaces lib/nnc/gpu/ccv_nnc_palettize.cu (ccv_nnc_compat_depalettize, lib/nnc/gpu/ccv_nnc_compat.h:59) and the depalettize-then-run prologue the reference's
// GPU rows carry for their weights (GEMM: lib/nnc/cmd/blas/gpu/ccv_nnc_gemm_gpu_cublas.cu:289-334, 658-752; convolution:
// lib/nnc/cmd/convolution/gpu/ccv_nnc_conv_gpu_cudnn.cu:79-95, 328-345; transposed convolution: ccv_nnc_conv_transpose_gpu_cudnn.cu:72, 127; attention's head
// projection: ccv_nnc_scaled_dot_product_attention_flash_attn.cu:123, 342).  The layout is the one the host's quantiser writes
// (lib/nnc/ccv_nnc_palettize.c:9-208, read back by :211-956):
//   the tensor's `count` elements are cut into blocks of `number_in_blocks`; block i = [ palette: 2^qbits elements ][ indices of its elements ];
//   indices are a big-endian bit stream -- element j of a block occupies bits [j q, (j + 1) q) counted from the most significant bit of the first index byte;
//   every block but the last is full; the last one's index bytes stop after the last whole group the quantiser wrote (2 elements for 4 bits, 8 for 5 and 7,
//   4 for 6, 1 for 8), and every block's stride uses the reference's integer divisions (number_in_blocks / 8 * 5, ...), whatever number_in_blocks is.
//
// This is HBM-bound byte work -- q / 8 bytes read and 2 / 4 / 8 bytes written per element -- so the kernels are about the store side:
//   * a lane owns a GROUP of eight consecutive elements: q consecutive index bytes in, 16 / 32 / 64 consecutive bytes out (16-byte stores), consecutive lanes
//     consecutive groups, so a wave reads one contiguous run of index bytes and writes one contiguous run of values;
//   * blocks of at least 2048 elements (a group for every lane of a workgroup) and PALETTE_LDS_REUSE uses per palette entry stage the palette in LDS once
//     per workgroup (grid: block x chunk of its groups); smaller blocks (the reference's tests use 128 elements with up to 256-entry palettes: most entries
//     are never looked up) read the palette through the vector cache instead -- one flat grid over all groups, no workgroup idles on a 16-group block.
// A command that meets a CCV_QX input runs on a dense image of it: depalettized_exec() below.
#include "common.h"

namespace nnc {
namespace {

constexpr int PAL_THREADS = 256;
constexpr int PALETTE_LDS_REUSE = 4;   // stage the palette in LDS when a block has >= this many elements per palette entry
constexpr int PAL_GROUPS_PER_LANE = 4; // LDS form: groups a lane walks (stride PAL_THREADS) per workgroup

// index bytes of a FULL block, as the host's quantiser lays them out (ccv_nnc_palettize.c: the `ui0` strides of each bit width)
__host__ __device__ __forceinline__ size_t index_bytes_per_block(const int qbits, const int nib)
{
	switch (qbits) {
		case 4: return (size_t)(nib / 2);
		case 5: return (size_t)(nib / 8) * 5;
		case 6: return (size_t)(nib / 4) * 3;
		case 7: return (size_t)(nib / 8) * 7;
		default: return (size_t)nib;
	}
}

// One group: up to eight elements starting at element 8 g of a block.  `idx` points at the group's first index byte, `valid` (1 .. 8) elements exist,
// only the index bytes those elements touch are read (the stream of the tensor's last block ends with them).
template <typename ELEM, int Q, typename LUT>
__device__ __forceinline__ void decode_group(const unsigned char* const idx, const int valid, const LUT palette, ELEM* const out, const bool vec)
{
	const int nbytes = valid == 8 ? Q : (valid * Q + 7) >> 3;
	unsigned long long w = 0; // the group's Q bytes as one big-endian number: element j = bits [(7 - j) Q, (8 - j) Q)
#pragma unroll
	for (int i = 0; i < Q; i++) {
		const unsigned long long b = i < nbytes ? idx[i] : 0;
		w |= b << (8 * (Q - 1 - i));
	}
	ELEM v[8];
#pragma unroll
	for (int j = 0; j < 8; j++) v[j] = palette[(unsigned)(w >> (Q * (7 - j))) & ((1u << Q) - 1)];
	if (valid == 8 && vec) { // 16-byte stores: 1 (halves), 2 (floats), 4 (doubles) per group
		typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
		union { ELEM e[8]; u32x4 q[sizeof(ELEM) / 2]; } pack;
#pragma unroll
		for (int j = 0; j < 8; j++) pack.e[j] = v[j];
#pragma unroll
		for (int k = 0; k < (int)sizeof(ELEM) / 2; k++) ((u32x4*)out)[k] = pack.q[k];
	} else
#pragma unroll
		for (int j = 0; j < 8; j++) if (j < valid) out[j] = v[j];
}

struct pal_geom_t {
	size_t count;        // elements of the tensor
	size_t block_stride; // bytes from one block to the next in the palettized stream: palette + a full block's index bytes
	int nib;             // elements per block
	int gpb;             // groups of eight per block (the last one partial when nib is not a multiple of 8)
	int vec;             // out is 16-byte aligned
};

// LDS form: blockIdx.x = block, blockIdx.y = chunk of PAL_THREADS * PAL_GROUPS_PER_LANE groups of it.
template <typename ELEM, int Q>
__global__ void __launch_bounds__(PAL_THREADS) depalettize_lds_kernel(const unsigned char* const in, ELEM* const out, const pal_geom_t g)
{
	__shared__ ELEM palette[1 << Q];
	const size_t block = blockIdx.x;
	const unsigned char* const base = in + block * g.block_stride;
	for (int i = threadIdx.x; i < (1 << Q); i += PAL_THREADS) palette[i] = ((const ELEM*)base)[i];
	__syncthreads();
	const size_t e0 = block * (size_t)g.nib;
	const size_t left = g.count - e0;
	const int n = left < (size_t)g.nib ? (int)left : g.nib; // elements of this block
	const int groups = (n + 7) >> 3;
	const unsigned char* const idx = base + sizeof(ELEM) * (1 << Q);
	const int g0 = blockIdx.y * (PAL_THREADS * PAL_GROUPS_PER_LANE);
#pragma unroll
	for (int k = 0; k < PAL_GROUPS_PER_LANE; k++) {
		const int gi = g0 + k * PAL_THREADS + (int)threadIdx.x;
		if (gi >= groups) break;
		const int valid = n - gi * 8 < 8 ? n - gi * 8 : 8;
		decode_group<ELEM, Q>(idx + (size_t)gi * Q, valid, palette, out + e0 + (size_t)gi * 8, g.vec && (g.nib & 7) == 0);
	}
}

// Flat form: one lane per group over the whole tensor, the palette read through the vector cache.
template <typename ELEM, int Q>
__global__ void __launch_bounds__(PAL_THREADS) depalettize_flat_kernel(const unsigned char* const in, ELEM* const out, const pal_geom_t g, const size_t total_groups)
{
	const size_t stride = (size_t)gridDim.x * PAL_THREADS;
	for (size_t G = (size_t)blockIdx.x * PAL_THREADS + threadIdx.x; G < total_groups; G += stride) {
		const size_t block = G / (size_t)g.gpb;
		const int gi = (int)(G - block * (size_t)g.gpb);
		const size_t e0 = block * (size_t)g.nib;
		const size_t left = g.count - e0;
		const int n = left < (size_t)g.nib ? (int)left : g.nib;
		if (gi * 8 >= n) continue; // (groups past the end of the tensor's last block)
		const int valid = n - gi * 8 < 8 ? n - gi * 8 : 8;
		const unsigned char* const base = in + block * g.block_stride;
		decode_group<ELEM, Q>(base + sizeof(ELEM) * (1 << Q) + (size_t)gi * Q, valid, (const ELEM*)base, out + e0 + (size_t)gi * 8, g.vec && (g.nib & 7) == 0);
	}
}

template <typename ELEM, int Q>
int depalettize_launch(const unsigned char* const in, ELEM* const out, const size_t count, const int nib, hipStream_t stream)
{
	pal_geom_t g;
	g.count = count;
	g.nib = nib;
	g.gpb = (nib + 7) / 8;
	g.block_stride = sizeof(ELEM) * ((size_t)1 << Q) + index_bytes_per_block(Q, nib);
	g.vec = (((uintptr_t)out) & 15) == 0;
	const size_t blocks = (count + (size_t)nib - 1) / (size_t)nib;
	if ((size_t)nib >= (size_t)PALETTE_LDS_REUSE << Q && g.gpb >= PAL_THREADS && blocks <= 0x7fffffffu) { // (a block must also give every lane of the workgroup a group)
		const int chunk = PAL_THREADS * PAL_GROUPS_PER_LANE;
		const unsigned chunks = (unsigned)((g.gpb + chunk - 1) / chunk);
		if (chunks <= 65535u) {
			hipLaunchKernelGGL((depalettize_lds_kernel<ELEM, Q>), dim3((unsigned)blocks, chunks), dim3(PAL_THREADS), 0, stream, in, out, g);
			HIP_ENFORCE(hipGetLastError());
			note_kernel("depalettize_lds");
			return CCV_NNC_EXEC_SUCCESS;
		}
	}
	const size_t total_groups = blocks * (size_t)g.gpb;
	hipLaunchKernelGGL((depalettize_flat_kernel<ELEM, Q>), dim3(grid_for(total_groups, PAL_THREADS)), dim3(PAL_THREADS), 0, stream, in, out, g, total_groups);
	HIP_ENFORCE(hipGetLastError());
	note_kernel("depalettize_flat");
	return CCV_NNC_EXEC_SUCCESS;
}

template <typename ELEM>
int depalettize_bits(const unsigned char* const in, ELEM* const out, const size_t count, const int qbits, const int nib, hipStream_t stream)
{
	switch (qbits) {
		case 4: return depalettize_launch<ELEM, 4>(in, out, count, nib, stream);
		case 5: return depalettize_launch<ELEM, 5>(in, out, count, nib, stream);
		case 6: return depalettize_launch<ELEM, 6>(in, out, count, nib, stream);
		case 7: return depalettize_launch<ELEM, 7>(in, out, count, nib, stream);
		case 8: return depalettize_launch<ELEM, 8>(in, out, count, nib, stream);
	}
	return CCV_NNC_EXEC_INVALID;
}

} // namespace

size_t palettized_bytes(const int palette_datatype, const size_t count, const int qbits, const int nib)
{ // lib/nnc/ccv_nnc_easy.h:220-238 ccv_nnc_tensor_data_size_without_padding for CCV_QX
	if (nib <= 0 || qbits < 4 || qbits > 8) return 0;
	const size_t blocks = (count + (size_t)nib - 1) / (size_t)nib;
	return ((size_t)1 << qbits) * datatype_size(palette_datatype) * blocks + (count * (size_t)qbits + 7) / 8;
}

int depalettize(const void* const input, const int datatype, const size_t input_length, const int qbits, const int nib, void* const output, const size_t output_length, ccv_nnc_stream_context_t* const ctx)
{
	if (!output_length) return CCV_NNC_EXEC_SUCCESS;
	if (!input || !output || nib <= 0 || qbits < 4 || qbits > 8) return CCV_NNC_EXEC_INVALID;
	// the stream must hold every byte the kernels touch: the palettes of all blocks and the index bytes of every element
	if (input_length < palettized_bytes(datatype, output_length, qbits, nib)) return CCV_NNC_EXEC_INVALID;
	hipStream_t stream = stream_of(ctx);
	const unsigned char* const in = (const unsigned char*)input;
	switch (CCV_GET_DATA_TYPE(datatype)) { // the values are moved, never interpreted: 2-, 4- and 8-byte words
		case CCV_16F: return depalettize_bits<unsigned short>(in, (unsigned short*)output, output_length, qbits, nib, stream);
		case CCV_32F: return depalettize_bits<unsigned int>(in, (unsigned int*)output, output_length, qbits, nib, stream);
		case CCV_64F: return depalettize_bits<unsigned long long>(in, (unsigned long long*)output, output_length, qbits, nib, stream);
	}
	return CCV_NNC_EXEC_INVALID;
}

bool any_palettized(ccv_nnc_tensor_t* const* const inputs, const int input_size)
{
	for (int i = 0; i < input_size; i++)
		if (inputs[i] && CCV_GET_DATA_TYPE(inputs[i]->info.datatype) == CCV_QX) return true;
	return false;
}

// A command with palettized INPUTS (weights; the reference never writes a CCV_QX tensor on the GPU) runs on dense images of them: every CCV_QX input is expanded
// into the stream's palette arena (device_rt.cpp nnc_palette_of: a third grow-only buffer, because the command underneath may grow -- that is, free and
// re-allocate -- both the workspace and the half-staging arena) and handed on through a shadow tensor struct of the palette's datatype.
int depalettized_exec(const nnc_exec_f inner, const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const ctx)
{
	if (!any_palettized(inputs, input_size)) return inner(cmd, hint, flags, inputs, input_size, outputs, output_size, ctx);
	constexpr int MAX_IN = 32;
	if (input_size > MAX_IN) return CCV_NNC_EXEC_INVALID;
	for (int i = 0; i < output_size; i++)
		if (outputs[i] && CCV_GET_DATA_TYPE(outputs[i]->info.datatype) == CCV_QX) return CCV_NNC_EXEC_INVALID;
	ccv_nnc_tensor_view_t shadow[MAX_IN];
	ccv_nnc_tensor_t* in_s[MAX_IN];
	size_t at[MAX_IN], total = 0;
	for (int i = 0; i < input_size; i++) {
		ccv_nnc_tensor_t* const t = inputs[i];
		in_s[i] = t;
		if (!t || CCV_GET_DATA_TYPE(t->info.datatype) != CCV_QX) continue;
		if (CCV_IS_TENSOR_VIEW(t) && !tensor_contiguous(t)) return CCV_NNC_EXEC_INVALID; // the stream has no strides to speak of
		int same = -1;
		for (int j = 0; j < i && same < 0; j++)
			if (inputs[j] && inputs[j]->data.u8 == t->data.u8 && inputs[j]->info.datatype == t->info.datatype && tensor_count(inputs[j]->info) == tensor_count(t->info)) same = j;
		if (same >= 0) { at[i] = at[same]; continue; }
		at[i] = total;
		total += (tensor_count(t->info) * datatype_size((t->info.datatype & 0xff) << 12) + 255) & ~(size_t)255;
	}
	char* const arena = (char*)nnc_palette_of(ctx, total);
	if (total && !arena) return CCV_NNC_EXEC_OOM;
	for (int i = 0; i < input_size; i++) {
		ccv_nnc_tensor_t* const t = inputs[i];
		if (!t || CCV_GET_DATA_TYPE(t->info.datatype) != CCV_QX) continue;
		const int palette_datatype = (t->info.datatype & 0xff) << 12, qbits = (t->info.datatype & 0xf00) >> 8, nib = t->info.reserved;
		const size_t count = tensor_count(t->info);
		bool done = false;
		for (int j = 0; j < i && !done; j++) done = in_s[j] != inputs[j] && at[j] == at[i];
		if (!done) {
			const int ret = depalettize(t->data.u8, palette_datatype, palettized_bytes(palette_datatype, count, qbits, nib), qbits, nib, arena + at[i], count, ctx);
			if (ret != CCV_NNC_EXEC_SUCCESS) return ret;
		}
		memset(&shadow[i], 0, sizeof(ccv_nnc_tensor_view_t));
		memcpy(&shadow[i], t, CCV_IS_TENSOR_VIEW(t) ? sizeof(ccv_nnc_tensor_view_t) : sizeof(ccv_nnc_tensor_t));
		ccv_nnc_tensor_t* const s = (ccv_nnc_tensor_t*)&shadow[i];
		s->info.datatype = palette_datatype;
		s->info.reserved = 0;
		s->data.u8 = (unsigned char*)(arena + at[i]);
		s->dataof = 0;
		s->data_size = 0;
		s->alias_ref = 0;
		in_s[i] = s;
	}
	// the shadows live on this frame: whatever runs underneath runs now, not from the look-ahead's slot (as for the half-staged rows)
	deferred_suppress(1);
	const int ret = inner(cmd, hint, flags, in_s, input_size, outputs, output_size, ctx);
	deferred_suppress(-1);
	return ret;
}

} // namespace nnc

extern "C" int nnc_mi355x_depalettize(const void* input, int datatype, size_t input_length, int qbits, int number_in_blocks, void* output, size_t output_length, ccv_nnc_stream_context_t* stream_context)
{
	return nnc::depalettize(input, datatype, input_length, qbits, number_in_blocks, output, output_length, stream_context);
}
extern "C" size_t nnc_mi355x_palettized_bytes(int datatype, size_t count, int qbits, int number_in_blocks)
{
	return nnc::palettized_bytes(datatype, count, qbits, number_in_blocks);
}
