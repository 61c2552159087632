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
// This is HBM-bound byte work -- q / 8 bytes read and 2 / 4 / 8 bytes written per element, plus a palette per block -- so the kernel is about the store side:
//   * a TASK is a 16-byte share of a group of eight consecutive elements (8 halves, 4 floats, 2 doubles): ONE 16-byte store per lane, consecutive lanes
//     consecutive 16 bytes whatever the element size; the group's index bytes arrive in ONE 8-byte load (gfx950 global loads need no alignment; the lanes
//     that share a group load the same bytes), get byte-swapped into a 64-bit big-endian window and cost a constant shift + mask per element;
//   * a lane has four tasks' index loads in flight before it looks anything up;
//   * palettes are staged in LDS: a workgroup expands 1024 tasks -- a chunk of one large block, or a run of consecutive small blocks with their palettes side
//     by side (the reference's tests use 128-element blocks: up to 64 of them per workgroup), so no workgroup idles on a 16-group block and no look-up goes
//     to the vector cache.
// Measured on the MI355X (tools/palette_bench.py; profiles/r05_v9_palette_bench.txt = the first form of this file, r05_v10_palette_bench.txt = this one), 235 M
// elements: 4.3 - 5.6 TB/s of algorithmic traffic = 0.54 - 0.70 of the 8 TB/s peak (the guide's achievable figure is 6.3).  On the way: byte-by-byte index loads
// with a 64-bit shift per byte + one block per workgroup or the palette through the vector cache for small blocks 2.1 - 3.3 TB/s; the 8-byte load alone 3.1 - 5.7
// (small blocks still slow); runs of blocks per workgroup 3.8 - 5.6 (floats low: a lane wrote 32 bytes as two stores 32 bytes apart); 16-byte shares: this.
// A command that meets a CCV_QX input runs on a dense image of it: depalettized_exec() below.
#include "common.h"

namespace nnc {
namespace {

constexpr int PAL_THREADS = 256;
constexpr int PAL_GROUPS_PER_LANE = 4; // tasks a lane takes (PAL_THREADS apart): four independent index loads in flight
constexpr int PAL_CHUNK = PAL_THREADS * PAL_GROUPS_PER_LANE; // 16-byte shares of groups (tasks) per workgroup
constexpr int PAL_LDS_BYTES = 16384;   // palettes a workgroup may stage: 8 runs of blocks per CU stay resident

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

// One group: up to eight elements starting at element 8 g of a block.  `idx` points at the group's first index byte, `valid` (1 .. 8) elements exist.
// Fast path (a full group whose eight bytes idx[0 .. 7] all lie inside the stream -- every group but the stream's last one or two): ONE 8-byte load (gfx950
// global loads need no alignment), a byte swap, and a constant shift + mask per element.  Otherwise byte by byte, and only the index bytes the group's
// elements touch are read (the stream of the tensor's last block ends with them).
template <int Q>
__device__ __forceinline__ unsigned long long group_bits(const unsigned char* const idx, const int valid, const bool whole)
{ // the group's Q bytes as one big-endian number in the TOP 8 Q bits of the result: element j = bits [64 - (j + 1) Q, 64 - j Q)
	if (whole) {
		unsigned long long raw;
		__builtin_memcpy(&raw, idx, 8);
		return __builtin_bswap64(raw);
	}
	const int nbytes = valid == 8 ? Q : (valid * Q + 7) >> 3;
	unsigned long long w = 0;
#pragma unroll
	for (int i = 0; i < Q; i++) {
		const unsigned long long b = i < nbytes ? idx[i] : 0;
		w |= b << (8 * (7 - i));
	}
	return w;
}
// A lane's share of a group: EPL = 16 / sizeof(ELEM) consecutive elements (8 halves, 4 floats, 2 doubles) -- ONE 16-byte store per lane, consecutive lanes
// consecutive 16 bytes, whatever the element size.  `w` holds the lane's first element in its top Q bits, `valid` counts the lane's elements that exist.
template <typename ELEM, int Q, typename LUT>
__device__ __forceinline__ void expand_share(const unsigned long long w, const int valid, const LUT palette, ELEM* const out, const bool vec)
{
	constexpr int EPL = 16 / (int)sizeof(ELEM);
	ELEM v[EPL];
#pragma unroll
	for (int j = 0; j < EPL; j++) v[j] = palette[(unsigned)(w >> (64 - Q * (j + 1))) & ((1u << Q) - 1)];
	if (valid == EPL && vec) {
		typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
		union { ELEM e[EPL]; u32x4 q; } pack;
#pragma unroll
		for (int j = 0; j < EPL; j++) pack.e[j] = v[j];
		*(u32x4*)out = pack.q;
	} else
#pragma unroll
		for (int j = 0; j < EPL; j++) if (j < valid) out[j] = v[j];
}

struct pal_geom_t {
	size_t count;        // elements of the tensor
	size_t in_len;       // bytes of the palettized stream
	size_t block_stride; // bytes from one block to the next in the palettized stream: palette + a full block's index bytes
	size_t blocks;
	int nib;             // elements per block
	int gpb;             // groups of eight per block (the last one partial when nib is not a multiple of 8)
	int bpw;             // consecutive blocks a workgroup expands (their palettes side by side in LDS); 1 when a block has PAL_CHUNK tasks or more
	int vec;             // out is 16-byte aligned
	unsigned div_m;      // L / gpb for the workgroup's local group index L < 2^31 as (L * div_m) >> div_sh (round-up reciprocal)
	int div_sh;
};

// blockIdx.x = a run of g.bpw consecutive blocks, blockIdx.y = chunk of PAL_CHUNK tasks (more than one chunk only when bpw == 1).  The palettes of the run
// are staged in LDS once; a lane then takes PAL_GROUPS_PER_LANE tasks, PAL_THREADS apart -- all its index loads first, then the look-ups and the stores.
template <typename ELEM, int Q>
__global__ void __launch_bounds__(PAL_THREADS) depalettize_kernel(const unsigned char* const in, ELEM* const out, const pal_geom_t g)
{
	__shared__ __attribute__((aligned(16))) unsigned char lds[PAL_LDS_BYTES];
	ELEM* const palette = (ELEM*)lds; // [bpw][1 << Q]
	const size_t block0 = (size_t)blockIdx.x * (size_t)g.bpw;
	const int nblk = g.blocks - block0 < (size_t)g.bpw ? (int)(g.blocks - block0) : g.bpw;
	for (int i = threadIdx.x; i < nblk << Q; i += PAL_THREADS) { // (a block's stride may be odd -- 5 index bytes for eight 5-bit elements --: the palettes are not aligned to their words)
		ELEM v;
		__builtin_memcpy(&v, in + (block0 + (size_t)(i >> Q)) * g.block_stride + sizeof(ELEM) * (size_t)(i & ((1 << Q) - 1)), sizeof(ELEM));
		palette[i] = v;
	}
	__syncthreads();
	constexpr int EPL = 16 / (int)sizeof(ELEM), LPG = 8 / EPL; // elements per lane, lanes per group (1 / 2 / 4)
	const int t0 = blockIdx.y * PAL_CHUNK;
	unsigned long long w[PAL_GROUPS_PER_LANE];
	int valid[PAL_GROUPS_PER_LANE], slot[PAL_GROUPS_PER_LANE];
	size_t first[PAL_GROUPS_PER_LANE]; // the lane's first element in the tensor
#pragma unroll
	for (int k = 0; k < PAL_GROUPS_PER_LANE; k++) {
		const int t = t0 + k * PAL_THREADS + (int)threadIdx.x; // the task's index within the run: group t / LPG, share t % LPG
		const int l = t / LPG, sub = t - l * LPG;
		const int b = g.bpw == 1 ? 0 : (int)(((unsigned long long)(unsigned)l * g.div_m) >> g.div_sh);
		const int gi = l - b * g.gpb;
		valid[k] = 0; w[k] = 0; slot[k] = b << Q; first[k] = 0;
		if (b < nblk) {
			const size_t block = block0 + (size_t)b, e0 = block * (size_t)g.nib, left = g.count - e0;
			const int n = left < (size_t)g.nib ? (int)left : g.nib; // elements of this block
			const int in_group = n - gi * 8 < 8 ? n - gi * 8 : 8;   // elements of the group (<= 0: past the block's end)
			const int mine = in_group - sub * EPL;
			if (mine > 0) {
				valid[k] = mine < EPL ? mine : EPL;
				first[k] = e0 + (size_t)gi * 8 + sub * EPL;
				const size_t off = block * g.block_stride + sizeof(ELEM) * (1 << Q) + (size_t)gi * Q;
				w[k] = group_bits<Q>(in + off, in_group, in_group == 8 && off + 8 <= g.in_len) << (Q * EPL * sub); // (the lanes of a group load the same bytes: one request)
			}
		}
	}
#pragma unroll
	for (int k = 0; k < PAL_GROUPS_PER_LANE; k++)
		if (valid[k]) expand_share<ELEM, Q>(w[k], valid[k], palette + slot[k], out + first[k], g.vec && (g.nib & 7) == 0);
}

template <typename ELEM, int Q>
int depalettize_launch(const unsigned char* const in, const size_t in_len, ELEM* const out, const size_t count, const int nib, hipStream_t stream)
{
	pal_geom_t g;
	g.count = count;
	g.in_len = in_len;
	g.nib = nib;
	g.gpb = (nib + 7) / 8;
	g.block_stride = sizeof(ELEM) * ((size_t)1 << Q) + index_bytes_per_block(Q, nib);
	g.vec = (((uintptr_t)out) & 15) == 0;
	g.blocks = (count + (size_t)nib - 1) / (size_t)nib;
	// small blocks share a workgroup: as many as give every lane its PAL_GROUPS_PER_LANE groups, within the LDS the palettes may take
	constexpr int LPG = (int)sizeof(ELEM) / 2; // lanes per group (a lane stores 16 bytes)
	long bpw = PAL_CHUNK / ((long)g.gpb * LPG);
	const long fit = PAL_LDS_BYTES / (long)(sizeof(ELEM) << Q);
	if (bpw > fit) bpw = fit;
	if (bpw < 1) bpw = 1;
	g.bpw = (int)bpw;
	int sh = 0;
	while ((1ll << sh) < g.gpb) sh++;
	g.div_sh = 31 + sh; // the reciprocal of gpb: m = ceil(2^(31 + s) / gpb), 2^s >= gpb; exact for the 31-bit numerators a run's local index has
	g.div_m = (unsigned)(((1ull << g.div_sh) + (unsigned long long)g.gpb - 1) / (unsigned long long)g.gpb);
	const size_t runs = (g.blocks + (size_t)g.bpw - 1) / (size_t)g.bpw;
	const size_t chunks = g.bpw == 1 ? ((size_t)g.gpb * LPG + PAL_CHUNK - 1) / PAL_CHUNK : 1;
	if (runs > 0x7fffffffu || chunks > 65535u) return CCV_NNC_EXEC_INVALID; // (2^31 runs of >= 1024 groups: no tensor is that large)
	hipLaunchKernelGGL((depalettize_kernel<ELEM, Q>), dim3((unsigned)runs, (unsigned)chunks), dim3(PAL_THREADS), 0, stream, in, out, g);
	HIP_ENFORCE(hipGetLastError());
	note_kernel("depalettize");
	return CCV_NNC_EXEC_SUCCESS;
}

template <typename ELEM>
int depalettize_bits(const unsigned char* const in, const size_t in_len, ELEM* const out, const size_t count, const int qbits, const int nib, hipStream_t stream)
{
	switch (qbits) {
		case 4: return depalettize_launch<ELEM, 4>(in, in_len, out, count, nib, stream);
		case 5: return depalettize_launch<ELEM, 5>(in, in_len, out, count, nib, stream);
		case 6: return depalettize_launch<ELEM, 6>(in, in_len, out, count, nib, stream);
		case 7: return depalettize_launch<ELEM, 7>(in, in_len, out, count, nib, stream);
		case 8: return depalettize_launch<ELEM, 8>(in, in_len, out, count, nib, stream);
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
		case CCV_16F: return depalettize_bits<unsigned short>(in, input_length, (unsigned short*)output, output_length, qbits, nib, stream);
		case CCV_32F: return depalettize_bits<unsigned int>(in, input_length, (unsigned int*)output, output_length, qbits, nib, stream);
		case CCV_64F: return depalettize_bits<unsigned long long>(in, input_length, (unsigned long long*)output, output_length, qbits, nib, stream);
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
