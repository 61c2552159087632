// Host-side launcher for the MFMA contraction template: tile grid, split-K policy, workspace, reduce pass.
#pragma once
#include "common.h"
#include "mfma_gemm.h"
#include "mfma_gemm_f16_buf.h"
#include "mfma_gemm_bf16x3.h"

namespace nnc {

// The grid of a batched contraction without split-K and the kernel's `splits` argument that goes with it (mfma_gemm.h gemm_batch_xcd_map): one dimension over
// (entry, tile) with every batch entry on ONE XCD, or -- a batch of one, TUNE_GEMM_BATCH_XCD = 0, a grid beyond 2^31 -- the entries on grid z.
static inline dim3 gemm_batch_grid(const long tiles, const int zcount, int* const splits_arg)
{
	const long flat = (long)((zcount + 7) / 8) * 8 * tiles;
	// (few entries would leave XCDs without work: 64 or more, or whole rounds of eight)
	if ((zcount >= 64 || (zcount >= 16 && zcount % 8 == 0)) && tune(TUNE_GEMM_BATCH_XCD) > 0 && flat <= 0x7fffffffL) { *splits_arg = -zcount; return dim3((unsigned)flat, 1, 1); }
	*splits_arg = 1;
	return dim3((unsigned)tiles, 1, (unsigned)zcount);
}


struct GemmOut {
	float* c;
	long ldm, ldn;
	const float* bias; // per output column n (may be null)
	float alpha;
	int accumulate; // c += result (CCV_NNC_ACCUMULATE_OUTPUT)
	long bias_ldm;  // row stride of bias (0 = a single row broadcast over the output rows)
	long bias_ldn = 1; // column stride of bias (bias_ldm = 1, bias_ldn = 0: one value per output row)
};

// May the epilogue write four consecutive output columns of a row in one access (mfma_gemm.h "epilogues": the block tile is read back row-major out of LDS)?
// n-contiguous output, row stride / column count / batch offset multiples of 4 elements, the base aligned to 4 elements.  TUNE_GEMM_VEC_EPILOGUE = 0 turns it off.
static inline int epi_vec_ok(const void* const c, const size_t es, const long ldm, const long ldn, const int N, const int zcount, const long c_z)
{
	return tune(TUNE_GEMM_VEC_EPILOGUE) && ldn == 1 && ldm % 4 == 0 && N % 4 == 0 && ((uintptr_t)c & (4 * es - 1)) == 0 && (zcount <= 1 || c_z % 4 == 0);
}

// Split the reduction so that a contraction with few output tiles still fills 256 CUs (2 workgroups per CU fit by
// LDS).  Every slice keeps >= 8 K-steps; slices are multiples of BK so only the last one is ragged.
static inline int gemm_auto_splits(long tiles, int K)
{
	const long target = (long)device_cu_count() * 3;
	if (tiles >= target) return 1;
	long s = (target + tiles - 1) / tiles;
	const long max_s = K / (GEMM_BK * 8);
	if (s > max_s) s = max_s;
	if (s > 512) s = 512;
	return s < 1 ? 1 : (int)s;
}

// Upper bound of the split-K scratch gemm_run() requests for an M x N x K contraction (any tile shape it may pick).
inline size_t gemm_workspace_bound(long M, long N, long K)
{
	size_t worst = 0;
	static const int shapes[5][2] = { { 2, 2 }, { 2, 1 }, { 1, 2 }, { 1, 1 }, { 4, 4 } }; // (4, 4): the 256 x 256 tile of mfma_gemm_bf16x3.h
	for (int i = 0; i < 5; i++) {
		const long tiles = ((M + 64 * shapes[i][0] - 1) / (64 * shapes[i][0])) * ((N + 64 * shapes[i][1] - 1) / (64 * shapes[i][1]));
		int s = gemm_auto_splits(tiles, (int)(K > 0x7fffffffL ? 0x7fffffffL : K));
		if (s > 1) s = ((s < 8 ? 8 : s) + 7) & ~7;
		const size_t b = s > 1 ? sizeof(float) * (size_t)M * N * s : 0;
		if (b > worst) worst = b;
	}
	return worst;
}

// Pick the block tile: 128x128 unless an output dimension would be mostly padding (64-channel layers), where the
// narrower tile keeps the MFMA pipe on useful work.  Score = useful / issued work x a mild preference for big tiles.
static inline void gemm_pick_tile(const int M, const int N, const int K, const int zcount, int* wm, int* wn)
{
	static const int shapes[4][2] = { { 2, 2 }, { 2, 1 }, { 1, 2 }, { 1, 1 } };
	static const double eff[4] = { 1.0, 0.93, 0.93, 0.8 }; // 64 x 64: both outputs 64 channels (the Winograd filter gradient of conv1_2)
	if (g_force_tile) { *wm = g_force_tile & 0xff; *wn = g_force_tile >> 8; return; }
	// The per-image products of a 1 x 1 convolution on NCHW tensors (a batch of >= 64 small matrices): with at most 8 K-steps a workgroup is mostly its
	// prologue and its epilogue, and what hides those is MORE workgroups per CU -- four 64 x 64 tiles fit by LDS where two 128 x 128 do.  Likewise a filter
	// gradient whose whole output is a few tiles (it splits K either way).  Measured on ResNet-50's layers at batch 256 (tools/conv1x1_bench.py,
	// profiles/r04_v4_conv1x1_bench_f32.txt): 64 -> 256 at 56^2 forward 0.413 -> 0.375 ms, its filter gradient 0.485 -> 0.419, 256 -> 128 forward 0.472 -> 0.430;
	// with K >= 512 the big tile is as good or better.
	if ((zcount >= 64 && K <= 256 && (long)M * N <= 1L << 20) || (zcount == 1 && (long)M * N <= 32768 && K >= 65536)) { *wm = 1; *wn = 1; return; }
	// 256 x 128 / 128 x 256 tiles (WM, WN = 4, 2 / 2, 4: one workgroup per CU) were measured on the whole VGG-D step: 4x2
	// 108 vs 127 TFLOP/s and 2x4 121 vs 127 for the 128 x 128 tile on the same layers (profiles/r01_v6_bigtile_bench.json),
	// although an isolated probe of one layer had them 4 % ahead; the kernel template still supports them (tools/kprobe.cpp).
	double best = -1;
	for (int i = 0; i < 4; i++) {
		const long bm = 64 * shapes[i][0], bn = 64 * shapes[i][1];
		const double padded = (double)((M + bm - 1) / bm * bm) * (double)((N + bn - 1) / bn * bn);
		const double score = (double)M * N / padded * eff[i];
		if (score > best + 1e-9) { best = score; *wm = shapes[i][0]; *wn = shapes[i][1]; }
	}
}

// The buffer-load form of a plain-matrix operand (BufMatLoader, mfma_gemm.h): which loaders qualify, and whether one's offsets fit.
template <class L> struct is_vec_mat_loader { static constexpr bool value = false; };
template <bool KC> struct is_vec_mat_loader<MatLoader<KC, true>> { static constexpr bool value = true; };
template <bool KC>
static inline bool buf_loader_ok(const MatLoader<KC, true>& l)
{
	if (l.R <= 0 || l.K <= 0) return false;
	if (KC) return l.ldk == 1 && l.ldr > 0 && l.ldr * 128 * 4 + (long)l.K * 4 < 0x7fffffffL;
	return l.ldr == 1 && l.ldk > 0 && l.R % 4 == 0 && (long)l.K * l.ldk * 4 + (long)l.R * 4 < 0x7fffffffL;
}

// zcount > 1 (batched GEMM / grouped conv): every z applies the given element offsets to A, B, C and bias.
template <class LA, class LB, int WM, int WN>
static int gemm_run_tile(const char* name, const LA& la, const LB& lb, const GemmOut out, const int M, const int N, const int K, const int zcount, const long a_z, const long b_z, const long c_z, const long bias_z, int splits, const int flags, ccv_nnc_stream_context_t* const ctx, const KOrder ko)
{
	constexpr int BM = 64 * WM, BN = 64 * WN;
	hipStream_t stream = stream_of(ctx);
	const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
	const long tiles = (long)tiles_m * tiles_n;
	if (tiles > 0x7fffffffL) return CCV_NNC_EXEC_INVALID;
	// the automatic choice never splits a batched contraction (callers size their workspace for one slab set); an
	// explicit splits > 1 with zcount > 1 is the batched split-K of the Winograd filter gradient: z slab sets back to back
	if (splits <= 0) splits = (zcount == 1 && !(flags & CCV_NNC_ZERO_MEMORY_ALLOC)) ? gemm_auto_splits(tiles, K) : 1;
	int k_per_split = K;
	if (splits > 1) {
		// whole K-slices per XCD (see the kernel's block -> (slice, tile) map): the slice count is a multiple of 8;
		// trailing slices may be empty (they write zero slabs), never more than 7 of them
		if (splits < 8) splits = 8;
		splits = (splits + 7) & ~7;
		k_per_split = ((K + splits - 1) / splits + GEMM_BK - 1) / GEMM_BK * GEMM_BK;
	}
	note_kernel(name);
	if (K <= 0) splits = 1;
	// Kernel symbol as rocprofv3 prints it: nnc::mfma_gemm_f32_kernel<LA, LB, EpiStore|EpiPartial, WM, WN>; __PRETTY_FUNCTION__ carries LA / LB / WM / WN.
	char prof_name[192];
	snprintf(prof_name, sizeof(prof_name), "%s|%s EPI = %s", name, __PRETTY_FUNCTION__ + (sizeof(__PRETTY_FUNCTION__) > 110 ? sizeof(__PRETTY_FUNCTION__) - 110 : 0), splits <= 1 ? "EpiStore" : "EpiPartial");
	const double flops = 2.0 * (double)M * (double)N * (double)K * (double)zcount;
	if (splits <= 1) {
		EpiStore epi;
		epi.c = out.c; epi.ldm = out.ldm; epi.ldn = out.ldn; epi.bias = out.bias; epi.alpha = out.alpha; epi.accumulate = out.accumulate; epi.M = M; epi.N = N; epi.bias_ldm = out.bias_ldm; epi.bias_ldn = out.bias_ldn;
		epi.vec = epi_vec_ok(out.c, sizeof(float), out.ldm, out.ldn, N, zcount, c_z);
		ProfScope prof(prof_name, flops, 0, M, N, K, zcount, 1, stream);
		int sarg;
		const dim3 grid = gemm_batch_grid(tiles, zcount, &sarg);
		hipLaunchKernelGGL(HIP_KERNEL_NAME(mfma_gemm_f32_kernel<LA, LB, EpiStore, WM, WN>), grid, dim3(GEMM_THREADS), 0, stream, la, lb, epi, tiles_m, tiles_n, K, K > 0 ? K : 1, sarg, a_z, b_z, c_z, bias_z, ko);
		HIP_ENFORCE(hipGetLastError());
		return CCV_NNC_EXEC_SUCCESS;
	}
	const long slab = (long)M * N;
	float* ws = (float*)workspace_of(ctx, sizeof(float) * (size_t)slab * splits * zcount);
	if (!ws) return CCV_NNC_EXEC_OOM;
	EpiPartial epi;
	epi.c = ws; epi.bias = 0; epi.slab = slab; epi.M = M; epi.N = N;
	epi.vec = tune(TUNE_GEMM_VEC_EPILOGUE) && N % 4 == 0; // (the slabs start at the 256-byte aligned workspace and are M * N floats each)
	{
		ProfScope prof(prof_name, flops, 0, M, N, K, zcount, splits, stream);
		hipLaunchKernelGGL(HIP_KERNEL_NAME(mfma_gemm_f32_kernel<LA, LB, EpiPartial, WM, WN>), dim3((unsigned)(tiles * splits), 1, (unsigned)zcount), dim3(GEMM_THREADS), 0, stream, la, lb, epi, tiles_m, tiles_n, K, k_per_split, splits, a_z, b_z, slab * splits, 0L, ko);
	}
	HIP_ENFORCE(hipGetLastError());
	hipLaunchKernelGGL(HIP_KERNEL_NAME(splitk_reduce_kernel<float>), dim3(grid_for((size_t)slab * 4, 256), (unsigned)zcount), dim3(256), 0, stream, (const float*)ws, splits, slab, out.c, out.ldm, out.ldn, out.bias, out.bias_ldm, out.alpha, out.accumulate, M, N, c_z, bias_z, out.bias_ldn);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}

// The same contraction on the bf16 matrix pipe with exactly split operands (mfma_gemm_bf16x3.h): tile grid, split-K policy, workspace and reduce pass as above.
template <bool AKC, bool BKC, int TM, int TN, int WM, int WN>
static int gemm_run_bf16x3_tile(const char* name, const BufMatLoader<AKC>& la, const BufMatLoader<BKC>& lb, const GemmOut out, const int M, const int N, const int K, const int zcount, const long a_z, const long b_z, const long c_z, const long bias_z, int splits, const int flags, ccv_nnc_stream_context_t* const ctx)
{
	constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, NT = 64 * WM * WN;
	hipStream_t stream = stream_of(ctx);
	const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
	const long tiles = (long)tiles_m * tiles_n;
	if (tiles > 0x7fffffffL) return CCV_NNC_EXEC_INVALID;
	if (splits <= 0) splits = (zcount == 1 && !(flags & CCV_NNC_ZERO_MEMORY_ALLOC)) ? gemm_auto_splits(tiles, K) : 1;
	int k_per_split = K;
	if (splits > 1) {
		if (splits < 8) splits = 8;
		splits = (splits + 7) & ~7;
		k_per_split = ((K + splits - 1) / splits + GEMM_BK - 1) / GEMM_BK * GEMM_BK;
	}
	note_kernel(name);
	char prof_name[192];
	snprintf(prof_name, sizeof(prof_name), "%s|bf16x3 %dx%d %s%s EPI = %s", name, BM, BN, AKC ? "k" : "r", BKC ? "k" : "r", splits <= 1 ? "EpiStore" : "EpiPartial");
	const double flops = 2.0 * (double)M * (double)N * (double)K * (double)zcount;
	if (splits <= 1) {
		EpiStore epi;
		epi.c = out.c; epi.ldm = out.ldm; epi.ldn = out.ldn; epi.bias = out.bias; epi.alpha = out.alpha; epi.accumulate = out.accumulate; epi.M = M; epi.N = N; epi.bias_ldm = out.bias_ldm; epi.bias_ldn = out.bias_ldn;
		epi.vec = epi_vec_ok(out.c, sizeof(float), out.ldm, out.ldn, N, zcount, c_z);
		ProfScope prof(prof_name, flops, 0, M, N, K, zcount, 1, stream);
		int sarg;
		const dim3 grid = gemm_batch_grid(tiles, zcount, &sarg);
		hipLaunchKernelGGL(HIP_KERNEL_NAME(mfma_gemm_bf16x3_kernel<AKC, BKC, EpiStore, TM, TN, WM, WN>), grid, dim3(NT), 0, stream, la, lb, epi, tiles_m, tiles_n, K, K, sarg, a_z, b_z, c_z, bias_z);
		HIP_ENFORCE(hipGetLastError());
		return CCV_NNC_EXEC_SUCCESS;
	}
	const long slab = (long)M * N;
	float* ws = (float*)workspace_of(ctx, sizeof(float) * (size_t)slab * splits * zcount);
	if (!ws) return CCV_NNC_EXEC_OOM;
	EpiPartial epi;
	epi.c = ws; epi.bias = 0; epi.slab = slab; epi.M = M; epi.N = N;
	epi.vec = tune(TUNE_GEMM_VEC_EPILOGUE) && N % 4 == 0;
	{
		ProfScope prof(prof_name, flops, 0, M, N, K, zcount, splits, stream);
		hipLaunchKernelGGL(HIP_KERNEL_NAME(mfma_gemm_bf16x3_kernel<AKC, BKC, EpiPartial, TM, TN, WM, WN>), dim3((unsigned)(tiles * splits), 1, (unsigned)zcount), dim3(NT), 0, stream, la, lb, epi, tiles_m, tiles_n, K, k_per_split, splits, a_z, b_z, slab * splits, 0L);
	}
	HIP_ENFORCE(hipGetLastError());
	hipLaunchKernelGGL(HIP_KERNEL_NAME(splitk_reduce_kernel<float>), dim3(grid_for((size_t)slab * 4, 256), (unsigned)zcount), dim3(256), 0, stream, (const float*)ws, splits, slab, out.c, out.ldm, out.ldn, out.bias, out.bias_ldm, out.alpha, out.accumulate, M, N, c_z, bias_z, out.bias_ldn);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}
// Where the split form is taken (TUNE_GEMM_BF16X3 = 1; measured on the MI355X, tools/bf16x3_bench.py, profiles/r06_v2_bf16x3_bench.txt): only where its 256 x 256
// tile fills the chip -- at 128 x 128 it merely equals the fp32 instructions (0.95 - 1.04 x) --, for two k-contiguous operands (the Winograd-domain forward /
// data-gradient products 1.10 - 1.19 x, the fc layers' forward products 1.10 - 1.14 x) and for two row-contiguous ones with both outputs >= 512 (the filter
// gradients of the 512-channel layers 1.05 x; at 256 x 256 / 512 x 256 outputs the split form loses 3 - 6 %).
// 0: the fp32 matrix instructions; 128 / 256: the split form with that block tile.
template <bool AKC, bool BKC>
static inline int gemm_bf16x3_tile(const int M, const int N, const int K, const int zcount, const int splits, const int flags)
{
	const long mode = tune(TUNE_GEMM_BF16X3);
	if (mode <= 0 || K % 16) return 0;
	const long t256 = (long)((M + 255) / 256) * ((N + 255) / 256), t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
	int s = splits;
	if (s <= 0) s = (zcount == 1 && !(flags & CCV_NNC_ZERO_MEMORY_ALLOC)) ? gemm_auto_splits(t256, K) : 1;
	if (s < 1) s = 1;
	const long cus = (long)device_cu_count();
	if (mode == 3) return 128;
	if (mode == 4) return 256;
	if (mode == 2) return (M >= 256 && N >= 256 && t256 * zcount * s >= cus) ? 256 : 128;
	// Batched products without split-K whose B operand is row-contiguous -- the 1 x 1 convolutions on NCHW tensors, one product per image (forward: w . planes,
	// data gradient: w^T . planes) -- with 256 or more of K and 128 or more output rows: 1.09 - 1.39 x on ResNet-50's layers at batch 256 (1024 -> 512 at 14^2
	// 0.519 -> 0.381 ms, 256 -> 1024 data gradient 0.259 -> 0.191; with K = 64 / 128 or 64 rows the split form loses: profiles/r06_v14_conv1x1_f32_bf16x3.txt);
	// the 128 x 128 tile (N = 196 / 784 pixels: 0.371 against 0.407 ms for 512 -> 256 at 28^2).
	if (!BKC && zcount >= 8 && splits == 1) return (K >= 256 && M >= 128 && N >= 128) ? 128 : 0;
	// One product of a k-contiguous and a row-contiguous operand (the data gradient of a 1 x 1 convolution over NHWC images, conv_pointwise in cmd_conv.cpp):
	// the larger tile that fills the chip
	if (AKC != BKC) {
		if (K < 256 || M < 256 || N < 256) return 0;
		// in K-slices (the fc layers' data gradients, 256 x 18432 x 4096 in 8 slices: 0.421 -> 0.335 ms with the 128 x 128 tile, 0.363 with 256 x 256;
		// profiles/r06_v18_bf16x3_bench_tile128.txt)
		if (s > 1) return K / s >= 256 ? 128 : 0;
		return t256 * zcount >= cus ? 256 : (t128 * zcount >= cus ? 128 : 0);
	}
	if (!AKC && K >= 128 && M >= 128 && N >= 128 && t128 * zcount * s >= cus) return 128; // (see below)
	if (K < 128 || M < 256 || N < 256) return 0;
	// Two row-contiguous operands (filter gradients): the 128 x 128 tile.  With 256 x 256 a short reduction was all epilogue (256 KB per workgroup: fc7's filter gradient,
	// 4096 x 4096 x 256, 0.242 ms against 0.189 on the fp32 instructions in round 6's first half) and 256 x 256 outputs lost 3 - 8 %; with 128 x 128 the same product runs
	// 0.145 against 0.170 ms, the Winograd filter gradient of a 256-channel layer (256 x 256 x 50176 x 36 in 16 slices) 1.656 against 1.770, of the 512-channel layers
	// 1.629 / 0.596 against 1.683 / 0.639 (256 x 256 tile: 1.647 / 0.606), 512 x 256 outputs the same as the fp32 instructions (same file)
	if (!AKC) return t128 * zcount * s >= cus ? 128 : 0;
	return t256 * zcount * s >= cus ? 256 : 0;
}
template <bool AKC, bool BKC>
static int gemm_run_bf16x3(const int tile, const char* name, const BufMatLoader<AKC>& la, const BufMatLoader<BKC>& lb, const GemmOut out, const int M, const int N, const int K, const int zcount, const long a_z, const long b_z, const long c_z, const long bias_z, const int splits, const int flags, ccv_nnc_stream_context_t* const ctx)
{
	if (tile == 256) return gemm_run_bf16x3_tile<AKC, BKC, 4, 2, 2, 4>(name, la, lb, out, M, N, K, zcount, a_z, b_z, c_z, bias_z, splits, flags, ctx);
	return gemm_run_bf16x3_tile<AKC, BKC, 2, 2, 2, 2>(name, la, lb, out, M, N, K, zcount, a_z, b_z, c_z, bias_z, splits, flags, ctx);
}

template <class LA, class LB>
static int gemm_run(const char* name, LA la, LB lb, const GemmOut out, const int M, const int N, const int K, const int zcount, const long a_z, const long b_z, const long c_z, const long bias_z, int splits, const int flags, ccv_nnc_stream_context_t* const ctx, const KOrder ko = KOrder())
{
	if (M <= 0 || N <= 0) return CCV_NNC_EXEC_SUCCESS;
	la.finish();
	lb.finish();
	// masked lanes read this device's page of zeros, addressed relative to each operand's own base (mfma_gemm.h)
	const float* zp = zero_page_of(ctx);
	la.zoff = zp - la.p;
	lb.zoff = zp - lb.p;
	int wm = 2, wn = 2;
	gemm_pick_tile(M, N, K, zcount, &wm, &wn);
	// two plain matrices in 16-byte chunks, whole K-steps: the buffer-load form (no address VALU in the K loop, mfma_gemm.h)
	if constexpr (is_vec_mat_loader<LA>::value && is_vec_mat_loader<LB>::value) {
		if (tune(TUNE_GEMM_BUFFER_LOADS) && K > 0 && K % GEMM_BK == 0 && buf_loader_ok(la) && buf_loader_ok(lb)) {
			BufMatLoader<LA::KCONTIG> ba; BufMatLoader<LB::KCONTIG> bb;
			ba.p = la.p; ba.zoff = 0; ba.ldr = la.ldr; ba.ldk = la.ldk; ba.R = la.R; ba.K = la.K;
			bb.p = lb.p; bb.zoff = 0; bb.ldr = lb.ldr; bb.ldk = lb.ldk; bb.R = lb.R; bb.K = lb.K;
			typedef BufMatLoader<LA::KCONTIG> BA; typedef BufMatLoader<LB::KCONTIG> BB;
			const int split_tile = g_force_tile ? 0 : gemm_bf16x3_tile<LA::KCONTIG, LB::KCONTIG>(M, N, K, zcount, splits, flags);
			if (split_tile) return gemm_run_bf16x3<LA::KCONTIG, LB::KCONTIG>(split_tile, name, ba, bb, out, M, N, K, zcount, a_z, b_z, c_z, bias_z, splits, flags, ctx);
			if (wm == 2 && wn == 2) return gemm_run_tile<BA, BB, 2, 2>(name, ba, bb, out, M, N, K, zcount, a_z, b_z, c_z, bias_z, splits, flags, ctx, ko);
			if (wm == 2) return gemm_run_tile<BA, BB, 2, 1>(name, ba, bb, out, M, N, K, zcount, a_z, b_z, c_z, bias_z, splits, flags, ctx, ko);
			if (wn == 1) return gemm_run_tile<BA, BB, 1, 1>(name, ba, bb, out, M, N, K, zcount, a_z, b_z, c_z, bias_z, splits, flags, ctx, ko);
			return gemm_run_tile<BA, BB, 1, 2>(name, ba, bb, out, M, N, K, zcount, a_z, b_z, c_z, bias_z, splits, flags, ctx, ko);
		}
	}
	if (wm == 2 && wn == 2) return gemm_run_tile<LA, LB, 2, 2>(name, la, lb, out, M, N, K, zcount, a_z, b_z, c_z, bias_z, splits, flags, ctx, ko);
	if (wm == 2) return gemm_run_tile<LA, LB, 2, 1>(name, la, lb, out, M, N, K, zcount, a_z, b_z, c_z, bias_z, splits, flags, ctx, ko);
	if (wn == 1) return gemm_run_tile<LA, LB, 1, 1>(name, la, lb, out, M, N, K, zcount, a_z, b_z, c_z, bias_z, splits, flags, ctx, ko);
	return gemm_run_tile<LA, LB, 1, 2>(name, la, lb, out, M, N, K, zcount, a_z, b_z, c_z, bias_z, splits, flags, ctx, ko);
}

// ---- half precision (mfma_gemm_f16.h): same policy (tile grid, split-K, workspace, reduce pass), CCV_16F operands and result ----
struct GemmOutH {
	half_t* c;
	long ldm, ldn;
	const half_t* bias;
	float alpha;
	int accumulate;
	long bias_ldm;
	long bias_ldn = 1;
};

template <class LA, class LB, int WM, int WN>
static int gemm_run_tile_h(const char* name, const LA& la, const LB& lb, const GemmOutH out, const int M, const int N, const int K, const int zcount, const long a_z, const long b_z, const long c_z, const long bias_z, int splits, const int flags, ccv_nnc_stream_context_t* const ctx, const KOrder ko)
{
	constexpr int BM = 64 * WM, BN = 64 * WN;
	hipStream_t stream = stream_of(ctx);
	const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
	const long tiles = (long)tiles_m * tiles_n;
	if (tiles > 0x7fffffffL) return CCV_NNC_EXEC_INVALID;
	if (splits <= 0) splits = (zcount == 1 && !(flags & CCV_NNC_ZERO_MEMORY_ALLOC)) ? (g_force_splits ? g_force_splits : gemm_auto_splits(tiles, K)) : 1;
	int k_per_split = K;
	if (splits > 1) {
		if (splits < 8) splits = 8;
		splits = (splits + 7) & ~7;
		k_per_split = ((K + splits - 1) / splits + GEMM_BK - 1) / GEMM_BK * GEMM_BK;
	}
	note_kernel(name);
	if (K <= 0) splits = 1;
	char prof_name[192];
	snprintf(prof_name, sizeof(prof_name), "%s|nnc::mfma_gemm_f16_kernel<%d, %d> EPI = %s", name, WM, WN, splits <= 1 ? "EpiStoreH" : "EpiPartialH");
	const double flops = 2.0 * (double)M * (double)N * (double)K * (double)zcount;
	const bool v8 = tune(TUNE_GEMM_HALF_CHUNK8) && loader_vec8_ok(la) && loader_vec8_ok(lb) && (zcount <= 1 || (a_z % 8 == 0 && b_z % 8 == 0));
	if (splits <= 1) {
		EpiStoreH epi;
		epi.c = out.c; epi.ldm = out.ldm; epi.ldn = out.ldn; epi.bias = out.bias; epi.alpha = out.alpha; epi.accumulate = out.accumulate; epi.M = M; epi.N = N; epi.bias_ldm = out.bias_ldm; epi.bias_ldn = out.bias_ldn;
		epi.vec = epi_vec_ok(out.c, sizeof(half_t), out.ldm, out.ldn, N, zcount, c_z);
		if (epi.vec && tune(TUNE_GEMM_VEC_EPILOGUE) != 3 && out.ldm % 8 == 0 && N % 8 == 0 && ((uintptr_t)out.c & 15) == 0 && (zcount <= 1 || c_z % 8 == 0)) epi.vec = 2; // 16-byte stores (TUNE_GEMM_VEC_EPILOGUE = 3: 8-byte ones, measurements)
		ProfScope prof(prof_name, flops, -2.0 * zcount * ((double)M * K + (double)N * K + (double)M * N), M, N, K, zcount, 1, stream);
		int sarg;
		const dim3 grid = gemm_batch_grid(tiles, zcount, &sarg);
		if (v8) hipLaunchKernelGGL(HIP_KERNEL_NAME(mfma_gemm_f16_kernel<LA, LB, EpiStoreH, WM, WN, 8>), grid, dim3(GEMM_THREADS), 0, stream, la, lb, epi, tiles_m, tiles_n, K, K > 0 ? K : 1, sarg, a_z, b_z, c_z, bias_z, ko);
		else hipLaunchKernelGGL(HIP_KERNEL_NAME(mfma_gemm_f16_kernel<LA, LB, EpiStoreH, WM, WN>), grid, dim3(GEMM_THREADS), 0, stream, la, lb, epi, tiles_m, tiles_n, K, K > 0 ? K : 1, sarg, a_z, b_z, c_z, bias_z, ko);
		HIP_ENFORCE(hipGetLastError());
		return CCV_NNC_EXEC_SUCCESS;
	}
	const long slab = (long)M * N;
	float* ws = (float*)workspace_of(ctx, sizeof(float) * (size_t)slab * splits * zcount);
	if (!ws) return CCV_NNC_EXEC_OOM;
	EpiPartialH epi;
	epi.c = ws; epi.bias = 0; epi.slab = slab; epi.M = M; epi.N = N;
	epi.vec = tune(TUNE_GEMM_VEC_EPILOGUE) && N % 4 == 0;
	{
		ProfScope prof(prof_name, flops, -2.0 * zcount * ((double)M * K + (double)N * K + (double)M * N), M, N, K, zcount, splits, stream);
		if (v8) hipLaunchKernelGGL(HIP_KERNEL_NAME(mfma_gemm_f16_kernel<LA, LB, EpiPartialH, WM, WN, 8>), dim3((unsigned)(tiles * splits), 1, (unsigned)zcount), dim3(GEMM_THREADS), 0, stream, la, lb, epi, tiles_m, tiles_n, K, k_per_split, splits, a_z, b_z, slab * splits, 0L, ko);
		else hipLaunchKernelGGL(HIP_KERNEL_NAME(mfma_gemm_f16_kernel<LA, LB, EpiPartialH, WM, WN>), dim3((unsigned)(tiles * splits), 1, (unsigned)zcount), dim3(GEMM_THREADS), 0, stream, la, lb, epi, tiles_m, tiles_n, K, k_per_split, splits, a_z, b_z, slab * splits, 0L, ko);
	}
	HIP_ENFORCE(hipGetLastError());
	hipLaunchKernelGGL(HIP_KERNEL_NAME(splitk_reduce_kernel<half_t>), dim3(grid_for((size_t)slab * 4, 256), (unsigned)zcount), dim3(256), 0, stream, (const float*)ws, splits, slab, out.c, out.ldm, out.ldn, out.bias, out.bias_ldm, out.alpha, out.accumulate, M, N, c_z, bias_z, out.bias_ldn);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}

// Can this loader's operand be staged in 16-byte chunks of eight halves (TileFetchH8, mfma_gemm_f16.h)?  Eight consecutive k / rows adjacent in memory and 16-byte
// aligned: channel counts and every stride a multiple of eight, an aligned base, no ragged k tail inside a chunk.
template <class L> static inline bool loader_vec8_ok(const L&) { return false; }
static inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
template <bool S, bool I> static inline bool loader_vec8_ok(const Im2colKC<true, S, I>& l) { return al16(l.p) && l.C % 8 == 0 && l.s_w % 8 == 0 && l.s_h % 8 == 0 && l.s_n % 8 == 0; }
static inline bool loader_vec8_ok(const MatLoader<true, true>& l) { return al16(l.p) && l.ldk == 1 && l.ldr % 8 == 0 && l.K % 8 == 0; }
static inline bool loader_vec8_ok(const MatLoader<false, true>& l) { return al16(l.p) && l.ldr == 1 && l.ldk % 8 == 0 && l.R % 8 == 0; }
template <bool I> static inline bool loader_vec8_ok(const WgtDgradNC<true, I>& l) { return al16(l.p) && l.C % 8 == 0 && l.ko_stride % 8 == 0; }
template <bool I> static inline bool loader_vec8_ok(const Im2colNC<true, I>& l) { return al16(l.p) && l.C % 8 == 0 && l.s_w % 8 == 0 && l.s_h % 8 == 0 && l.s_n % 8 == 0; }
static inline bool loader_vec8_ok(const PlaneKC<true>& l) { return al16(l.p) && l.P % 8 == 0 && l.ldr % 8 == 0 && l.s_n % 8 == 0; }

// A half-precision contraction whose rows are the pixels of an NHWC view, written straight into an NCHW tensor (EpiStoreHT, mfma_gemm_f16.h): one slab, no batch.
template <class LA, class LB>
static int gemm_run_h_planar(const char* name, LA la, LB lb, EpiStoreHT epi, const int K, ccv_nnc_stream_context_t* const ctx, const KOrder ko, const bool small_tiles = false)
{
	const int M = epi.M, N = epi.N;
	if (M <= 0 || N <= 0) return CCV_NNC_EXEC_SUCCESS;
	la.finish();
	lb.finish();
	epi.d_p.init(epi.P);
	epi.vec = (tune(TUNE_GEMM_VEC_EPILOGUE) != 3 && epi.P % 8 == 0 && M % 8 == 0 && (((uintptr_t)epi.c) & 15) == 0) ? 2 : 1; // eight pixels of a plane per store (16 bytes) where the planes allow
	const half_t* zp = (const half_t*)zero_page_of(ctx);
	la.zoff = zp - (const half_t*)la.p;
	lb.zoff = zp - (const half_t*)lb.p;
	hipStream_t stream = stream_of(ctx);
	note_kernel(name);
	const long big_tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
	const bool big = g_force_tile ? !((g_force_tile & 0xff) == 1 && (g_force_tile >> 8) == 1) : (!small_tiles && M > 64 && N > 64 && (big_tiles >= device_cu_count() || K >= 4096));
	const bool v8 = tune(TUNE_GEMM_HALF_CHUNK8) && loader_vec8_ok(la) && loader_vec8_ok(lb);
	// 256 x 128 block tiles, 128 x 64 per wave (round 6; nnc_mi355x_debug_force_tile(4, 2)): twice the MFMAs per K-step against 1.5 x the fragment reads and chunk
	// addresses.  Measured (tools/conv_half_bench.py, profiles/r06_v11_conv_half_bench.txt): 1 - 6 % faster on every forward / data-gradient shape of the two trainers that
	// fills the chip with such tiles (64 -> 128 at 32^2: 0.121 -> 0.118 / 0.176 -> 0.166 ms), equal on the rest -- the kernel is not bound by that ratio either.
	const long tall_tiles = (long)((M + 255) / 256) * ((N + 127) / 128);
	const bool tall = v8 && (g_force_tile ? g_force_tile == (4 | (2 << 8)) : (big && tall_tiles >= 2L * device_cu_count()));
	char prof_name[192];
	snprintf(prof_name, sizeof(prof_name), "%s|nnc::mfma_gemm_f16_kernel<%d, %d> EPI = EpiStoreHT", name, tall ? 4 : (big ? 2 : 1), tall ? 2 : (big ? 2 : 1));
	ProfScope prof(prof_name, 2.0 * (double)M * (double)N * (double)K, -2.0 * ((double)M * K + (double)N * K + (double)M * N), M, N, K, 1, 1, stream);
	if (tall) {
		hipLaunchKernelGGL(HIP_KERNEL_NAME(mfma_gemm_f16_kernel<LA, LB, EpiStoreHT, 4, 2, 8>), dim3((unsigned)tall_tiles, 1, 1), dim3(GEMM_THREADS), 0, stream, la, lb, epi, (M + 255) / 256, (N + 127) / 128, K, K > 0 ? K : 1, 1, 0L, 0L, 0L, 0L, ko);
		HIP_ENFORCE(hipGetLastError());
		return CCV_NNC_EXEC_SUCCESS;
	}
	if (big && v8) hipLaunchKernelGGL(HIP_KERNEL_NAME(mfma_gemm_f16_kernel<LA, LB, EpiStoreHT, 2, 2, 8>), dim3((unsigned)big_tiles, 1, 1), dim3(GEMM_THREADS), 0, stream, la, lb, epi, (M + 127) / 128, (N + 127) / 128, K, K > 0 ? K : 1, 1, 0L, 0L, 0L, 0L, ko);
	else if (big) hipLaunchKernelGGL(HIP_KERNEL_NAME(mfma_gemm_f16_kernel<LA, LB, EpiStoreHT, 2, 2>), dim3((unsigned)big_tiles, 1, 1), dim3(GEMM_THREADS), 0, stream, la, lb, epi, (M + 127) / 128, (N + 127) / 128, K, K > 0 ? K : 1, 1, 0L, 0L, 0L, 0L, ko);
	else if (v8) hipLaunchKernelGGL(HIP_KERNEL_NAME(mfma_gemm_f16_kernel<LA, LB, EpiStoreHT, 1, 1, 8>), dim3((unsigned)(((M + 63) / 64) * ((N + 63) / 64)), 1, 1), dim3(GEMM_THREADS), 0, stream, la, lb, epi, (M + 63) / 64, (N + 63) / 64, K, K > 0 ? K : 1, 1, 0L, 0L, 0L, 0L, ko);
	else hipLaunchKernelGGL(HIP_KERNEL_NAME(mfma_gemm_f16_kernel<LA, LB, EpiStoreHT, 1, 1>), dim3((unsigned)(((M + 63) / 64) * ((N + 63) / 64)), 1, 1), dim3(GEMM_THREADS), 0, stream, la, lb, epi, (M + 63) / 64, (N + 63) / 64, K, K > 0 ? K : 1, 1, 0L, 0L, 0L, 0L, ko);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}

// The buffer-load form of a half-precision plain-matrix operand (mfma_gemm_f16_buf.h): dword-aligned chunk addresses, offsets inside 31 bits.
template <bool KC>
static inline bool buf_loader_h_ok(const MatLoader<KC, true>& l, const long z_off, const int zcount)
{
	if (l.R <= 0 || l.K <= 0 || (((uintptr_t)l.p) & 15)) return false;
	if (zcount > 1 && (z_off & 7)) return false;
	if (KC) return l.ldk == 1 && l.ldr > 0 && l.ldr % 8 == 0 && l.ldr * 256 * 2 + (long)l.K * 2 < 0x7fffffffL;
	// row-contiguous: the descriptor's range ends with element (R - 1, K - 1), and the raw buffer's range check is per DWORD -- with an odd R (a padded view,
	// ldk > R) the dword holding that last element would straddle the range and read as zeros: the last k term of row R - 1 lost
	return l.ldr == 1 && l.R % 2 == 0 && l.ldk > 0 && l.ldk % 2 == 0 && ((long)l.K * l.ldk + l.R) * 2 < 0x7fffffffL;
}

template <bool AKC, bool BKC, int TM, int TN, int WM, int WN, int BK>
static int gemm_run_buf_tile_h(const char* name, const BufMatLoader<AKC>& la, const BufMatLoader<BKC>& lb, const GemmOutH out, const int M, const int N, const int K, const int zcount, const long a_z, const long b_z, const long c_z, const long bias_z, int splits, const int flags, ccv_nnc_stream_context_t* const ctx)
{
	hipStream_t stream = stream_of(ctx);
	constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
	const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
	const long tiles = (long)tiles_m * tiles_n;
	if (tiles > 0x7fffffffL) return CCV_NNC_EXEC_INVALID;
	if (splits <= 0) splits = (zcount == 1 && !(flags & CCV_NNC_ZERO_MEMORY_ALLOC)) ? gemm_auto_splits(tiles, K) : 1;
	int k_per_split = K;
	if (splits > 1) {
		if (splits < 8) splits = 8;
		splits = (splits + 7) & ~7;
		k_per_split = ((K + splits - 1) / splits + BK - 1) / BK * BK;
	}
	note_kernel(name);
	char prof_name[192];
	snprintf(prof_name, sizeof(prof_name), "%s|nnc::mfma_gemm_f16_buf_kernel<%d, %d, %d x %d x %d> EPI = %s", name, (int)AKC, (int)BKC, BM, BN, BK, splits <= 1 ? "EpiStoreH" : "EpiPartialH");
	const double flops = 2.0 * (double)M * (double)N * (double)K * (double)zcount;
	if (splits <= 1) {
		EpiStoreH epi;
		epi.c = out.c; epi.ldm = out.ldm; epi.ldn = out.ldn; epi.bias = out.bias; epi.alpha = out.alpha; epi.accumulate = out.accumulate; epi.M = M; epi.N = N; epi.bias_ldm = out.bias_ldm; epi.bias_ldn = out.bias_ldn;
		epi.vec = epi_vec_ok(out.c, sizeof(half_t), out.ldm, out.ldn, N, zcount, c_z);
		if (epi.vec && tune(TUNE_GEMM_VEC_EPILOGUE) != 3 && out.ldm % 8 == 0 && N % 8 == 0 && ((uintptr_t)out.c & 15) == 0 && (zcount <= 1 || c_z % 8 == 0)) epi.vec = 2; // 16-byte stores (TUNE_GEMM_VEC_EPILOGUE = 3: 8-byte ones, measurements)
		ProfScope prof(prof_name, flops, -2.0 * zcount * ((double)M * K + (double)N * K + (double)M * N), M, N, K, zcount, 1, stream);
		int sarg;
		const dim3 grid = gemm_batch_grid(tiles, zcount, &sarg);
		hipLaunchKernelGGL(HIP_KERNEL_NAME(mfma_gemm_f16_buf_kernel<AKC, BKC, EpiStoreH, TM, TN, WM, WN, BK>), grid, dim3(64 * WM * WN), 0, stream, la, lb, epi, tiles_m, tiles_n, K, K, sarg, a_z, b_z, c_z, bias_z);
		HIP_ENFORCE(hipGetLastError());
		return CCV_NNC_EXEC_SUCCESS;
	}
	const long slab = (long)M * N;
	float* ws = (float*)workspace_of(ctx, sizeof(float) * (size_t)slab * splits * zcount);
	if (!ws) return CCV_NNC_EXEC_OOM;
	EpiPartialH epi;
	epi.c = ws; epi.bias = 0; epi.slab = slab; epi.M = M; epi.N = N;
	epi.vec = tune(TUNE_GEMM_VEC_EPILOGUE) && N % 4 == 0;
	{
		ProfScope prof(prof_name, flops, -2.0 * zcount * ((double)M * K + (double)N * K + (double)M * N), M, N, K, zcount, splits, stream);
		hipLaunchKernelGGL(HIP_KERNEL_NAME(mfma_gemm_f16_buf_kernel<AKC, BKC, EpiPartialH, TM, TN, WM, WN, BK>), dim3((unsigned)(tiles * splits), 1, (unsigned)zcount), dim3(64 * WM * WN), 0, stream, la, lb, epi, tiles_m, tiles_n, K, k_per_split, splits, a_z, b_z, slab * splits, 0L);
	}
	HIP_ENFORCE(hipGetLastError());
	hipLaunchKernelGGL(HIP_KERNEL_NAME(splitk_reduce_kernel<half_t>), dim3(grid_for((size_t)slab * 4, 256), (unsigned)zcount), dim3(256), 0, stream, (const float*)ws, splits, slab, out.c, out.ldm, out.ldn, out.bias, out.bias_ldm, out.alpha, out.accumulate, M, N, c_z, bias_z, out.bias_ldn);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}

template <bool AKC, bool BKC>
static int gemm_run_buf_h(const char* name, const BufMatLoader<AKC>& la, const BufMatLoader<BKC>& lb, const GemmOutH out, const int M, const int N, const int K, const int zcount, const long a_z, const long b_z, const long c_z, const long bias_z, int splits, const int flags, ccv_nnc_stream_context_t* const ctx)
{
	// 256 x 256 (eight waves of 128 x 64) for the very large products only: measured on the MI355X 8192^3 715 vs 620 TFLOP/s, but 4096^3 545 vs 655 and
	// ResNet-50's 1x1 layers slower (one workgroup per CU: nothing hides its barrier) -- profiles/r03_v9_half_bench.txt
	const long t256 = (long)((M + 255) / 256) * ((N + 255) / 256);
	const long mode = tune(TUNE_GEMM_BUFFER_LOADS); // 1 = the rules below; 2 = never the 256 x 256 tile; 3 = 256 x 256 wherever it fits; 4 = BK 32 only (tests, tools/half_modes.sh)
	const bool k64 = K % 64 == 0 && mode != 4;
	if (mode != 2 && ((M >= 2048 && N >= 2048 && t256 * zcount >= 2 * device_cu_count()) || (mode == 3 && M >= 192 && N >= 192))) {
		if (k64 && AKC && BKC) return gemm_run_buf_tile_h<AKC, BKC, 4, 2, 2, 4, 64>( // (a transpose-read operand needs the registers of the second load set: K-steps of 32 there)
			name, la, lb, out, M, N, K, zcount, a_z, b_z, c_z, bias_z, splits, flags, ctx);
		return gemm_run_buf_tile_h<AKC, BKC, 4, 2, 2, 4, 32>(name, la, lb, out, M, N, K, zcount, a_z, b_z, c_z, bias_z, splits, flags, ctx);
	}
	// K-steps of 64 where K allows: half the barriers per MFMA (4096^3: 806 vs 655 TFLOP/s)
	if (k64) return gemm_run_buf_tile_h<AKC, BKC, 2, 2, 2, 2, 64>(name, la, lb, out, M, N, K, zcount, a_z, b_z, c_z, bias_z, splits, flags, ctx);
	return gemm_run_buf_tile_h<AKC, BKC, 2, 2, 2, 2, 32>(name, la, lb, out, M, N, K, zcount, a_z, b_z, c_z, bias_z, splits, flags, ctx);
}

// la.p / lb.p point at HALVES (cast to the loaders' float* type); every offset / stride / z offset is in elements.
template <class LA, class LB>
static int gemm_run_h(const char* name, LA la, LB lb, const GemmOutH out, const int M, const int N, const int K, const int zcount, const long a_z, const long b_z, const long c_z, const long bias_z, int splits, const int flags, ccv_nnc_stream_context_t* const ctx, const KOrder ko = KOrder(), const bool small_tiles = false)
{
	if (M <= 0 || N <= 0) return CCV_NNC_EXEC_SUCCESS;
	la.finish();
	lb.finish();
	const half_t* zp = (const half_t*)zero_page_of(ctx);
	la.zoff = zp - (const half_t*)la.p;
	lb.zoff = zp - (const half_t*)lb.p;
	// two tile shapes: 128 x 128, and 64 x 64 when an output dimension is <= 64 (channels) or the tile grid would not fill the chip
	const long big_tiles = (long)((M + 127) / 128) * ((N + 127) / 128);
	// two plain matrices, whole K-steps, the 128 x 128 tile: the buffer-load kernel (mfma_gemm_f16_buf.h: no VALU in the K loop, transpose reads for a row-contiguous operand)
	if constexpr (is_vec_mat_loader<LA>::value && is_vec_mat_loader<LB>::value) {
		if (tune(TUNE_GEMM_BUFFER_LOADS) && M > 64 && N > 64 && (big_tiles * zcount >= device_cu_count() || K >= 4096) && K > 0 && K % GEMM_BK == 0 && buf_loader_h_ok(la, a_z, zcount) && buf_loader_h_ok(lb, b_z, zcount)) {
			BufMatLoader<LA::KCONTIG> ba; BufMatLoader<LB::KCONTIG> bb;
			ba.p = la.p; ba.zoff = 0; ba.ldr = la.ldr; ba.ldk = la.ldk; ba.R = la.R; ba.K = la.K;
			bb.p = lb.p; bb.zoff = 0; bb.ldr = lb.ldr; bb.ldk = lb.ldk; bb.R = lb.R; bb.K = lb.K;
			return gemm_run_buf_h<LA::KCONTIG, LB::KCONTIG>(name, ba, bb, out, M, N, K, zcount, a_z, b_z, c_z, bias_z, splits, flags, ctx);
		}
	}
	if (g_force_tile) { // measurement aid: (1, 1) = the 64 x 64 tile, anything else the 128 x 128 one
		if ((g_force_tile & 0xff) == 1 && (g_force_tile >> 8) == 1) return gemm_run_tile_h<LA, LB, 1, 1>(name, la, lb, out, M, N, K, zcount, a_z, b_z, c_z, bias_z, splits, flags, ctx, ko);
		return gemm_run_tile_h<LA, LB, 2, 2>(name, la, lb, out, M, N, K, zcount, a_z, b_z, c_z, bias_z, splits, flags, ctx, ko);
	}
	if (!small_tiles && M > 64 && N > 64 && (big_tiles * zcount >= device_cu_count() || K >= 4096)) return gemm_run_tile_h<LA, LB, 2, 2>(name, la, lb, out, M, N, K, zcount, a_z, b_z, c_z, bias_z, splits, flags, ctx, ko);
	return gemm_run_tile_h<LA, LB, 1, 1>(name, la, lb, out, M, N, K, zcount, a_z, b_z, c_z, bias_z, splits, flags, ctx, ko);
}

} // namespace nnc

namespace nnc {

// One strided matrix operand of a contraction: element(r, k) = p[r * ldr + k * ldk], r in [0,R), k in [0,K).  T = float or half_t.
template <class T> struct MatOperandT { const T* p; long ldr, ldk; int R, K; };
typedef MatOperandT<float> MatOperand;
template <class T> struct gemm_out_of;
template <> struct gemm_out_of<float> { typedef GemmOut type; };
template <> struct gemm_out_of<half_t> { typedef GemmOutH type; };

template <class T>
static inline bool mat_vec_ok(const MatOperandT<T>& m, bool kc, long zoff, int zcount)
{ // 4-element chunks: 16-byte loads of floats, 8-byte loads of halves
	if (((uintptr_t)m.p) & (4 * sizeof(T) - 1)) return false;
	if (zcount > 1 && zoff % 4) return false;
	return kc ? (m.K % 4 == 0 && m.ldr % 4 == 0) : (m.R % 4 == 0 && m.ldk % 4 == 0);
}

// the launch of one loader pair, by element type: half precision has vector loaders only (callers check gemm_strided_ok first)
template <class T> struct gemm_dispatch;
template <> struct gemm_dispatch<float> {
	template <bool AKC, bool BKC, bool VEC>
	static int run(const char* name, const MatOperand& A, const MatOperand& B, const GemmOut& out, const int M, const int N, const int K, const int zcount, const long a_z, const long b_z, const long c_z, const long bias_z, const int flags, ccv_nnc_stream_context_t* const ctx)
	{
		MatLoader<AKC, VEC> la; la.p = A.p; la.ldr = A.ldr; la.ldk = A.ldk; la.R = M; la.K = K;
		MatLoader<BKC, VEC> lb; lb.p = B.p; lb.ldr = B.ldr; lb.ldk = B.ldk; lb.R = N; lb.K = K;
		return gemm_run(name, la, lb, out, M, N, K, zcount, a_z, b_z, c_z, bias_z, 0, flags, ctx);
	}
};
template <> struct gemm_dispatch<half_t> {
	template <bool AKC, bool BKC, bool VEC>
	static int run(const char* name, const MatOperandT<half_t>& A, const MatOperandT<half_t>& B, const GemmOutH& out, const int M, const int N, const int K, const int zcount, const long a_z, const long b_z, const long c_z, const long bias_z, const int flags, ccv_nnc_stream_context_t* const ctx)
	{
		if (!VEC) return CCV_NNC_EXEC_INVALID;
		MatLoader<AKC, true> la; la.p = (const float*)A.p; la.ldr = A.ldr; la.ldk = A.ldk; la.R = M; la.K = K;
		MatLoader<BKC, true> lb; lb.p = (const float*)B.p; lb.ldr = B.ldr; lb.ldk = B.ldk; lb.R = N; lb.K = K;
		return gemm_run_h(name, la, lb, out, M, N, K, zcount, a_z, b_z, c_z, bias_z, 0, flags, ctx);
	}
};

// Can both operands be read in 4-element chunks (the only form the half-precision core has)?
template <class T>
static bool gemm_strided_vec(const MatOperandT<T>& A, const MatOperandT<T>& B, const int zcount, const long a_z, const long b_z)
{
	const int K = A.K;
	const bool a_kc = (A.ldk == 1 || K == 1), b_kc = (B.ldk == 1 || K == 1);
	if (!a_kc && A.ldr != 1 && A.R != 1) return false;
	if (!b_kc && B.ldr != 1 && B.R != 1) return false;
	return mat_vec_ok(A, a_kc, a_z, zcount) && (a_kc ? A.ldk == 1 : A.ldr == 1) && mat_vec_ok(B, b_kc, b_z, zcount) && (b_kc ? B.ldk == 1 : B.ldr == 1);
}

// C(m, n) = alpha * sum_k A(m, k) * B(n, k) for arbitrary (unit-stride-in-one-dimension) operands.
template <class T>
static int gemm_strided(const char* name, const MatOperandT<T> A, const MatOperandT<T> B, const typename gemm_out_of<T>::type out, const int zcount, const long a_z, const long b_z, const long c_z, const long bias_z, const int flags, ccv_nnc_stream_context_t* const ctx)
{
	const int M = A.R, N = B.R, K = A.K;
	if (B.K != K) return CCV_NNC_EXEC_INVALID;
	const bool a_kc = (A.ldk == 1 || K == 1), b_kc = (B.ldk == 1 || K == 1);
	if (!a_kc && A.ldr != 1 && M != 1) return CCV_NNC_EXEC_INVALID;
	if (!b_kc && B.ldr != 1 && N != 1) return CCV_NNC_EXEC_INVALID;
	const bool vec = gemm_strided_vec(A, B, zcount, a_z, b_z);
#define NNC_GEMM_CASE(AKC, BKC, VEC) return gemm_dispatch<T>::template run<AKC, BKC, VEC>(name, A, B, out, M, N, K, zcount, a_z, b_z, c_z, bias_z, flags, ctx)
	// A degenerate (length-1) dimension makes both views legal; prefer the vectorisable one.
	if (a_kc && b_kc) { if (vec) NNC_GEMM_CASE(true, true, true); else NNC_GEMM_CASE(true, true, false); }
	else if (a_kc && !b_kc) { if (vec) NNC_GEMM_CASE(true, false, true); else NNC_GEMM_CASE(true, false, false); }
	else if (!a_kc && b_kc) { if (vec) NNC_GEMM_CASE(false, true, true); else NNC_GEMM_CASE(false, true, false); }
	else { if (vec) NNC_GEMM_CASE(false, false, true); else NNC_GEMM_CASE(false, false, false); }
#undef NNC_GEMM_CASE
}

} // namespace nnc
