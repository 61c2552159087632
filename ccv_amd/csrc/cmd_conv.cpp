// CCV_NNC_CONVOLUTION_FORWARD / BACKWARD on gfx950: implicit-GEMM on the fp32 MFMA contraction core.
// Semantics follow the reference CPU backend (the oracle):
//   forward   lib/nnc/cmd/convolution/ccv_nnc_conv_cpu_ref.c:13-172   b = bias + sum_{i,j,c} w[k,i,j,c] * a[n, y*s-p+i*d, x*s-p+j*d, g*Cg+c]
//   backward  lib/nnc/cmd/convolution/ccv_nnc_conv_cpu_ref.c:174-345  inputs (g, a, w) -> outputs (h, dw, dbias); CCV_NNC_ACCUMULATE_OUTPUT
//             accumulates into dw / dbias (:186-192); h is always overwritten (:286)
// and replace the cuDNN calls of lib/nnc/cmd/convolution/gpu/ccv_nnc_conv_gpu_cudnn.cu:24-114, 204-367.
// Layout on device: NHWC activations (n, y, x, c) with c innermost, weights [K][kh][kw][Cg]; tensor views are accepted for
// the inputs as long as the channel stride is 1.  NCHW tensors are routed through the layout kernels of cmd_util (workspace).
#include "gemm_launch.h"
#include "winograd.h"
#include "wino_fused.h"
#include "wino_wgrad_fused.h"
#include "conv_c3.h"

using namespace nnc;

namespace {

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// NNC_MI355X_CONV_ALGO_FUSE_RELU (include/nnc_mi355x.h): the forward command in progress on this thread was asked to write
// max(0, .); a kernel that does so in its epilogue says so, otherwise _conv_forw_any rectifies the output afterwards.
static thread_local int tl_relu_want = 0, tl_relu_done = 0;
// ... and on the way back (the same bit on CONVOLUTION_BACKWARD): the data gradient is masked by a > 0, a being the command's forward
// input -- a ReLU's output.  _conv_back publishes a's NHWC image while the data gradient runs; a kernel that masked as it wrote says so.
static thread_local int tl_mask_want = 0, tl_mask_done = 0;
static thread_local Image4 tl_mask = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
static bool mask_fits(const Image4& dst)
{
	const Image4& m = tl_mask;
	return m.p && m.n == dst.n && m.h == dst.h && m.w == dst.w && m.c == dst.c && m.sc == 1 && aligned16(m.p) && m.sw % 4 == 0 && m.sh % 4 == 0 && (m.n == 1 || m.sn % 4 == 0);
}

struct conv_geom_t {
	int N, H, W, C;     // input
	int OH, OW, K;      // output
	int kh, kw, Cg, Kg, groups;
	int sy, sx, pby, pbx, dy, dx;
};

static bool conv_geometry(const ccv_nnc_cmd_t& cmd, const ccv_nnc_hint_t& hint, const Image4& a, const Image4& b, const ccv_nnc_tensor_t* w, conv_geom_t* g)
{
	g->N = a.n; g->H = a.h; g->W = a.w; g->C = a.c;
	g->OH = b.h; g->OW = b.w; g->K = b.c;
	g->groups = cmd.info.convolution.groups > 0 ? cmd.info.convolution.groups : 1;
	g->kh = cmd.info.size.dim[0]; g->kw = cmd.info.size.dim[1];
	if (g->K != cmd.info.convolution.count || g->K % g->groups || g->C % g->groups || a.n != b.n) return false;
	g->Cg = g->C / g->groups; g->Kg = g->K / g->groups;
	g->sy = hint.stride.dim[0] > 0 ? hint.stride.dim[0] : 1;
	g->sx = hint.stride.dim[1] > 0 ? hint.stride.dim[1] : 1;
	g->pby = hint.border.begin[0]; g->pbx = hint.border.begin[1];
	g->dy = cmd.info.convolution.dilation[0] > 1 ? cmd.info.convolution.dilation[0] : 1;
	g->dx = cmd.info.convolution.dilation[1] > 1 ? cmd.info.convolution.dilation[1] : 1;
	(void)w; // weight layout / shape: weights_shape()
	return true;
}

static bool image_fits_int(const Image4& t)
{ // the gathers index inside one image with 32-bit arithmetic
	return (long)t.h * t.sh < 0x7fffffffL && (long)t.w * t.sw < 0x7fffffffL && (long)(t.h - 1) * t.sh + (long)(t.w - 1) * t.sw + t.c < 0x7fffffffL;
}

static bool pixel_linear(const Image4& t)
{ // pixels (n, y, x) form one arithmetic progression with step sw and channels are dense
	return t.sc == 1 && t.sh == (long)t.w * t.sw && (t.n == 1 || t.sn == (long)t.h * t.sh);
}

// A 1 x 1, stride-1, unpadded, ungrouped convolution over pixel-linear NHWC images IS a product of plain matrices -- [pixels][C] . [K][C]^T -- and goes to the
// contraction launcher as such: buffer loads without address arithmetic in the K loop and, in fp32, the split form on the bf16 matrix pipe (gemm_launch.h
// gemm_bf16x3_tile), neither of which the im2col loaders can take.  (Round 6: ResNet-50's 7 x 7 maps -- 49 pixels, no multiple of four, so the NCHW-native
// route of conv1x1_nchw_* does not apply -- ran their 512 <-> 2048 layers through the im2col walk at 77 - 87 TFLOP/s, 5.8 ms of the fp32 step.)
static bool conv_pointwise(const conv_geom_t& g, const Image4& in, const Image4& out)
{
	return g.kh == 1 && g.kw == 1 && g.sy == 1 && g.sx == 1 && g.pby == 0 && g.pbx == 0 && g.groups == 1 && g.OH == g.H && g.OW == g.W && pixel_linear(in) && pixel_linear(out)
		&& (long)g.N * g.H * g.W <= 0x7fffffffL;
}

// ---- Winograd F(4x4, 3x3) (winograd.h) --------------------------------------------------------------------------------
// Algorithm numbers of the two conv rows (ccv_nnc_cmd_t.algorithm; -1 = the backend's own choice; what autotune returns).
// 2 = the fused Winograd kernel (wino_fused.h) for forward and the data gradient; the filter gradient under 2 is algorithm 1's.
enum { CONV_ALGO_IMPLICIT_GEMM = 0, CONV_ALGO_WINOGRAD = 1, CONV_ALGO_WINOGRAD_FUSED = 2, CONV_ALGO_COUNT = 3 };

// the Winograd transform kernels are one-thread-per-item (no grid-stride loop): launch exactly ceil(n / threads) blocks
static unsigned blocks_exact(const size_t n, const int threads) { return (unsigned)((n + threads - 1) / threads); }

struct wino_plan_t {
	int TH, TW, T;
	size_t u_bytes, v_bytes, m_bytes;
	size_t total() const { return u_bytes + v_bytes + m_bytes; }
};

// `src` = the tensor the 6x6 tiles are cut from (C_src channels), `dst` = the tensor the 4x4 tiles are written to (C_dst).
static bool wino_plan(const conv_geom_t& g, const int dst_h, const int dst_w, const int C_src, const int C_dst, wino_plan_t* p)
{
	if (g.kh != 3 || g.kw != 3 || g.sy != 1 || g.sx != 1 || g.dy != 1 || g.dx != 1 || g.groups != 1) return false;
	if (C_src % 4 || C_dst % 4 || g.pby < 0 || g.pby > 2 || g.pbx < 0 || g.pbx > 2) return false;
	p->TH = (dst_h + 3) / 4; p->TW = (dst_w + 3) / 4;
	const long T = (long)g.N * p->TH * p->TW;
	if (T <= 0 || T * (C_src > C_dst ? C_src : C_dst) > 0x7fffffffL) return false;
	p->T = (int)T;
	p->u_bytes = (sizeof(float) * 36 * (size_t)C_src * C_dst + 255) & ~(size_t)255;
	p->v_bytes = (sizeof(float) * 36 * (size_t)T * C_src + 255) & ~(size_t)255;
	p->m_bytes = (sizeof(float) * 36 * (size_t)T * C_dst + 255) & ~(size_t)255;
	return true;
}

// Images per slice for the Winograd-via-HBM stages (TUNE_WINO_SLICE_KB): the transformed images V and M of the whole batch
// are far beyond every cache (conv1_2 at batch 256: 7.4 GB each), so each is written to HBM and read back once.  Run per
// slice of images instead, V + M of a slice stay within the 256 MB Infinity Cache and the same scratch addresses are
// reused slice after slice: the contraction reads V, and the output transform reads M, from on-die memory.
static int wino_slice_images(const int N, const int tiles_per_image, const int C_src, const int C_dst)
{
	const long kb = tune(TUNE_WINO_SLICE_KB);
	if (kb <= 0) return N;
	const double per_image = 36.0 * sizeof(float) * (double)tiles_per_image * (double)(C_src + C_dst);
	long nb = (long)((double)kb * 1024.0 / per_image);
	if (nb < 1) nb = 1;
	return nb > N ? N : (int)nb;
}

// dst (+ bias) = conv3x3(src, w), stride 1, source padding (pad_y, pad_x);  FLIP: dgrad's mirrored / role-swapped weights.
template <bool FLIP>
static int conv_wino_run(const char* name, const conv_geom_t& g, const wino_plan_t& p, const Image4& src, const float* w, const float* bias, const Image4& dst, const int pad_y, const int pad_x, const int flags, ccv_nnc_stream_context_t* const ctx)
{
	const int Cs = src.c, Cd = dst.c;
	const int per_image = p.TH * p.TW;
	const int nb = wino_slice_images(g.N, per_image, Cs, Cd);
	const size_t v_bytes = nb == g.N ? p.v_bytes : (sizeof(float) * 36 * (size_t)nb * per_image * Cs + 255) & ~(size_t)255;
	const size_t m_bytes = nb == g.N ? p.m_bytes : (sizeof(float) * 36 * (size_t)nb * per_image * Cd + 255) & ~(size_t)255;
	char* ws = (char*)workspace_of(ctx, p.u_bytes + v_bytes + m_bytes);
	if (!ws) return CCV_NNC_EXEC_OOM;
	float* const U = (float*)ws;
	float* const V = (float*)(ws + p.u_bytes);
	float* const M = (float*)(ws + p.u_bytes + v_bytes);
	hipStream_t stream = stream_of(ctx);
	hipLaunchKernelGGL(HIP_KERNEL_NAME(wino_weight_kernel<FLIP>), dim3(blocks_exact((size_t)Cs * Cd, 256)), dim3(256), 0, stream, w, U, g.K, g.C);
	HIP_ENFORCE(hipGetLastError());
	for (int n0 = 0; n0 < g.N; n0 += nb) {
		const int ns = g.N - n0 < nb ? g.N - n0 : nb;
		const int T = ns * per_image;
		WinoTiles ti;
		ti.TH = p.TH; ti.TW = p.TW; ti.T = T;
		ti.H = src.h; ti.W = src.w; ti.sn = src.sn; ti.sh = src.sh; ti.sw = src.sw; ti.oy = -pad_y; ti.ox = -pad_x; ti.C4 = Cs / 4;
		ti.relu = (!FLIP && tl_relu_want) ? 1 : 0;
		if (ti.relu) tl_relu_done = 1;
		ti.mask = 0; ti.m_sn = ti.m_sh = ti.m_sw = 0;
		ti.d_c4.init(ti.C4); ti.d_tw.init(ti.TW); ti.d_th.init(ti.TH);
		hipLaunchKernelGGL(wino_input_kernel, dim3(blocks_exact((size_t)T * ti.C4, 256)), dim3(256), 0, stream, (const float*)src.p + (long)n0 * src.sn, V, ti);
		HIP_ENFORCE(hipGetLastError());
		// 36 GEMMs M[z] (T x Cd) = V[z] (T x Cs) * U[z]^T (Cd x Cs), both operands reduction-contiguous, one launch (grid z)
		MatLoader<true, true> la, lb;
		la.p = V; la.ldr = Cs; la.ldk = 1; la.R = T; la.K = Cs;
		lb.p = U; lb.ldr = Cs; lb.ldk = 1; lb.R = Cd; lb.K = Cs;
		GemmOut out = { M, Cd, 1, 0, 1.f, 0 };
		const int ret = gemm_run(name, la, lb, out, T, Cd, Cs, 36, (long)T * Cs, (long)Cd * Cs, (long)T * Cd, 0L, 1, flags, ctx);
		if (ret != CCV_NNC_EXEC_SUCCESS) return ret;
		ti.H = dst.h; ti.W = dst.w; ti.sn = dst.sn; ti.sh = dst.sh; ti.sw = dst.sw; ti.C4 = Cd / 4;
		ti.d_c4.init(ti.C4);
		if (FLIP && mask_fits(dst)) { // data gradient under a ReLU backward: the output transform reads the map's 16 bytes next to each store
			ti.mask = tl_mask.p + (long)n0 * tl_mask.sn; ti.m_sn = tl_mask.sn; ti.m_sh = tl_mask.sh; ti.m_sw = tl_mask.sw;
			tl_mask_done = 1;
		}
		hipLaunchKernelGGL(wino_output_kernel, dim3(blocks_exact((size_t)T * ti.C4, 256)), dim3(256), 0, stream, (const float*)M, bias, dst.p + (long)n0 * dst.sn, ti);
		HIP_ENFORCE(hipGetLastError());
	}
	return CCV_NNC_EXEC_SUCCESS;
}

// ---- fused Winograd (wino_fused.h): dst (+ bias) = conv3x3(src, w) with neither V nor M in HBM ------------------------------
struct wino_fused_plan_t {
	int GH, GW, GYn, GXn, groups, KB, CCn;
	size_t uf_bytes;
};

static bool wino_fused_plan(const conv_geom_t& g, const Image4& src, const Image4& dst, wino_fused_plan_t* p)
{
	if (g.kh != 3 || g.kw != 3 || g.sy != 1 || g.sx != 1 || g.dy != 1 || g.dx != 1 || g.groups != 1) return false;
	if (src.c % WF_CC || src.c < 2 * WF_CC || src.sc != 1 || dst.sc != 1 || !aligned16(src.p) || src.sw % 4 || src.sh % 4 || (src.n > 1 && src.sn % 4)) return false;
	if (dst.c % 4 || !aligned16(dst.p) || dst.sw % 4 || dst.sh % 4 || (dst.n > 1 && dst.sn % 4)) return false; // 16-byte stores
	if (((long)(dst.h - 1) * dst.sh + (long)(dst.w - 1) * dst.sw + dst.c) * 4 >= (long)WF_OOB) return false;
	if (((long)(src.h - 1) * src.sh + (long)(src.w - 1) * src.sw + src.c) * 4 >= (long)WF_OOB) return false; // per-image buffer descriptor range
	const int TH = (dst.h + 3) / 4, TW = (dst.w + 3) / 4;
	static const int shapes[3][2] = { { 4, 4 }, { 2, 8 }, { 8, 2 } };
	long best = -1;
	for (int i = 0; i < 3; i++) { // least padding of the tile grid; ties go to the squarer group (smaller patch region)
		const long cover = (long)((TH + shapes[i][0] - 1) / shapes[i][0]) * ((TW + shapes[i][1] - 1) / shapes[i][1]);
		if (best < 0 || cover < best) { best = cover; p->GH = shapes[i][0]; p->GW = shapes[i][1]; }
	}
	p->GYn = (TH + p->GH - 1) / p->GH; p->GXn = (TW + p->GW - 1) / p->GW;
	const long groups = (long)src.n * p->GYn * p->GXn;
	p->KB = (dst.c + WF_KT - 1) / WF_KT; p->CCn = src.c / WF_CC;
	if (groups <= 0 || (groups + 3) / 4 * p->KB > 0x7fffffffL || (size_t)p->CCn * WF_U_FLOATS * 4 > 0xffffffffUL) return false;
	p->groups = (int)groups;
	p->uf_bytes = (sizeof(float) * (size_t)p->KB * p->CCn * WF_U_FLOATS + 255) & ~(size_t)255;
	return true;
}

// Most scratch the fused kernel asks for (its U fragments), for the scopes that stage layouts in front of it.
static size_t wino_fused_scratch_bound(const int Kout, const int Cred)
{
	if (Cred % WF_CC || Kout <= 0) return 0;
	return ((sizeof(float) * (size_t)((Kout + WF_KT - 1) / WF_KT) * (Cred / WF_CC) * WF_U_FLOATS + 255) & ~(size_t)255);
}

// ... and its mask bits (data gradient under a ReLU backward): 1 KB per (tile group, 32-channel block) of the gradient written
static size_t wino_fused_mask_bound(const conv_geom_t& g)
{
	if (!tl_mask_want) return 0;
	const long TH = (g.H + 3) / 4, TW = (g.W + 3) / 4;
	long most = 0;
	static const int shapes[3][2] = { { 4, 4 }, { 2, 8 }, { 8, 2 } };
	for (int i = 0; i < 3; i++) {
		const long cover = ((TH + shapes[i][0] - 1) / shapes[i][0]) * ((TW + shapes[i][1] - 1) / shapes[i][1]);
		if (cover > most) most = cover;
	}
	return (size_t)g.N * most * ((g.C + WF_KT - 1) / WF_KT) * 1024;
}

template <bool FLIP>
static int conv_wino_fused_run(const char* name, const conv_geom_t& g, const wino_fused_plan_t& p, const Image4& src, const float* w, const float* bias, const Image4& dst, const int pad_y, const int pad_x, ccv_nnc_stream_context_t* const ctx)
{
	// data gradient under a ReLU backward: the mask as bits in the epilogue's order, packed first (1 / 32 of the map; no room: unmasked, the caller's pass follows)
	size_t bits_bytes = FLIP && mask_fits(dst) && (long)p.groups * p.KB <= 0x7fffffffL ? (size_t)p.groups * p.KB * 1024 : 0;
	float* UF = (float*)workspace_of(ctx, p.uf_bytes + bits_bytes);
	if (!UF && bits_bytes) { bits_bytes = 0; UF = (float*)workspace_of(ctx, p.uf_bytes); }
	if (!UF) return CCV_NNC_EXEC_OOM;
	hipStream_t stream = stream_of(ctx);
	unsigned* const bits = bits_bytes ? (unsigned*)((char*)UF + p.uf_bytes) : 0;
	if (bits) {
		const dim3 grid((unsigned)((long)p.groups * p.KB));
		const Image4& m = tl_mask;
		if (p.GH == 4) hipLaunchKernelGGL(HIP_KERNEL_NAME(wino_mask_pack_kernel<4, 4>), grid, dim3(256), 0, stream, (const float*)m.p, m.sn, m.sh, m.sw, bits, dst.h, dst.w, dst.c, p.GYn, p.GXn, p.KB);
		else if (p.GH == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(wino_mask_pack_kernel<2, 8>), grid, dim3(256), 0, stream, (const float*)m.p, m.sn, m.sh, m.sw, bits, dst.h, dst.w, dst.c, p.GYn, p.GXn, p.KB);
		else hipLaunchKernelGGL(HIP_KERNEL_NAME(wino_mask_pack_kernel<8, 2>), grid, dim3(256), 0, stream, (const float*)m.p, m.sn, m.sh, m.sw, bits, dst.h, dst.w, dst.c, p.GYn, p.GXn, p.KB);
		HIP_ENFORCE(hipGetLastError());
		tl_mask_done = 1;
	}
	const int Kout = dst.c, Cred = src.c;
	hipLaunchKernelGGL(HIP_KERNEL_NAME(wino_weight_frag_kernel<FLIP>), dim3(blocks_exact((size_t)p.KB * WF_KT * Cred, 256)), dim3(256), 0, stream, w, UF, Kout, Cred, g.K, g.C);
	HIP_ENFORCE(hipGetLastError());
	WinoFusedArgs a;
	a.src = src.p; a.dst = dst.p; a.uf = UF; a.bias = bias;
	a.s_sn = src.sn; a.s_sh = src.sh; a.s_sw = src.sw; a.d_sn = dst.sn; a.d_sh = dst.sh; a.d_sw = dst.sw;
	a.H = src.h; a.W = src.w; a.OH = dst.h; a.OW = dst.w; a.pad_y = pad_y; a.pad_x = pad_x;
	a.GYn = p.GYn; a.GXn = p.GXn; a.groups = p.groups; a.C = Cred; a.K = Kout; a.CCn = p.CCn; a.KB = p.KB;
	a.dst_image_bytes = (unsigned)(((long)(dst.h - 1) * dst.sh + (long)(dst.w - 1) * dst.sw + dst.c) * 4);
	a.src_image_bytes = (unsigned)(((long)(src.h - 1) * src.sh + (long)(src.w - 1) * src.sw + src.c) * 4);
	a.uf_kb_bytes = (unsigned)((size_t)p.CCn * WF_U_FLOATS * 4);
	// persistent: one workgroup per CU; teams of `team` workgroups (a divisor of KB, at most 8) share a range of tile-group quads
	// on one XCD (see the kernel); the grid is a multiple of 8 * team, workgroups beyond the work exit at once
	long wgs = tune(TUNE_WINO_FUSED_GRID) > 0 ? tune(TUNE_WINO_FUSED_GRID) : device_cu_count();
	int team = p.KB % 8 == 0 ? 8 : (p.KB % 4 == 0 ? 4 : (p.KB % 2 == 0 ? 2 : 1)); // (8 since round 5: 256 -> 256 at 56 x 56 0.81 -> 0.74 ms at batch 64, even at batch 256: tools/wf5_probe.cpp)
	while (team > 1 && wgs < 8 * team) team >>= 1;
	if (wgs < 8 * team) wgs = 8 * team;
	const unsigned grid = (unsigned)(wgs / (8 * team) * (8 * team));
	a.team = team;
	a.relu = (!FLIP && tl_relu_want) ? 1 : 0;
	if (a.relu) tl_relu_done = 1;
	a.mask_bits = bits;
	note_kernel(name);
	char prof_name[96];
	snprintf(prof_name, sizeof(prof_name), bits ? "%s|nnc::wino_fused_kernel<%d, %d, 0, true>" : "%s|nnc::wino_fused_kernel<%d, %d>", name, p.GH, p.GW);
	const long T = (long)src.n * ((dst.h + 3) / 4) * ((dst.w + 3) / 4);
	// (the PAIRED patch schedule of wino_fused.h -- SCHED = 1 + SKEW -- is not instantiated here: measured slower on the MI355X, tools/wf_probe.cpp,
	// profiles/r03_v2_wf_probe_paired_schedule.txt)
	ProfScope prof(prof_name, 2.0 * 36.0 * (double)T * Kout * Cred, 0, (int)T, Kout, Cred, 36, 1, stream);
	// (tools/wino_fused2.h: the same decomposition on two waves per SIMD, each wave half of the transform-domain columns -- measured on the MI355X, not
	// faster: VALU costs matrix-pipe time whichever wave issues it, profiles/r03_v7_issue_probes.txt -- is an experiment, not part of the library)
	if (bits) {
		if (p.GH == 4) hipLaunchKernelGGL(HIP_KERNEL_NAME(wino_fused_kernel<4, 4, 0, true>), dim3(grid), dim3(256), 0, stream, a);
		else if (p.GH == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(wino_fused_kernel<2, 8, 0, true>), dim3(grid), dim3(256), 0, stream, a);
		else hipLaunchKernelGGL(HIP_KERNEL_NAME(wino_fused_kernel<8, 2, 0, true>), dim3(grid), dim3(256), 0, stream, a);
	} else if (p.GH == 4) hipLaunchKernelGGL(HIP_KERNEL_NAME(wino_fused_kernel<4, 4>), dim3(grid), dim3(256), 0, stream, a);
	else if (p.GH == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(wino_fused_kernel<2, 8>), dim3(grid), dim3(256), 0, stream, a);
	else hipLaunchKernelGGL(HIP_KERNEL_NAME(wino_fused_kernel<8, 2>), dim3(grid), dim3(256), 0, stream, a);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}

// algorithm -1: the fused kernel where it measures faster on the MI355X (tools/conv_algo_sweep.py, profiles/r03_v8_conv_algo_sweep.txt): up to
// TUNE_WINO_FUSED_MAX_C reduction channels (default 128), and up to twice that when the 16-tile groups pad the tile grid by less than a fifth (256 -> 256 at
// 55^2: 2.53 vs 3.13 ms via HBM).  Round 3 tried 256 as the default -- at batch 256 it is 0.2 ms of the VGG-D step (512 -> 512 at 13^2: 0.74 vs 0.80 ms; 256 -> 512
// at 27^2 even), but at batch 64 the same layers have 4 items per persistent workgroup and run 51 TFLOP/s (forward 8513 -> 7033 images/s), and DawnNet's 256-channel
// layers at 8^2 pad four-fold: the rule stays where few items or heavy padding cannot hurt it.
static bool wino_fused_preferred(const int C_red, const wino_fused_plan_t& p, const Image4& dst)
{
	const long maxc = tune(TUNE_WINO_FUSED_MAX_C);
	if (C_red <= maxc) return true;
	const long tiles = (long)((dst.h + 3) / 4) * ((dst.w + 3) / 4), padded = (long)p.GYn * p.GH * p.GXn * p.GW;
	return C_red <= 2 * maxc && padded * 5 < tiles * 6;
}

// dw (+)= sum over tiles: the F(3x3, 4x4) form -- V = B^T a B exactly as in forward, W = G' g G'^T on the output gradient,
// 36 contractions dU[z] (K x C) = W[z]^T V[z] over the T tiles (batched split-K: both operands are read along their
// contiguous channel rows, the reduction index strides over tiles), then dw = A'^T dU A'.
struct wino_wgrad_plan_t {
	wino_plan_t t;
	int splits;
	size_t w_bytes, du_bytes, head_bytes, bp_bytes;
	long blocks; // of the output-gradient transform
	size_t total() const { return head_bytes + t.v_bytes + w_bytes + du_bytes + bp_bytes; }
};

static bool wino_wgrad_plan(const conv_geom_t& g, wino_wgrad_plan_t* p)
{
	if (!wino_plan(g, g.OH, g.OW, g.C, g.K, &p->t)) return false;
	// enough K-slices that 36 x tiles x slices fills the chip ~4 times over, each slice keeping >= 8 K-steps of tiles
	const long tiles = (long)((g.K + 127) / 128) * ((g.C + 127) / 128) * 36;
	long s = ((long)device_cu_count() * 8 + tiles - 1) / tiles;
	const long max_s = p->t.T / (GEMM_BK * 8);
	if (s > max_s) s = max_s;
	if (s > 64) s = 64;
	p->splits = s <= 1 ? 1 : (int)(((s < 8 ? 8 : s) + 7) & ~7);
	p->w_bytes = p->t.m_bytes; // 36 x T x K, like forward's M
	p->du_bytes = (sizeof(float) * 36 * (size_t)g.K * g.C + 255) & ~(size_t)255;
	// head: what the calls made underneath take from the base of the workspace -- gemm_run's slab sets, colsum_f32's partials
	p->head_bytes = p->splits > 1 ? sizeof(float) * 36 * (size_t)g.K * g.C * p->splits : 0;
	const size_t colsum_bound = sizeof(float) * (size_t)device_cu_count() * 4 * g.K;
	if (p->head_bytes < colsum_bound) p->head_bytes = colsum_bound;
	p->head_bytes = (p->head_bytes + 255) & ~(size_t)255;
	p->blocks = ((long)p->t.T * (g.K / 4) + 255) / 256;
	p->bp_bytes = sizeof(float) * (size_t)p->blocks * g.K;
	return true;
}

// dbias != 0: also produce the bias gradient (column sums of gr) inside the output-gradient transform; *bias_done tells the
// caller whether that happened (it needs (K / 4) | 256).
static int conv_wino_wgrad(const conv_geom_t& g, const wino_wgrad_plan_t& p, const Image4& gr, const Image4& a, float* dw, float* dbias, bool* bias_done, const int flags, ccv_nnc_stream_context_t* const ctx)
{
	// [ head: nested calls' scratch (they take the workspace base) | V | W | dU | per-block column sums ]
	// (sized for the whole batch: a slice -- TUNE_WINO_SLICE_KB -- uses the front of each region)
	char* ws = (char*)workspace_of(ctx, p.total());
	if (!ws) return CCV_NNC_EXEC_OOM;
	float* const V = (float*)(ws + p.head_bytes);
	float* const W = (float*)(ws + p.head_bytes + p.t.v_bytes);
	float* const dU = (float*)(ws + p.head_bytes + p.t.v_bytes + p.w_bytes);
	float* const BP = (float*)(ws + p.head_bytes + p.t.v_bytes + p.w_bytes + p.du_bytes);
	const bool fuse_bias = dbias && 256 % (g.K / 4) == 0;
	const int acc = (flags & CCV_NNC_ACCUMULATE_OUTPUT) ? 1 : 0;
	hipStream_t stream = stream_of(ctx);
	const int per_image = p.t.TH * p.t.TW;
	const int nb = wino_slice_images(g.N, per_image, g.C, g.K);
	for (int n0 = 0; n0 < g.N; n0 += nb) {
		const int ns = g.N - n0 < nb ? g.N - n0 : nb;
		const int T = ns * per_image;
		WinoTiles ti;
		ti.TH = p.t.TH; ti.TW = p.t.TW; ti.T = T;
		ti.H = a.h; ti.W = a.w; ti.sn = a.sn; ti.sh = a.sh; ti.sw = a.sw; ti.oy = -g.pby; ti.ox = -g.pbx; ti.C4 = g.C / 4;
		ti.d_c4.init(ti.C4); ti.d_tw.init(ti.TW); ti.d_th.init(ti.TH);
		hipLaunchKernelGGL(wino_input_kernel, dim3(blocks_exact((size_t)T * ti.C4, 256)), dim3(256), 0, stream, (const float*)a.p + (long)n0 * a.sn, V, ti);
		HIP_ENFORCE(hipGetLastError());
		ti.H = gr.h; ti.W = gr.w; ti.sn = gr.sn; ti.sh = gr.sh; ti.sw = gr.sw; ti.oy = 0; ti.ox = 0; ti.C4 = g.K / 4;
		ti.d_c4.init(ti.C4);
		const long blocks = ((long)T * (g.K / 4) + 255) / 256;
		const float* const grs = (const float*)gr.p + (long)n0 * gr.sn;
		if (fuse_bias) hipLaunchKernelGGL(HIP_KERNEL_NAME(wino_outgrad_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, stream, grs, W, ti, BP);
		else hipLaunchKernelGGL(HIP_KERNEL_NAME(wino_outgrad_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, stream, grs, W, ti, (float*)0);
		HIP_ENFORCE(hipGetLastError());
		MatLoader<false, true> la, lb; // rows = channels (contiguous), reduction index = tile (stride = channel count)
		la.p = W; la.ldr = 1; la.ldk = g.K; la.R = g.K; la.K = T;
		lb.p = V; lb.ldr = 1; lb.ldk = g.C; lb.R = g.C; lb.K = T;
		GemmOut out = { dU, g.C, 1, 0, 1.f, n0 > 0 ? 1 : 0 }; // later slices add onto the first one's dU
		int splits = p.splits;
		if (nb < g.N) { // a slice has fewer tiles: keep >= 8 K-steps per K-slice
			const long max_s = T / (GEMM_BK * 8);
			if (splits > max_s) splits = max_s <= 1 ? 1 : (int)(max_s & ~7L);
			if (splits < 8) splits = 1;
		}
		const int ret = gemm_run("conv_wgrad_wino", la, lb, out, g.K, g.C, T, 36, (long)T * g.K, (long)T * g.C, (long)g.K * g.C, 0L, splits, flags, ctx);
		if (ret != CCV_NNC_EXEC_SUCCESS) return ret;
		if (fuse_bias) { // fold the per-block rows; its own partials land in the head region, the slice's W is dead by now
			const int r = colsum_f32(BP, blocks, g.K, g.K, dbias, acc || n0 > 0, ctx);
			if (r != CCV_NNC_EXEC_SUCCESS) return r;
		}
	}
	hipLaunchKernelGGL(wino_wgrad_final_kernel, dim3(blocks_exact((size_t)g.K * g.C, 256)), dim3(256), 0, stream, (const float*)dU, dw, g.K, g.C, acc);
	HIP_ENFORCE(hipGetLastError());
	if (bias_done) *bias_done = fuse_bias;
	return CCV_NNC_EXEC_SUCCESS;
}

// ---- fused Winograd filter gradient (wino_wgrad_fused.h): dw (+)= , dbias (+)= with neither V nor W in HBM ------------------------------------
struct wino_wgrad_fused_plan_t {
	int TH, TW, GYn, GXn, groups, kblocks, cblocks, slices, per_slice;
	size_t partial_bytes, bias_bytes, du_bytes;
	size_t total() const { return partial_bytes + bias_bytes + du_bytes; }
};
static bool wino_wgrad_fused_plan(const conv_geom_t& g, wino_wgrad_fused_plan_t* p)
{
	if (g.kh != 3 || g.kw != 3 || g.sy != 1 || g.sx != 1 || g.dy != 1 || g.dx != 1 || g.groups != 1) return false;
	if (g.C % WG_CB || g.K % WG_KB || g.pby < 0 || g.pby > 2 || g.pbx < 0 || g.pbx > 2) return false;
	p->TH = (g.OH + 3) / 4; p->TW = (g.OW + 3) / 4;
	p->GYn = (p->TH + WG_GH - 1) / WG_GH; p->GXn = (p->TW + WG_GW - 1) / WG_GW;
	const long groups = (long)g.N * p->GYn * p->GXn;
	if (groups <= 0 || groups > 0x7fffffffL) return false;
	p->groups = (int)groups;
	p->kblocks = g.K / WG_KB; p->cblocks = g.C / WG_CB;
	// one workgroup per CU: slices x blocks ~ the CU count, slices a multiple of 8 (one eighth per XCD), never more slices than tile groups / 8
	const long nb = (long)p->kblocks * p->cblocks;
	long s = (device_cu_count() / nb) & ~7L;
	if (s < 8) s = 8;
	while (s > 8 && (groups + s - 1) / s < 8) s -= 8;
	p->slices = (int)s;
	p->per_slice = (int)((groups + s - 1) / s);
	p->partial_bytes = (sizeof(float) * 36 * (size_t)s * g.K * g.C + 255) & ~(size_t)255;
	p->bias_bytes = (sizeof(float) * 4 * (size_t)s * g.K + 255) & ~(size_t)255;
	p->du_bytes = (sizeof(float) * 36 * (size_t)g.K * g.C + 255) & ~(size_t)255;
	return true;
}
static bool wino_wgrad_fused_images_ok(const Image4& a, const Image4& gr, const float* dw)
{
	const auto ok = [](const Image4& t) { return t.sc == 1 && aligned16(t.p) && t.sw % 4 == 0 && t.sh % 4 == 0 && (t.n == 1 || t.sn % 4 == 0) && ((long)(t.h - 1) * t.sh + (long)(t.w - 1) * t.sw + t.c) * 4 < 0x7ffff000L; };
	return ok(a) && ok(gr) && dw != 0;
}
// algorithm -1: where both channel counts are at most TUNE_WINO_WGRAD_FUSED_MAX (measured on the MI355X, DESIGN.md section 5.6: the blocks of a
// (K / 32) x (C / 64) grid each read and transform every tile, so the redundancy grows with the channel counts while the via-HBM GEMMs get better)
static bool wino_wgrad_fused_preferred(const conv_geom_t& g)
{
	const long m = tune(TUNE_WINO_WGRAD_FUSED_MAX);
	return m > 0 && g.C <= m && g.K <= m;
}
static int conv_wino_wgrad_fused(const conv_geom_t& g, const wino_wgrad_fused_plan_t& p, const Image4& gr, const Image4& a, float* dw, float* dbias, bool* bias_done, const int flags, ccv_nnc_stream_context_t* const ctx)
{
	char* const ws = (char*)workspace_of(ctx, p.total());
	if (!ws) return CCV_NNC_EXEC_OOM;
	hipStream_t stream = stream_of(ctx);
	WinoWgradFusedArgs k;
	k.a = a.p; k.g = gr.p; k.partial = (float*)ws; k.bias_partial = dbias ? (float*)(ws + p.partial_bytes) : 0;
	k.a_sn = a.sn; k.a_sh = a.sh; k.a_sw = a.sw; k.g_sn = gr.sn; k.g_sh = gr.sh; k.g_sw = gr.sw;
	k.H = a.h; k.W = a.w; k.OH = gr.h; k.OW = gr.w; k.pad_y = g.pby; k.pad_x = g.pbx;
	k.GYn = p.GYn; k.GXn = p.GXn; k.groups = p.groups; k.C = g.C; k.K = g.K; k.kblocks = p.kblocks; k.cblocks = p.cblocks;
	k.slices = p.slices; k.per_slice = p.per_slice;
	k.a_image_bytes = (unsigned)(((long)(a.h - 1) * a.sh + (long)(a.w - 1) * a.sw + a.c) * 4);
	k.g_image_bytes = (unsigned)(((long)(gr.h - 1) * gr.sh + (long)(gr.w - 1) * gr.sw + gr.c) * 4);
	note_kernel("conv_wgrad_wino_fused");
	{
		const long T = (long)g.N * p.TH * p.TW;
		ProfScope prof("conv_wgrad_wino_fused|nnc::wino_wgrad_fused_kernel", 2.0 * 36.0 * (double)T * g.K * g.C, 0, g.K, g.C, (int)T, 36, p.slices, stream);
		hipLaunchKernelGGL(HIP_KERNEL_NAME(wino_wgrad_fused_kernel<0>), dim3((unsigned)(p.slices * p.kblocks * p.cblocks)), dim3(256), 0, stream, k);
		HIP_ENFORCE(hipGetLastError());
	}
	const long n = 36L * g.K * g.C;
	const int acc = (flags & CCV_NNC_ACCUMULATE_OUTPUT) ? 1 : 0;
	float* const dU = (float*)(ws + p.partial_bytes + p.bias_bytes);
	hipLaunchKernelGGL(wino_wgrad_fused_fold_kernel, dim3(blocks_exact((size_t)n, 256)), dim3(256), 0, stream, (const float*)k.partial, (const float*)k.bias_partial, dU, dbias, n, g.K, p.slices, acc);
	HIP_ENFORCE(hipGetLastError());
	hipLaunchKernelGGL(wino_wgrad_final_kernel, dim3(blocks_exact((size_t)g.K * g.C, 256)), dim3(256), 0, stream, (const float*)dU, dw, g.K, g.C, acc);
	HIP_ENFORCE(hipGetLastError());
	if (bias_done) *bias_done = dbias != 0;
	return CCV_NNC_EXEC_SUCCESS;
}

static bool wino_images_ok(const Image4& src, const Image4& dst, const float* w, const float* bias)
{
	return src.sc == 1 && dst.sc == 1 && aligned16(src.p) && aligned16(dst.p) && aligned16(w) && (!bias || aligned16(bias)) &&
		src.sw % 4 == 0 && src.sh % 4 == 0 && (src.n == 1 || src.sn % 4 == 0) && dst.sw % 4 == 0 && dst.sh % 4 == 0 && (dst.n == 1 || dst.sn % 4 == 0);
}

// The backend's own choice (algorithm -1).  Measured on the MI355X at batch 256 (tools/conv_algo_sweep.py): Winograd
// is 2.1-3.2x faster than the implicit GEMM on EVERY 3x3 VGG-D layer from 64 channels up, so it is taken whenever the
// geometry allows and there are enough channels for the 36 GEMMs to have a K-step (32) of depth and a full N tile.
static bool wino_preferred(const wino_plan_t& p, const int C_src, const int C_dst)
{
	return C_src >= 32 && C_dst >= 32 && p.T >= 32;
}

// ---- first-layer convolution (3 input channels): conv_c3.h ----------------------------------------------------------------
static bool conv_c3_ok(const conv_geom_t& g, const Image4& a, const Image4& b)
{
	if (g.C != 3 || g.kh != 3 || g.kw != 3 || g.sy < 1 || g.sx < 1 || g.sy > 4 || g.sx > 4 || g.dy != 1 || g.dx != 1 || g.groups != 1) return false;
	if (g.K != 16 && g.K != 32 && g.K != 64) return false;
	if (a.sc != 1 || a.sw != 3 || b.sc != 1 || !aligned16(b.p) || b.sw % 4 || b.sh % 4 || (b.n > 1 && b.sn % 4)) return false;
	if ((long)a.h * a.sh >= 0x7fffffffL) return false; // 32-bit offsets within an image
	if (((long)(b.h - 1) * b.sh + (long)(b.w - 1) * b.sw + b.c) * 4 >= 0x7ffff000L) return false; // one image within a buffer descriptor's range
	return (long)b.n * b.h * ((b.w + 15) / 16) < 0x7fffffffL;
}
static void conv_c3_args(const conv_geom_t& g, const Image4& a, const float* w, const float* bias, const Image4& b, ConvC3Args* c)
{
	c->a = a.p; c->w = w; c->bias = bias; c->b = b.p;
	c->a_sn = a.sn; c->a_sh = a.sh; c->b_sn = b.sn; c->b_sh = b.sh; c->b_sw = b.sw;
	c->N = g.N; c->H = g.H; c->W = g.W; c->OH = g.OH; c->OW = g.OW; c->K = g.K; c->pad_y = g.pby; c->pad_x = g.pbx; c->sy = g.sy; c->sx = g.sx;
	c->groups_per_row = (g.OW + 15) / 16; c->groups = g.N * g.OH * c->groups_per_row;
	c->d_gpr.init(c->groups_per_row); c->d_oh.init(g.OH);
	c->b_image_bytes = (unsigned)((((long)b.h - 1) * b.sh + ((long)b.w - 1) * b.sw + b.c) * 4);
}
static int conv_c3_forw(const conv_geom_t& g, const Image4& a, const float* w, const float* bias, const Image4& b, ccv_nnc_stream_context_t* const ctx)
{
	ConvC3Args c;
	conv_c3_args(g, a, w, bias, b, &c);
	c.relu = tl_relu_want ? 1 : 0;
	if (c.relu) tl_relu_done = 1;
	hipStream_t stream = stream_of(ctx);
	const long want = ((long)c.groups + 3) / 4, cap = (long)device_cu_count() * 8;
	const unsigned grid = (unsigned)(want < cap ? want : cap);
	note_kernel("conv_fwd_c3");
	ProfScope prof("conv_fwd_c3|nnc::conv3x3_c3_fwd_kernel", 2.0 * g.N * g.OH * g.OW * (double)g.K * 27, 0, g.N * g.OH * g.OW, g.K, 27, 1, 1, stream);
	if (g.K == 64) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_c3_fwd_kernel<4>), dim3(grid), dim3(256), 0, stream, c);
	else if (g.K == 32) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_c3_fwd_kernel<2>), dim3(grid), dim3(256), 0, stream, c);
	else hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_c3_fwd_kernel<1>), dim3(grid), dim3(256), 0, stream, c);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}
static size_t conv_c3_wgrad_scratch_bound(const int K) { return sizeof(float) * (size_t)device_cu_count() * 4 * 4 * K * 32; } // per-wave partials at the grid cap
// dw (+)= and dbias (+)= in one pass over the output gradient
static int conv_c3_wgrad(const conv_geom_t& g, const Image4& gr, const Image4& a, float* dw, float* dbias, const int flags, ccv_nnc_stream_context_t* const ctx)
{
	ConvC3Args c;
	conv_c3_args(g, a, 0, 0, gr, &c);
	const long want = ((long)c.groups + 3) / 4, cap = (long)device_cu_count() * 4;
	const unsigned grid = (unsigned)(want < cap ? want : cap);
	const int waves = (int)grid * 4;
	float* const part = (float*)workspace_of(ctx, sizeof(float) * (size_t)waves * g.K * 32);
	if (!part) return CCV_NNC_EXEC_OOM;
	hipStream_t stream = stream_of(ctx);
	note_kernel("conv_wgrad_c3");
	{
		ProfScope prof("conv_wgrad_c3|nnc::conv3x3_c3_wgrad_kernel", 2.0 * g.N * g.OH * g.OW * (double)g.K * 27, 0, g.K, 27, g.N * g.OH * g.OW, 1, 1, stream);
		if (g.K == 64) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_c3_wgrad_kernel<4>), dim3(grid), dim3(256), 0, stream, c, part);
		else if (g.K == 32) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_c3_wgrad_kernel<2>), dim3(grid), dim3(256), 0, stream, c, part);
		else hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x3_c3_wgrad_kernel<1>), dim3(grid), dim3(256), 0, stream, c, part);
	}
	HIP_ENFORCE(hipGetLastError());
	hipLaunchKernelGGL(convc3_wgrad_fold, dim3((unsigned)g.K), dim3(256), 0, stream, (const float*)part, (int)grid /* one partial per workgroup */, g.K, dw, dbias, (flags & CCV_NNC_ACCUMULATE_OUTPUT) ? 1 : 0);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}

static int conv_forw_nhwc(const conv_geom_t& g, const Image4& a, const float* w, const float* bias, const Image4& b, const int algo, const int flags, ccv_nnc_stream_context_t* const ctx)
{
	if (a.sc != 1 || !pixel_linear(b) || !image_fits_int(a)) return CCV_NNC_EXEC_INVALID;
	if (algo != CONV_ALGO_IMPLICIT_GEMM && conv_c3_ok(g, a, b)) return conv_c3_forw(g, a, w, bias, b, ctx);
	wino_plan_t wp;
	wino_fused_plan_t fp;
	if (algo != CONV_ALGO_IMPLICIT_GEMM && algo != CONV_ALGO_WINOGRAD && g.pby <= 2 && g.pbx <= 2 && g.pby >= 0 && g.pbx >= 0 && wino_fused_plan(g, a, b, &fp) && (algo == CONV_ALGO_WINOGRAD_FUSED || wino_fused_preferred(g.C, fp, b))) {
		const int r = conv_wino_fused_run<false>("conv_fwd_wino_fused", g, fp, a, w, bias, b, g.pby, g.pbx, ctx);
		if (r != CCV_NNC_EXEC_OOM) return r;
	}
	if (algo != CONV_ALGO_IMPLICIT_GEMM && wino_plan(g, g.OH, g.OW, g.C, g.K, &wp) && wino_images_ok(a, b, w, bias) && (algo >= CONV_ALGO_WINOGRAD || wino_preferred(wp, g.C, g.K))) {
		const int r = conv_wino_run<false>("conv_fwd_wino", g, wp, a, w, bias, b, g.pby, g.pbx, flags, ctx);
		if (r != CCV_NNC_EXEC_OOM) return r; // the transformed images did not fit the device: the implicit GEMM needs no such scratch
	}
	const long M = (long)g.N * g.OH * g.OW;
	const int Kred = g.kh * g.kw * g.Cg;
	if (M > 0x7fffffffL) return CCV_NNC_EXEC_INVALID;
	const bool vec = (g.Cg % 4 == 0) && aligned16(a.p) && aligned16(w) && a.sw % 4 == 0 && a.sh % 4 == 0 && (a.n == 1 || a.sn % 4 == 0);
	GemmOut out = { b.p, b.sw, 1, bias, 1.f, 0 };
	if (vec && conv_pointwise(g, a, b)) { // b [pixels][K] = a [pixels][C] . w [K][C]^T
		MatLoader<true, true> la, lb;
		la.p = a.p; la.ldr = a.sw; la.ldk = 1; la.R = (int)M; la.K = g.C;
		lb.p = w; lb.ldr = g.C; lb.ldk = 1; lb.R = g.K; lb.K = g.C;
		return gemm_run("conv_fwd_pointwise", la, lb, out, (int)M, g.K, g.C, 1, 0L, 0L, 0L, 0L, 1, flags, ctx);
	}
	KOrder ko; // taps of one 32-channel chunk in consecutive K-steps (L2 reuse of the re-read pixels), see mfma_gemm.h
	if (g.kh * g.kw > 1 && g.Cg % GEMM_BK == 0) ko.init(g.kh * g.kw, g.Cg);
#define CONV_FWD(VEC, INC) do { \
		Im2colKC<VEC, false, INC> la; \
		la.p = a.p; la.s_n = a.sn; la.s_h = (int)a.sh; la.s_w = (int)a.sw; la.H = g.H; la.W = g.W; \
		la.OW = g.OW; la.OHW = g.OH * g.OW; la.M = (int)M; la.C = g.Cg; la.KWC = g.kw * g.Cg; la.K = Kred; \
		la.my = g.sy; la.mx = g.sx; la.oy_off = -g.pby; la.ox_off = -g.pbx; la.ty = g.dy; la.tx = g.dx; la.dv_y = 1; la.dv_x = 1; \
		MatLoader<true, VEC> lb; \
		lb.p = w; lb.ldr = Kred; lb.ldk = 1; lb.R = g.Kg; lb.K = Kred; \
		return gemm_run("conv_fwd", la, lb, out, (int)M, g.Kg, Kred, g.groups, (long)g.Cg, (long)g.Kg * Kred, (long)g.Kg, (long)g.Kg, 1, flags, ctx, ko); \
	} while (0)
	// (INC = the division-free incremental k state of mfma_gemm.h: measured SLOWER on MI355X -- it trades ~56 quarter-rate
	// multiplies per K-step for ~80 more selects / 64-bit adds, and what the K-loop pays for is instruction COUNT.  Kept as
	// a template switch, not instantiated.)
	if (vec) CONV_FWD(true, false);
	else CONV_FWD(false, false);
#undef CONV_FWD
}

// h = sum_{k,i,j} g[n, (y+p-i*d)/s, (x+p-j*d)/s, k] * w[k,i,j,c]
// ---- data gradient of a stride-2 convolution by parity classes ------------------------------------------------------------------
// h[y, x] = sum over the taps (ky, kx) with (y + pad - ky, x + pad - kx) BOTH even of g[(y + pad - ky) / 2, (x + pad - kx) / 2] . w[ky, kx]:
// for a given parity (py, px) of (y, x) only the taps with ky = py + pad, kx = px + pad (mod 2) exist -- a quarter of them on
// average, 1 + 2 + 2 + 4 = 9 of the 4 x 9 the strided im2col walk tests for a 3 x 3 filter (which ran the ResNet-50 stage
// transitions at 112 TFLOP/s of mostly-zero work, 28 effective).  Per class the sum is a dense STRIDE-1 correlation of g with the
// class's sub-filter (taps in descending ky: tap i reads g[u - pb + i], pb = (ky_max - py - pad) / 2), i.e. the forward implicit
// GEMM with the roles of the channel axes swapped; its output -- the class's positions (2u + py, 2v + px) -- is computed into a dense
// quarter-size image and interleaved into h.
static __global__ void __launch_bounds__(256) parity_filter_kernel(const float* __restrict__ w, float* __restrict__ f, const int K, const int C, const int kh, const int kw, const int ny, const int nx, const int ky_max, const int kx_max, const size_t total)
{ // f[c][i][j][k] = w[k][ky_max - 2 i][kx_max - 2 j][c]
	for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
		size_t r = idx;
		const int k = (int)(r % K); r /= K;
		const int j = (int)(r % nx); r /= nx;
		const int i = (int)(r % ny); r /= ny;
		const int c = (int)r;
		f[idx] = w[(((size_t)k * kh + (ky_max - 2 * i)) * kw + (kx_max - 2 * j)) * C + c];
	}
}
static __global__ void __launch_bounds__(256) parity_scatter_kernel(const float* __restrict__ t, float* __restrict__ h, const int U, const int V, const int C4, const long h_sn, const long h_sh, const long h_sw, const int py, const int px, const size_t total)
{ // t: dense [N][U][V][C]; h[n][2u + py][2v + px][:] = t[n][u][v][:]   (C4 = C / 4 float4 groups, or C scalars when C4 < 0)
	const int cn = C4 < 0 ? -C4 : C4;
	for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
		size_t r = idx;
		const int c = (int)(r % cn); r /= cn;
		const int v = (int)(r % V); r /= V;
		const int u = (int)(r % U); r /= U;
		const long o = (long)r * h_sn + (long)(2 * u + py) * h_sh + (long)(2 * v + px) * h_sw;
		if (C4 < 0) h[o + c] = t[idx];
		else *(float4*)(h + o + 4 * c) = ((const float4*)t)[idx];
	}
}
static bool conv_dgrad_parity_ok(const conv_geom_t& g)
{
	return g.sy == 2 && g.sx == 2 && g.dy == 1 && g.dx == 1 && g.groups == 1 && g.kh >= 1 && g.kw >= 1 && g.pby >= 0 && g.pbx >= 0 && g.pby < g.kh && g.pbx < g.kw && g.H >= 2 && g.W >= 2;
}
static size_t conv_dgrad_parity_prefix(const conv_geom_t& g)
{ // the quarter image + the largest sub-filter
	const size_t U = (g.H + 1) / 2, V = (g.W + 1) / 2;
	return align256(sizeof(float) * (size_t)g.N * U * V * g.C) + align256(sizeof(float) * (size_t)((g.kh + 1) / 2) * ((g.kw + 1) / 2) * g.C * g.K);
}
static size_t conv_dgrad_parity_scratch(const conv_geom_t& g)
{
	if (!conv_dgrad_parity_ok(g)) return 0;
	return conv_dgrad_parity_prefix(g) + gemm_workspace_bound((long)g.N * ((g.H + 1) / 2) * ((g.W + 1) / 2), g.C, (long)((g.kh + 1) / 2) * ((g.kw + 1) / 2) * g.K) + 256;
}
static int conv_forw_nhwc(const conv_geom_t& g, const Image4& a, const float* w, const float* bias, const Image4& b, const int algo, const int flags, ccv_nnc_stream_context_t* const ctx);
static int conv_dgrad_parity(const conv_geom_t& g, const Image4& gr, const float* w, const Image4& h, const int flags, ccv_nnc_stream_context_t* const ctx)
{
	const int U0 = (g.H + 1) / 2, V0 = (g.W + 1) / 2;
	const size_t tbytes = align256(sizeof(float) * (size_t)g.N * U0 * V0 * g.C);
	WorkspaceScope ws(ctx, conv_dgrad_parity_prefix(g), gemm_workspace_bound((long)g.N * U0 * V0, g.C, (long)((g.kh + 1) / 2) * ((g.kw + 1) / 2) * g.K));
	char* const p = (char*)ws.prefix();
	if (!p) return CCV_NNC_EXEC_OOM;
	float* const T = (float*)p;
	float* const Fw = (float*)(p + tbytes);
	hipStream_t stream = stream_of(ctx);
	const bool vec = g.C % 4 == 0 && aligned16(h.p) && h.sn % 4 == 0 && h.sh % 4 == 0 && h.sw % 4 == 0;
	for (int py = 0; py < 2; py++)
		for (int px = 0; px < 2; px++) {
			const int U = (g.H - py + 1) / 2, V = (g.W - px + 1) / 2; // positions y = 2u + py < H
			if (U <= 0 || V <= 0) continue;
			// taps of this class: ky = (py + pad) mod 2, + 2, ... < kh
			const int ky0 = (py + g.pby) & 1, kx0 = (px + g.pbx) & 1;
			const int ny = ky0 < g.kh ? (g.kh - ky0 + 1) / 2 : 0, nx = kx0 < g.kw ? (g.kw - kx0 + 1) / 2 : 0;
			const size_t cells = (size_t)g.N * U * V * (vec ? g.C / 4 : g.C);
			if (ny == 0 || nx == 0) { // no tap reaches these positions: zero gradient
				HIP_ENFORCE(hipMemsetAsync(T, 0, sizeof(float) * (size_t)g.N * U * V * g.C, stream));
			} else {
				const int ky_max = ky0 + 2 * (ny - 1), kx_max = kx0 + 2 * (nx - 1);
				const int pb_y = (ky_max - py - g.pby) / 2, pb_x = (kx_max - px - g.pbx) / 2;
				if (ky_max - py - g.pby < 0 || kx_max - px - g.pbx < 0) return CCV_NNC_EXEC_NO_KERNEL;
				const size_t fn = (size_t)g.C * ny * nx * g.K;
				hipLaunchKernelGGL(parity_filter_kernel, dim3(grid_for(fn, 256)), dim3(256), 0, stream, w, Fw, g.K, g.C, g.kh, g.kw, ny, nx, ky_max, kx_max, fn);
				HIP_ENFORCE(hipGetLastError());
				conv_geom_t q;
				q.N = g.N; q.H = g.OH; q.W = g.OW; q.C = g.K; q.OH = U; q.OW = V; q.K = g.C;
				q.kh = ny; q.kw = nx; q.Cg = g.K; q.Kg = g.C; q.groups = 1; q.sy = 1; q.sx = 1; q.pby = pb_y; q.pbx = pb_x; q.dy = 1; q.dx = 1;
				Image4 ti;
				ti.p = T; ti.n = g.N; ti.h = U; ti.w = V; ti.c = g.C; ti.sc = 1; ti.sw = g.C; ti.sh = (long)V * g.C; ti.sn = (long)U * V * g.C;
				const int r = conv_forw_nhwc(q, gr, Fw, 0, ti, CONV_ALGO_IMPLICIT_GEMM, flags & ~CCV_NNC_ACCUMULATE_OUTPUT, ctx);
				if (r != CCV_NNC_EXEC_SUCCESS) return r;
			}
			hipLaunchKernelGGL(parity_scatter_kernel, dim3(grid_for(cells, 256)), dim3(256), 0, stream, (const float*)T, h.p, U, V, vec ? g.C / 4 : -g.C, h.sn, h.sh, h.sw, py, px, cells);
			HIP_ENFORCE(hipGetLastError());
		}
	return CCV_NNC_EXEC_SUCCESS;
}

static int conv_dgrad_nhwc(const conv_geom_t& g, const Image4& gr, const float* w, const Image4& h, const int algo, const int flags, ccv_nnc_stream_context_t* const ctx)
{
	if (gr.sc != 1 || !pixel_linear(h) || !image_fits_int(gr)) return CCV_NNC_EXEC_INVALID;
	wino_plan_t wp;
	wino_fused_plan_t fp;
	if (algo != CONV_ALGO_IMPLICIT_GEMM && algo != CONV_ALGO_WINOGRAD && g.pby <= 2 && g.pbx <= 2 && g.pby >= 0 && g.pbx >= 0 && wino_fused_plan(g, gr, h, &fp) && (algo == CONV_ALGO_WINOGRAD_FUSED || wino_fused_preferred(g.K, fp, h))) {
		const int r = conv_wino_fused_run<true>("conv_dgrad_wino_fused", g, fp, gr, w, 0, h, 2 - g.pby, 2 - g.pbx, ctx);
		if (r != CCV_NNC_EXEC_OOM) return r;
	}
	if (algo != CONV_ALGO_IMPLICIT_GEMM && wino_plan(g, g.H, g.W, g.K, g.C, &wp) && wino_images_ok(gr, h, w, 0) && (algo >= CONV_ALGO_WINOGRAD || wino_preferred(wp, g.K, g.C))) {
		const int r = conv_wino_run<true>("conv_dgrad_wino", g, wp, gr, w, 0, h, 2 - g.pby, 2 - g.pbx, flags, ctx);
		if (r != CCV_NNC_EXEC_OOM) return r;
	}
	if (algo != CONV_ALGO_IMPLICIT_GEMM && !(flags & CCV_NNC_ACCUMULATE_OUTPUT) && conv_dgrad_parity_ok(g) && g.kh * g.kw > 1) {
		const int r = conv_dgrad_parity(g, gr, w, h, flags, ctx);
		if (r != CCV_NNC_EXEC_OOM && r != CCV_NNC_EXEC_NO_KERNEL) return r; // (no room under a staging scope / odd borders: the strided walk below)
	}
	const long M = (long)g.N * g.H * g.W;
	const int Kred = g.kh * g.kw * g.Kg;
	if (M > 0x7fffffffL) return CCV_NNC_EXEC_INVALID;
	const bool vec = (g.Kg % 4 == 0) && (g.Cg % 4 == 0) && aligned16(gr.p) && aligned16(w) && gr.sw % 4 == 0 && gr.sh % 4 == 0 && (gr.n == 1 || gr.sn % 4 == 0);
	GemmOut out = { h.p, h.sw, 1, 0, 1.f, 0 };
	if (vec && conv_pointwise(g, h, gr)) { // h [pixels][C] = g [pixels][K] . w [K][C]: B(c, k) = w[k C + c], rows contiguous
		MatLoader<true, true> la;
		la.p = gr.p; la.ldr = gr.sw; la.ldk = 1; la.R = (int)M; la.K = g.K;
		MatLoader<false, true> lb;
		lb.p = w; lb.ldr = 1; lb.ldk = g.C; lb.R = g.C; lb.K = g.K;
		return gemm_run("conv_dgrad_pointwise", la, lb, out, (int)M, g.C, g.K, 1, 0L, 0L, 0L, 0L, 1, flags, ctx);
	}
	KOrder ko;
	if (g.kh * g.kw > 1 && g.Kg % GEMM_BK == 0) ko.init(g.kh * g.kw, g.Kg);
#define CONV_DGRAD(VEC, STRIDED, INC) do { \
		Im2colKC<VEC, STRIDED, INC> la; \
		la.p = gr.p; la.s_n = gr.sn; la.s_h = (int)gr.sh; la.s_w = (int)gr.sw; la.H = g.OH; la.W = g.OW; \
		la.OW = g.W; la.OHW = g.H * g.W; la.M = (int)M; la.C = g.Kg; la.KWC = g.kw * g.Kg; la.K = Kred; \
		la.my = 1; la.mx = 1; la.oy_off = g.pby; la.ox_off = g.pbx; la.ty = -g.dy; la.tx = -g.dx; la.dv_y = g.sy; la.dv_x = g.sx; \
		WgtDgradNC<VEC, INC> lb; \
		lb.p = w; lb.ko_stride = (long)g.kh * g.kw * g.Cg; lb.C = g.Cg; lb.Ko = g.Kg; lb.K = Kred; \
		return gemm_run("conv_dgrad", la, lb, out, (int)M, g.Cg, Kred, g.groups, (long)g.Kg, (long)g.Kg * g.kh * g.kw * g.Cg, (long)g.Cg, 0L, 1, flags, ctx, ko); \
	} while (0)
	if (g.sy != 1 || g.sx != 1) { if (vec) CONV_DGRAD(true, true, false); else CONV_DGRAD(false, true, false); }
	else { if (vec) CONV_DGRAD(true, false, false); else CONV_DGRAD(false, false, false); }
#undef CONV_DGRAD
}

// dw[k,i,j,c] (+)= sum_{n,y,x} g[n,y,x,k] * a[n, y*s-p+i*d, x*s-p+j*d, c]
static int conv_wgrad_nhwc(const conv_geom_t& g, const Image4& gr, const Image4& a, float* dw, float* dbias, bool* bias_done, const int algo, const int flags, ccv_nnc_stream_context_t* const ctx)
{
	if (bias_done) *bias_done = false;
	if (a.sc != 1 || !pixel_linear(gr) || !image_fits_int(a)) return CCV_NNC_EXEC_INVALID;
	if (algo != CONV_ALGO_IMPLICIT_GEMM && conv_c3_ok(g, a, gr)) {
		const int r = conv_c3_wgrad(g, gr, a, dw, dbias, flags, ctx);
		if (r == CCV_NNC_EXEC_SUCCESS && bias_done) *bias_done = dbias != 0;
		if (r != CCV_NNC_EXEC_OOM) return r;
	}
	wino_wgrad_fused_plan_t wfp;
	// (algorithm 2 = "the fused kernels" takes the fused filter gradient under the same rule as algorithm -1: the host's autotuner times the WHOLE backward
	// command per algorithm, and a 256-channel layer wants the fused data gradient next to the via-HBM filter gradient)
	if (algo != CONV_ALGO_IMPLICIT_GEMM && algo != CONV_ALGO_WINOGRAD && wino_wgrad_fused_plan(g, &wfp) && wino_wgrad_fused_images_ok(a, gr, dw) && wino_wgrad_fused_preferred(g)) {
		const int r = conv_wino_wgrad_fused(g, wfp, gr, a, dw, dbias, bias_done, flags, ctx);
		if (r != CCV_NNC_EXEC_OOM) return r;
	}
	wino_wgrad_plan_t wp;
	if (algo != CONV_ALGO_IMPLICIT_GEMM && wino_wgrad_plan(g, &wp) && wino_images_ok(a, gr, dw, 0) && (algo >= CONV_ALGO_WINOGRAD || wino_preferred(wp.t, g.C, g.K))) {
		const int r = conv_wino_wgrad(g, wp, gr, a, dw, dbias, bias_done, flags, ctx);
		if (r != CCV_NNC_EXEC_OOM) return r;
	}
	const long P = (long)g.N * g.OH * g.OW;
	if (P > 0x7fffffffL) return CCV_NNC_EXEC_INVALID;
	const int NN = g.kh * g.kw * g.Cg;
	const bool vec = (g.Cg % 4 == 0) && (g.Kg % 4 == 0) && aligned16(a.p) && aligned16(gr.p) && a.sw % 4 == 0 && a.sh % 4 == 0 && (a.n == 1 || a.sn % 4 == 0) && gr.sw % 4 == 0;
	GemmOut out = { dw, (long)NN, 1, 0, 1.f, (flags & CCV_NNC_ACCUMULATE_OUTPUT) ? 1 : 0 };
	if (vec && conv_pointwise(g, a, gr)) { // dw [K][C] = g^T [K][pixels] . a [pixels][C]: both operands with their rows (k / c) contiguous, the reduction over the pixels
		MatLoader<false, true> la, lb;
		la.p = gr.p; la.ldr = 1; la.ldk = gr.sw; la.R = g.K; la.K = (int)P;
		lb.p = a.p; lb.ldr = 1; lb.ldk = a.sw; lb.R = g.C; lb.K = (int)P;
		return gemm_run("conv_wgrad_pointwise", la, lb, out, g.K, g.C, (int)P, 1, 0L, 0L, 0L, 0L, 0, flags, ctx);
	}
#define CONV_WGRAD(VEC, INC) do { \
		MatLoader<false, VEC> la; \
		la.p = gr.p; la.ldr = 1; la.ldk = gr.sw; la.R = g.Kg; la.K = (int)P; \
		Im2colNC<VEC, INC> lb; \
		lb.p = a.p; lb.s_n = a.sn; lb.s_h = (int)a.sh; lb.s_w = (int)a.sw; lb.H = g.H; lb.W = g.W; lb.OW = g.OW; lb.OHW = g.OH * g.OW; \
		lb.C = g.Cg; lb.KWC = g.kw * g.Cg; lb.NN = NN; lb.K = (int)P; lb.sy = g.sy; lb.sx = g.sx; lb.py = g.pby; lb.px = g.pbx; lb.dy = g.dy; lb.dx = g.dx; \
		return gemm_run("conv_wgrad", la, lb, out, g.Kg, NN, (int)P, g.groups, (long)g.Kg, (long)g.Cg, (long)g.Kg * NN, 0L, 0, flags, ctx); \
	} while (0)
	if (vec) CONV_WGRAD(true, false);
	else CONV_WGRAD(false, false);
#undef CONV_WGRAD
}

// ---- 1x1 convolutions on NCHW tensors: plain batched GEMMs over the tensors WHERE THEY LIE ------------------------------------
// The reference's ResNet trainer keeps NCHW (bin/nnc/imagenet.c:354) and two thirds of its convolutions are 1x1, stride 1: with
// P = H * W,   forward  b_n [K x P] = w [K x C] . a_n [C x P]          one GEMM per image (grid z = N), output rows contiguous
//              dgrad    h_n [C x P] = w^T [C x K] . g_n [K x P]
//              wgrad    dw [K x C]  = sum_n g_n [K x P] . a_n^T [P x C]  one GEMM with the reduction running over (n, p): PlaneKC
//              dbias    [K]         = sum over planes of g              (chan_sum, cmd_norm.cpp)
// -- no layout pass at all, where the general NCHW route transposes the input, the output and the weights around an NHWC
// kernel (measured on the ResNet-50 step at batch 256: transposes were 80 of 167 ms).  Needs P % 4 == 0 and C % 4 == 0
// (16-byte chunks; 8-byte for halves) and dense tensors; 7 x 7 maps (P = 49) take the general route.  T = float or half_t.
template <class T> struct conv1x1_types;
template <> struct conv1x1_types<float> { typedef GemmOut Out; };
template <> struct conv1x1_types<half_t> { typedef GemmOutH Out; };
template <class LA, class LB> static int conv1x1_run(const char* name, float*, LA la, LB lb, const GemmOut out, int M, int N, int K, int z, long a_z, long b_z, long c_z, int splits, int flags, ccv_nnc_stream_context_t* ctx) { return gemm_run(name, la, lb, out, M, N, K, z, a_z, b_z, c_z, 0L, splits, flags, ctx); }
template <class LA, class LB> static int conv1x1_run(const char* name, half_t*, LA la, LB lb, const GemmOutH out, int M, int N, int K, int z, long a_z, long b_z, long c_z, int splits, int flags, ccv_nnc_stream_context_t* ctx) { return gemm_run_h(name, la, lb, out, M, N, K, z, a_z, b_z, c_z, 0L, splits, flags, ctx); }

static bool nchw_dense(const ccv_nnc_tensor_t* t, int* N, int* C, int* P)
{
	if (t->info.format != CCV_TENSOR_FORMAT_NCHW || !tensor_contiguous(t)) return false;
	const int nd = tensor_nd(t->info.dim);
	if (nd != 3 && nd != 4) return false;
	const int b = nd == 4;
	*N = b ? t->info.dim[0] : 1; *C = t->info.dim[b]; *P = t->info.dim[b + 1] * t->info.dim[b + 2];
	return true;
}
static bool conv1x1_cmd_ok(const ccv_nnc_cmd_t& cmd, const ccv_nnc_hint_t& hint)
{
	if (cmd.info.size.dim[0] != 1 || cmd.info.size.dim[1] != 1 || cmd.info.convolution.groups > 1) return false;
	for (int i = 0; i < 2; i++)
		if ((hint.stride.dim[i] > 1) || hint.border.begin[i] != 0 || hint.border.end[i] != 0) return false;
	return true;
}
template <class T> static bool chunk_aligned(const void* p) { return (((uintptr_t)p) & (4 * sizeof(T) - 1)) == 0; }

// CCV_NNC_EXEC_NO_KERNEL = not this path
template <class T>
static int conv1x1_nchw_forw(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, const ccv_nnc_tensor_t* a, const ccv_nnc_tensor_t* w, const ccv_nnc_tensor_t* bias, ccv_nnc_tensor_t* b, ccv_nnc_stream_context_t* const ctx)
{
	typedef typename conv1x1_types<T>::Out Out;
	int N, C, P, Nb, K, Pb;
	if (!conv1x1_cmd_ok(cmd, hint) || !nchw_dense(a, &N, &C, &P) || !nchw_dense(b, &Nb, &K, &Pb) || N != Nb || P != Pb) return CCV_NNC_EXEC_NO_KERNEL;
	if (K != cmd.info.convolution.count || !tensor_contiguous(w) || (long)tensor_count(w->info) != (long)K * C || (bias && (!tensor_contiguous(bias) || (int)tensor_count(bias->info) != K))) return CCV_NNC_EXEC_NO_KERNEL;
	if (P % 4 || C % 4 || ((long)C * P) % 4 || !chunk_aligned<T>(a->data.u8) || !chunk_aligned<T>(w->data.u8) || N <= 0 || P <= 0) return CCV_NNC_EXEC_NO_KERNEL;
	MatLoader<true, true> la;  // w [K][C]: reduction-contiguous rows
	la.p = (const float*)w->data.u8; la.ldr = C; la.ldk = 1; la.R = K; la.K = C;
	MatLoader<false, true> lb; // a_n [C][P]: output columns p contiguous, reduction index c strides by P
	lb.p = (const float*)a->data.u8; lb.ldr = 1; lb.ldk = P; lb.R = P; lb.K = C;
	Out out = { (T*)b->data.u8, (long)P, 1, bias ? (const T*)bias->data.u8 : 0, 1.f, 0, 1 };
	out.bias_ldn = 0; // one bias per output ROW (= output channel)
	return conv1x1_run("conv1x1_nchw_fwd", (T*)0, la, lb, out, K, P, C, N, 0L, (long)C * P, (long)K * P, 1, flags, ctx);
}

template <class T>
static int conv1x1_nchw_back(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, const ccv_nnc_tensor_t* g, const ccv_nnc_tensor_t* a, const ccv_nnc_tensor_t* w, ccv_nnc_tensor_t* h, ccv_nnc_tensor_t* dw, ccv_nnc_tensor_t* dbias, ccv_nnc_stream_context_t* const ctx)
{
	typedef typename conv1x1_types<T>::Out Out;
	int N, K, P;
	if (!conv1x1_cmd_ok(cmd, hint) || !nchw_dense(g, &N, &K, &P) || K != cmd.info.convolution.count) return CCV_NNC_EXEC_NO_KERNEL;
	const ccv_nnc_tensor_t* shape_src = a ? a : h;
	int Na, C, Pa;
	if (!shape_src || !nchw_dense(shape_src, &Na, &C, &Pa) || Na != N || Pa != P) return CCV_NNC_EXEC_NO_KERNEL;
	if (h && (!nchw_dense(h, &Na, &C, &Pa) || Na != N || Pa != P || !w || !tensor_contiguous(w) || (long)tensor_count(w->info) != (long)K * C)) return CCV_NNC_EXEC_NO_KERNEL;
	if (dw && (!a || !tensor_contiguous(dw) || (long)tensor_count(dw->info) != (long)K * C)) return CCV_NNC_EXEC_NO_KERNEL;
	if (dbias && (!tensor_contiguous(dbias) || (int)tensor_count(dbias->info) != K)) return CCV_NNC_EXEC_NO_KERNEL;
	if (P % 4 || C % 4 || K % 4 || !chunk_aligned<T>(g->data.u8) || (a && !chunk_aligned<T>(a->data.u8)) || (w && !chunk_aligned<T>(w->data.u8)) || (long)N * P > 0x7fffffffL) return CCV_NNC_EXEC_NO_KERNEL;
	const int acc = (flags & CCV_NNC_ACCUMULATE_OUTPUT) ? 1 : 0;
	int ret;
	if (dw) { // dw [K][C] = sum over (n, p): both operands rows of planes
		PlaneKC<true> la, lb;
		la.p = (const float*)g->data.u8; la.ldr = P; la.s_n = (long)K * P; la.R = K; la.K = N * P; la.P = P;
		lb.p = (const float*)a->data.u8; lb.ldr = P; lb.s_n = (long)C * P; lb.R = C; lb.K = N * P; lb.P = P;
		Out out = { (T*)dw->data.u8, (long)C, 1, 0, 1.f, acc, 0 };
		if ((ret = conv1x1_run("conv1x1_nchw_wgrad", (T*)0, la, lb, out, K, C, N * P, 1, 0L, 0L, 0L, 0, flags, ctx)) != CCV_NNC_EXEC_SUCCESS) return ret;
	}
	if (dbias && (ret = (sizeof(T) == sizeof(float) ? chan_sum_planes(g->data.f32, N, K, P, dbias->data.f32, acc, ctx) : chan_sum_planes_f16(g->data.u8, N, K, P, dbias->data.u8, acc, ctx))) != CCV_NNC_EXEC_SUCCESS) return ret;
	if (h) { // h_n [C][P] = w^T . g_n
		MatLoader<false, true> la; // w viewed [C rows][K reduction]: rows contiguous (element (c, k) = w[k * C + c])
		la.p = (const float*)w->data.u8; la.ldr = 1; la.ldk = C; la.R = C; la.K = K;
		MatLoader<false, true> lb;
		lb.p = (const float*)g->data.u8; lb.ldr = 1; lb.ldk = P; lb.R = P; lb.K = K;
		Out out = { (T*)h->data.u8, (long)P, 1, 0, 1.f, 0, 0 };
		if ((ret = conv1x1_run("conv1x1_nchw_dgrad", (T*)0, la, lb, out, C, P, K, N, 0L, (long)K * P, (long)C * P, 1, flags, ctx)) != CCV_NNC_EXEC_SUCCESS) return ret;
	}
	return CCV_NNC_EXEC_SUCCESS;
}

// ---- layout staging ---------------------------------------------------------------------------------------------------
// The kernels read NHWC activations and [K][kh][kw][Cg] weights.  NCHW activations (the reference's ResNet trainer) and
// NCHW-format weights [K][Cg][kh][kw] (what the reference's GPU tests hand over, test/int/nnc/cudnn.tests.c:50,65) are
// re-laid-out through the stream workspace by the tiled transpose of cmd_util.cpp: one extra read + write of the tensor,
// HBM-bound, instead of a second family of gather kernels whose channel-strided loads could not be 16-byte vectors.

static bool weights_shape(const ccv_nnc_tensor_t* w, int* K, int* kh, int* kw, int* Cg)
{
	if (!w || tensor_nd(w->info.dim) != 4 || !tensor_contiguous(w)) return false;
	const int* d = w->info.dim;
	if (w->info.format == CCV_TENSOR_FORMAT_NCHW) { *K = d[0]; *Cg = d[1]; *kh = d[2]; *kw = d[3]; }
	else if (w->info.format == CCV_TENSOR_FORMAT_NHWC) { *K = d[0]; *kh = d[1]; *kw = d[2]; *Cg = d[3]; }
	else return false;
	return true;
}

// A dense NHWC tensor header over `data` with the logical shape of `like` (which is NCHW or NHWC, 3-d or 4-d).
static void dense_nhwc_like(const ccv_nnc_tensor_t* like, const Image4& li, float* data, ccv_nnc_tensor_t* out)
{
	memset(out, 0, sizeof(*out));
	out->type = like->info.type & ~CCV_TENSOR_VIEW;
	out->info = like->info;
	out->info.format = CCV_TENSOR_FORMAT_NHWC;
	const int b = tensor_nd(like->info.dim) == 4;
	memset(out->info.dim, 0, sizeof(out->info.dim));
	if (b) out->info.dim[0] = li.n;
	out->info.dim[b] = li.h; out->info.dim[b + 1] = li.w; out->info.dim[b + 2] = li.c;
	out->data.f32 = data;
}

static int _conv_forw(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	if (input_size < 2 || output_size < 1 || !inputs[0] || !inputs[1] || !outputs[0]) return CCV_NNC_EXEC_INVALID;
	const ccv_nnc_tensor_t* a = inputs[0];
	const ccv_nnc_tensor_t* w = inputs[1];
	const ccv_nnc_tensor_t* bias = input_size > 2 ? inputs[2] : 0;
	ccv_nnc_tensor_t* b = outputs[0];
	if (CCV_GET_DATA_TYPE(a->info.datatype) != CCV_32F) return CCV_NNC_EXEC_INVALID;
	if (a->info.format != b->info.format) return CCV_NNC_EXEC_INVALID;
	if (a->info.format == CCV_TENSOR_FORMAT_NCHW && cmd.algorithm != CONV_ALGO_IMPLICIT_GEMM) {
		const int r = conv1x1_nchw_forw<float>(cmd, hint, flags, a, w, bias, b, stream_context);
		if (r != CCV_NNC_EXEC_NO_KERNEL) return r;
	}
	Image4 ai, bi;
	if (!image4(a, &ai) || !image4(b, &bi)) return CCV_NNC_EXEC_INVALID;
	int K, kh, kw, Cg;
	if (!weights_shape(w, &K, &kh, &kw, &Cg)) return CCV_NNC_EXEC_INVALID;
	conv_geom_t g;
	if (!conv_geometry(cmd, hint, ai, bi, 0, &g) || K != g.K || kh != g.kh || kw != g.kw || Cg != g.Cg) return CCV_NNC_EXEC_INVALID;
	if (bias && (bias->info.dim[0] != g.K || !tensor_contiguous(bias))) return CCV_NNC_EXEC_INVALID;
	const bool stage_io = a->info.format == CCV_TENSOR_FORMAT_NCHW, stage_w = w->info.format == CCV_TENSOR_FORMAT_NCHW;
	if (!stage_io && !stage_w) return conv_forw_nhwc(g, ai, w->data.f32, bias ? bias->data.f32 : 0, bi, cmd.algorithm, flags, stream_context);
	const size_t na = stage_io ? align256(sizeof(float) * tensor_count(a->info)) : 0, nb = stage_io ? align256(sizeof(float) * tensor_count(b->info)) : 0;
	const size_t nw = stage_w ? align256(sizeof(float) * tensor_count(w->info)) : 0;
	size_t inner = gemm_workspace_bound((long)g.N * g.OH * g.OW, g.Kg, (long)g.kh * g.kw * g.Cg);
	wino_plan_t wpl;
	if (cmd.algorithm != CONV_ALGO_IMPLICIT_GEMM && wino_plan(g, g.OH, g.OW, g.C, g.K, &wpl) && wpl.total() > inner) inner = wpl.total();
	if (cmd.algorithm != CONV_ALGO_IMPLICIT_GEMM && wino_fused_scratch_bound(g.K, g.C) > inner) inner = wino_fused_scratch_bound(g.K, g.C);
	WorkspaceScope ws(stream_context, na + nb + nw, inner);
	char* p = (char*)ws.prefix();
	if (!p) return CCV_NNC_EXEC_OOM;
	int ret;
	const float* wp = w->data.f32;
	if (stage_w) {
		if ((ret = weights_nchw_to_nhwc(w->data.f32, (float*)(p + na + nb), g.K, g.Cg, g.kh * g.kw, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
		wp = (const float*)(p + na + nb);
	}
	if (!stage_io) return conv_forw_nhwc(g, ai, wp, bias ? bias->data.f32 : 0, bi, cmd.algorithm, flags, stream_context);
	ccv_nnc_tensor_t at, bt;
	dense_nhwc_like(a, ai, (float*)p, &at);
	dense_nhwc_like(b, bi, (float*)(p + na), &bt);
	if ((ret = format_transform(a, &at, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
	Image4 as, bs;
	image4(&at, &as); image4(&bt, &bs);
	if ((ret = conv_forw_nhwc(g, as, wp, bias ? bias->data.f32 : 0, bs, cmd.algorithm, flags, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
	return format_transform(&bt, b, stream_context);
}

static int _conv_back(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	// inputs: gradient g, forward input a, [w]; outputs: [h], [dw], [dbias]   (ccv_nnc_convolution.c:16-37)
	if (input_size < 2 || output_size < 1 || !inputs[0]) return CCV_NNC_EXEC_INVALID;
	const ccv_nnc_tensor_t* gt = inputs[0];
	const ccv_nnc_tensor_t* a = inputs[1];
	const ccv_nnc_tensor_t* w = input_size > 2 ? inputs[2] : 0;
	ccv_nnc_tensor_t* h = outputs[0];
	ccv_nnc_tensor_t* dw = output_size > 1 ? outputs[1] : 0;
	ccv_nnc_tensor_t* dbias = output_size > 2 ? outputs[2] : 0;
	if (CCV_GET_DATA_TYPE(gt->info.datatype) != CCV_32F) return CCV_NNC_EXEC_INVALID;
	if (gt->info.format == CCV_TENSOR_FORMAT_NCHW && cmd.algorithm != CONV_ALGO_IMPLICIT_GEMM) {
		const int r = conv1x1_nchw_back<float>(cmd, hint, flags, gt, a, w, h, dw, dbias, stream_context);
		if (r != CCV_NNC_EXEC_NO_KERNEL) return r;
	}
	Image4 gi;
	if (!image4(gt, &gi)) return CCV_NNC_EXEC_INVALID;
	const ccv_nnc_tensor_t* shape_src = a ? a : h; // the forward input's shape
	if (!shape_src || shape_src->info.format != gt->info.format) return CCV_NNC_EXEC_INVALID;
	Image4 ai;
	if (!image4(shape_src, &ai)) return CCV_NNC_EXEC_INVALID;
	const ccv_nnc_tensor_t* wshape = dw ? dw : w;
	int K, kh, kw, Cg;
	if (!weights_shape(wshape, &K, &kh, &kw, &Cg)) return CCV_NNC_EXEC_INVALID;
	if (w && dw && w->info.format != dw->info.format) return CCV_NNC_EXEC_INVALID;
	conv_geom_t g;
	if (!conv_geometry(cmd, hint, ai, gi, 0, &g) || K != g.K || kh != g.kh || kw != g.kw || Cg != g.Cg) return CCV_NNC_EXEC_INVALID;
	if (dw && !a) return CCV_NNC_EXEC_INVALID;
	if (h && (!w || !tensor_contiguous(w))) return CCV_NNC_EXEC_INVALID;
	Image4 hi;
	if (h && (!image4(h, &hi) || hi.h != g.H || hi.w != g.W || hi.c != g.C || hi.n != g.N || h->info.format != gt->info.format)) return CCV_NNC_EXEC_INVALID;
	if (dbias && (!tensor_contiguous(dbias) || dbias->info.dim[0] != g.K)) return CCV_NNC_EXEC_INVALID;
	const int acc = (flags & CCV_NNC_ACCUMULATE_OUTPUT) ? 1 : 0;
	const bool stage_io = gt->info.format == CCV_TENSOR_FORMAT_NCHW, stage_w = wshape->info.format == CCV_TENSOR_FORMAT_NCHW;
	const size_t wbytes = align256(sizeof(float) * (size_t)g.K * g.kh * g.kw * g.Cg);
	const size_t ng = stage_io ? align256(sizeof(float) * tensor_count(gt->info)) : 0;
	const size_t na = stage_io && a && dw ? align256(sizeof(float) * tensor_count(a->info)) : 0;
	const size_t nh = stage_io && h ? align256(sizeof(float) * tensor_count(h->info)) : 0;
	const size_t nw = stage_w && h ? wbytes : 0, ndw = stage_w && dw ? wbytes : 0;
	const long P = (long)g.N * g.OH * g.OW;
	size_t inner = gemm_workspace_bound(g.Kg, (long)g.kh * g.kw * g.Cg, P);
	const size_t inner_d = gemm_workspace_bound((long)g.N * g.H * g.W, g.Cg, (long)g.kh * g.kw * g.Kg), inner_b = sizeof(float) * (size_t)g.K * 4096;
	if (inner_d > inner) inner = inner_d;
	if (inner_b > inner) inner = inner_b;
	wino_plan_t wpl;
	if (h && cmd.algorithm != CONV_ALGO_IMPLICIT_GEMM && wino_plan(g, g.H, g.W, g.K, g.C, &wpl) && wpl.total() > inner) inner = wpl.total();
	wino_wgrad_plan_t wgp;
	if (dw && cmd.algorithm != CONV_ALGO_IMPLICIT_GEMM && wino_wgrad_plan(g, &wgp) && wgp.total() > inner) inner = wgp.total();
	wino_wgrad_fused_plan_t wfgp;
	if (dw && cmd.algorithm != CONV_ALGO_IMPLICIT_GEMM && wino_wgrad_fused_plan(g, &wfgp) && wfgp.total() > inner) inner = wfgp.total();
	if (h && cmd.algorithm != CONV_ALGO_IMPLICIT_GEMM && wino_fused_scratch_bound(g.C, g.K) + wino_fused_mask_bound(g) > inner) inner = wino_fused_scratch_bound(g.C, g.K) + wino_fused_mask_bound(g);
	if (h && cmd.algorithm != CONV_ALGO_IMPLICIT_GEMM && conv_dgrad_parity_scratch(g) > inner) inner = conv_dgrad_parity_scratch(g);
	if (dw && cmd.algorithm != CONV_ALGO_IMPLICIT_GEMM && g.C == 3 && conv_c3_wgrad_scratch_bound(g.K) > inner) inner = conv_c3_wgrad_scratch_bound(g.K);
	WorkspaceScope ws(stream_context, ng + na + nh + nw + ndw, inner);
	char* p = (char*)ws.prefix();
	if ((ng + na + nh + nw + ndw) && !p) return CCV_NNC_EXEC_OOM;
	int ret;
	// stage the NCHW inputs
	ccv_nnc_tensor_t gs, as, hs;
	Image4 gim = gi, aim = ai, him = hi;
	if (stage_io) {
		dense_nhwc_like(gt, gi, (float*)p, &gs);
		if ((ret = format_transform(gt, &gs, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
		image4(&gs, &gim);
		if (na) {
			dense_nhwc_like(a, ai, (float*)(p + ng), &as);
			if ((ret = format_transform(a, &as, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
			image4(&as, &aim);
		}
		if (nh) { dense_nhwc_like(h, hi, (float*)(p + ng + na), &hs); image4(&hs, &him); }
	}
	bool bias_done = false;
	if (dw) {
		float* dwp = dw->data.f32;
		if (stage_w) {
			dwp = (float*)(p + ng + na + nh + nw);
			if (acc && (ret = weights_nchw_to_nhwc(dw->data.f32, dwp, g.K, g.Cg, g.kh * g.kw, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
		}
		if ((ret = conv_wgrad_nhwc(g, gim, aim, dwp, dbias ? dbias->data.f32 : 0, &bias_done, cmd.algorithm, flags, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
		if (stage_w && (ret = weights_nhwc_to_nchw(dwp, dw->data.f32, g.K, g.Cg, g.kh * g.kw, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
	}
	if (dbias && !bias_done) {
		if (!pixel_linear(gim)) return CCV_NNC_EXEC_INVALID;
		if ((ret = colsum_f32(gim.p, P, g.K, gim.sw, dbias->data.f32, acc, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
	}
	if (h) {
		const float* wp = w->data.f32;
		if (stage_w) {
			if ((ret = weights_nchw_to_nhwc(w->data.f32, (float*)(p + ng + na + nh), g.K, g.Cg, g.kh * g.kw, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
			wp = (const float*)(p + ng + na + nh);
		}
		if (tl_mask_want) tl_mask = aim;
		ret = conv_dgrad_nhwc(g, gim, wp, him, cmd.algorithm, flags, stream_context);
		tl_mask.p = 0;
		if (ret != CCV_NNC_EXEC_SUCCESS) return ret;
		if (stage_io && (ret = format_transform(&hs, h, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
	}
	return CCV_NNC_EXEC_SUCCESS;
}

// ---- CONVOLUTION_TRANSPOSE_FORWARD (lib/nnc/cmd/convolution/ccv_nnc_conv_transpose_cpu_ref.c:13-; replaces
// convolution/gpu/ccv_nnc_conv_transpose_gpu_cudnn.cu:24-172): b = bias + sum over the kernel taps of a scattered through w, with
// w [C_a][kh][kw][count / groups] -- exactly the DATA GRADIENT of the convolution whose output channels are a's channels and whose
// input is b (stride / border of the hint relate b to a as a convolution's input to its output).  So: the dgrad path above with
// (g := a, h := b), then the bias added per channel of b.
static __global__ void __launch_bounds__(256) chan_bias_add_kernel(float* b, const float* bias, const size_t n, const int C, const long inner)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) b[i] += bias[(int)((inner == 1 ? i : i / inner) % C)];
}
static int _conv_transpose_forw(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	if (input_size < 2 || output_size < 1 || !inputs[0] || !inputs[1] || !outputs[0]) return CCV_NNC_EXEC_INVALID;
	ccv_nnc_tensor_t* a = inputs[0];
	ccv_nnc_tensor_t* w = inputs[1];
	ccv_nnc_tensor_t* bias = input_size > 2 ? inputs[2] : 0;
	ccv_nnc_tensor_t* b = outputs[0];
	Image4 ai, bi;
	if (!image4(a, &ai) || !image4(b, &bi) || a->info.format != b->info.format || !tensor_contiguous(b)) return CCV_NNC_EXEC_INVALID;
	if (bi.c != cmd.info.convolution.count) return CCV_NNC_EXEC_INVALID; // (count sits at the same offset in both parameter structs)
	ccv_nnc_cmd_t conv = cmd;
	conv.cmd = CCV_NNC_CONVOLUTION_BACKWARD;
	conv.info.convolution.count = ai.c; // the convolution whose gradient this is has a's channels as its outputs
	ccv_nnc_tensor_t* ins[3] = { a, 0, w };
	ccv_nnc_tensor_t* outs[1] = { b };
	const int ret = _conv_back(conv, hint, flags & ~CCV_NNC_ACCUMULATE_OUTPUT, ins, 3, outs, 1, stream_context);
	if (ret != CCV_NNC_EXEC_SUCCESS) return ret;
	if (bias) {
		if (!tensor_contiguous(bias) || (int)tensor_count(bias->info) != bi.c) return CCV_NNC_EXEC_INVALID;
		const size_t n = tensor_count(b->info);
		const long inner = b->info.format == CCV_TENSOR_FORMAT_NCHW ? (long)bi.h * bi.w : 1;
		hipLaunchKernelGGL(chan_bias_add_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream_of(stream_context), b->data.f32, (const float*)bias->data.f32, n, bi.c, inner);
		HIP_ENFORCE(hipGetLastError());
	}
	return CCV_NNC_EXEC_SUCCESS;
}

// ---- half precision: the three contractions as implicit GEMMs on the half-precision core (mfma_gemm_f16.h) ---------------------
// CCV_16F activations (NHWC) and weights ([K][kh][kw][Cg]) with channel counts that are multiples of 4 (8-byte chunks); the same
// loaders as the fp32 path -- they compute element offsets -- over half pointers.  Anything else in half precision (NCHW tensors,
// NCHW-format weights, 3 input channels, odd strides) runs the fp32 kernels on fp32 images (half_stage.cpp).
static bool aligned8(const void* p) { return (((uintptr_t)p) & 7) == 0; }
static bool half_image_ok(const Image4& t) { return t.sc == 1 && aligned8(t.p) && t.sw % 4 == 0 && t.sh % 4 == 0 && (t.n == 1 || t.sn % 4 == 0) && image_fits_int(t); }

// The half-precision forward / data-gradient contraction with at most one 128 x 128 output tile per CU and a long reduction (the 4 x 4 maps of the CIFAR trainer:
// 256 tiles for 256 CUs, 144 K-steps each): 64 x 64 tiles put four workgroups on every CU.  Measured (tools/conv_half_bench.py, profiles/r04_v8_conv_half_bench_chunk8.txt)
// 512 -> 512 at 4 x 4, batch 512: forward 0.079 ms against 0.058 + 0.035 for eight K-slices of the big tile and the pass that folds their slabs (round 4's first
// answer, from before the 16-byte chunks: 0.127 unsplit, 0.072 + 0.035 in slices); at 7 x 7, batch 256 (392 tiles) the big tile unsplit stays best (0.108 vs 0.130).
static bool conv_h_small_tiles(const conv_geom_t& g, const long M, const int N, const int Kred)
{
	if (g.groups != 1 || g_force_tile || g_force_splits) return false;
	const long tiles = ((M + 127) / 128) * ((N + 127) / 128);
	return tiles <= (long)device_cu_count() && Kred >= 2304;
}
// K-slices of those contractions: 1, except where the launcher is told otherwise (nnc_mi355x_debug_force_splits: measurements)
static int conv_h_splits(const conv_geom_t& g, const long M, const int N, const int Kred)
{
	return g.groups == 1 && g_force_splits > 1 ? g_force_splits : 1;
}

// Can the forward / data-gradient contraction write an NCHW result itself (EpiStoreHT: groups of four pixels of a plane in one store, one K-slice)?
static bool conv_h_planar_ok(const conv_geom_t& g, const long M, const int N, const int Kred, const long P, const void* dst)
{
	return tune(TUNE_GEMM_VEC_EPILOGUE) == 1 && g.groups == 1 && P % 4 == 0 && M % 4 == 0 && aligned8(dst) && conv_h_splits(g, M, N, Kred) == 1 && M * (long)N < 0x7fffffff0L && M <= 0x7fffffffL;
}

// planar != 0: the result goes to that NCHW tensor [N][K][OH * OW] (b is not written)
// what a half-precision convolution moves once: input, filter, output (the launch record's algorithmic bytes: bench.py's per-shape bound)
static double conv_h_bytes(const conv_geom_t& g) { return 2.0 * ((double)g.N * g.H * g.W * g.C + (double)g.K * g.kh * g.kw * g.Cg + (double)g.N * g.OH * g.OW * g.K); }
static int conv_forw_h(const conv_geom_t& g, const Image4& a, const void* w, const void* bias, const Image4& b, const int flags, ccv_nnc_stream_context_t* const ctx, void* const planar = 0)
{
	prof_next_bytes(conv_h_bytes(g));
	const long M = (long)g.N * g.OH * g.OW;
	const int Kred = g.kh * g.kw * g.Cg;
	GemmOutH out = { (half_t*)b.p, b.sw, 1, (const half_t*)bias, 1.f, 0, 0 };
	KOrder ko;
	if (g.kh * g.kw > 1 && g.Cg % GEMM_BK == 0) ko.init(g.kh * g.kw, g.Cg);
	Im2colKC<true, false, false> la;
	la.p = a.p; la.s_n = a.sn; la.s_h = (int)a.sh; la.s_w = (int)a.sw; la.H = g.H; la.W = g.W;
	la.OW = g.OW; la.OHW = g.OH * g.OW; la.M = (int)M; la.C = g.Cg; la.KWC = g.kw * g.Cg; la.K = Kred;
	la.my = g.sy; la.mx = g.sx; la.oy_off = -g.pby; la.ox_off = -g.pbx; la.ty = g.dy; la.tx = g.dx; la.dv_y = 1; la.dv_x = 1;
	MatLoader<true, true> lb;
	lb.p = (const float*)w; lb.ldr = Kred; lb.ldk = 1; lb.R = g.Kg; lb.K = Kred;
	if (!planar && g.C % 8 == 0 && a.sw % 8 == 0 && conv_pointwise(g, a, b)) { // two plain matrices (conv_pointwise above): the buffer-load kernel
		MatLoader<true, true> pa;
		pa.p = a.p; pa.ldr = a.sw; pa.ldk = 1; pa.R = (int)M; pa.K = g.C;
		return gemm_run_h("conv_fwd_h_pointwise", pa, lb, out, (int)M, g.K, g.C, 1, 0L, 0L, 0L, 0L, 1, flags, ctx);
	}
	if (planar) {
		EpiStoreHT epi;
		epi.c = (half_t*)planar; epi.bias = (const half_t*)bias; epi.M = (int)M; epi.N = g.Kg; epi.P = g.OH * g.OW;
		return gemm_run_h_planar("conv_fwd_h", la, lb, epi, Kred, ctx, ko, conv_h_small_tiles(g, M, g.Kg, Kred));
	}
	return gemm_run_h("conv_fwd_h", la, lb, out, (int)M, g.Kg, Kred, g.groups, (long)g.Cg, (long)g.Kg * Kred, (long)g.Kg, (long)g.Kg, conv_h_splits(g, M, g.Kg, Kred), flags, ctx, ko, conv_h_small_tiles(g, M, g.Kg, Kred));
}

static int conv_dgrad_h(const conv_geom_t& g, const Image4& gr, const void* w, const Image4& h, const int flags, ccv_nnc_stream_context_t* const ctx, void* const planar = 0)
{
	prof_next_bytes(conv_h_bytes(g));
	const long M = (long)g.N * g.H * g.W;
	const int Kred = g.kh * g.kw * g.Kg;
	GemmOutH out = { (half_t*)h.p, h.sw, 1, 0, 1.f, 0, 0 };
	if (!planar && g.C % 8 == 0 && g.K % 8 == 0 && gr.sw % 8 == 0 && conv_pointwise(g, h, gr)) { // h [pixels][C] = g [pixels][K] . w [K][C]
		MatLoader<true, true> pa;
		pa.p = gr.p; pa.ldr = gr.sw; pa.ldk = 1; pa.R = (int)M; pa.K = g.K;
		MatLoader<false, true> pb;
		pb.p = (const float*)w; pb.ldr = 1; pb.ldk = g.C; pb.R = g.C; pb.K = g.K;
		return gemm_run_h("conv_dgrad_h_pointwise", pa, pb, out, (int)M, g.C, g.K, 1, 0L, 0L, 0L, 0L, 1, flags, ctx);
	}
	KOrder ko;
	if (g.kh * g.kw > 1 && g.Kg % GEMM_BK == 0) ko.init(g.kh * g.kw, g.Kg);
#define CONV_DGRAD_H(STRIDED) do { \
		Im2colKC<true, STRIDED, false> la; \
		la.p = gr.p; la.s_n = gr.sn; la.s_h = (int)gr.sh; la.s_w = (int)gr.sw; la.H = g.OH; la.W = g.OW; \
		la.OW = g.W; la.OHW = g.H * g.W; la.M = (int)M; la.C = g.Kg; la.KWC = g.kw * g.Kg; la.K = Kred; \
		la.my = 1; la.mx = 1; la.oy_off = g.pby; la.ox_off = g.pbx; la.ty = -g.dy; la.tx = -g.dx; la.dv_y = g.sy; la.dv_x = g.sx; \
		WgtDgradNC<true, false> lb; \
		lb.p = (const float*)w; lb.ko_stride = (long)g.kh * g.kw * g.Cg; lb.C = g.Cg; lb.Ko = g.Kg; lb.K = Kred; \
		if (planar) { \
			EpiStoreHT epi; \
			epi.c = (half_t*)planar; epi.bias = 0; epi.M = (int)M; epi.N = g.Cg; epi.P = g.H * g.W; \
			return gemm_run_h_planar("conv_dgrad_h", la, lb, epi, Kred, ctx, ko, conv_h_small_tiles(g, M, g.Cg, Kred)); \
		} \
		return gemm_run_h("conv_dgrad_h", la, lb, out, (int)M, g.Cg, Kred, g.groups, (long)g.Kg, (long)g.Kg * g.kh * g.kw * g.Cg, (long)g.Cg, 0L, conv_h_splits(g, M, g.Cg, Kred), flags, ctx, ko, conv_h_small_tiles(g, M, g.Cg, Kred)); \
	} while (0)
	if (g.sy != 1 || g.sx != 1) CONV_DGRAD_H(true);
	else CONV_DGRAD_H(false);
#undef CONV_DGRAD_H
}

static int conv_wgrad_h(const conv_geom_t& g, const Image4& gr, const Image4& a, void* dw, const int flags, ccv_nnc_stream_context_t* const ctx)
{
	prof_next_bytes(conv_h_bytes(g));
	const long P = (long)g.N * g.OH * g.OW;
	const int NN = g.kh * g.kw * g.Cg;
	GemmOutH out = { (half_t*)dw, (long)NN, 1, 0, 1.f, (flags & CCV_NNC_ACCUMULATE_OUTPUT) ? 1 : 0, 0 };
	MatLoader<false, true> la;
	la.p = gr.p; la.ldr = 1; la.ldk = gr.sw; la.R = g.Kg; la.K = (int)P;
	if (g.C % 8 == 0 && g.K % 8 == 0 && a.sw % 8 == 0 && gr.sw % 8 == 0 && conv_pointwise(g, a, gr)) { // dw [K][C] = g^T . a: both operands with their rows contiguous
		MatLoader<false, true> pb;
		pb.p = a.p; pb.ldr = 1; pb.ldk = a.sw; pb.R = g.C; pb.K = (int)P;
		return gemm_run_h("conv_wgrad_h_pointwise", la, pb, out, g.K, g.C, (int)P, 1, 0L, 0L, 0L, 0L, 0, flags, ctx);
	}
	Im2colNC<true, false> lb;
	lb.p = a.p; lb.s_n = a.sn; lb.s_h = (int)a.sh; lb.s_w = (int)a.sw; lb.H = g.H; lb.W = g.W; lb.OW = g.OW; lb.OHW = g.OH * g.OW;
	lb.C = g.Cg; lb.KWC = g.kw * g.Cg; lb.NN = NN; lb.K = (int)P; lb.sy = g.sy; lb.sx = g.sx; lb.py = g.pby; lb.px = g.pbx; lb.dy = g.dy; lb.dx = g.dx;
	return gemm_run_h("conv_wgrad_h", la, lb, out, g.Kg, NN, (int)P, g.groups, (long)g.Kg, (long)g.Cg, (long)g.Kg * NN, 0L, 0, flags, ctx);
}

// CCV_NNC_EXEC_NO_KERNEL = "not this path": the caller runs the command on fp32 images instead.
// ---- half-precision tensors in NCHW, kernels larger than 1 x 1 (the CIFAR-10 / ResNet trainers in their fp16 mode) ----------------
// The fp32 Winograd kernels are the fastest 3 x 3 path this backend has (the half-precision implicit-GEMM filter gradient is slower
// than fp32 Winograd's), and NCHW tensors are re-laid-out to NHWC for them anyway: here that one pass per tensor also carries the
// half <-> float conversion (transpose_half_to_float / transpose_float_to_half), so an fp16 NCHW convolution moves
// 6 bytes per activation element around the kernel where "convert, transpose, run, transpose, convert" moved 22.
static void dense_nhwc_f32(const Image4& li, const bool batched, float* data, ccv_nnc_tensor_t* out, Image4* oi)
{
	memset(out, 0, sizeof(*out));
	out->type = CCV_TENSOR_GPU_MEMORY;
	out->info.type = CCV_TENSOR_GPU_MEMORY;
	out->info.format = CCV_TENSOR_FORMAT_NHWC;
	out->info.datatype = CCV_32F;
	const int b = batched ? 1 : 0;
	if (b) out->info.dim[0] = li.n;
	out->info.dim[b] = li.h; out->info.dim[b + 1] = li.w; out->info.dim[b + 2] = li.c;
	out->data.f32 = data;
	image4(out, oi);
}
// Which half NCHW convolutions take the f16 implicit-GEMM core (TUNE_CONV_NCHW_HALF_F16 = the least reduction channels; the loaders' chunk conditions)
static bool conv_nchw_half_f16_ok(const conv_geom_t& g)
{
	const long least = tune(TUNE_CONV_NCHW_HALF_F16);
	return least > 0 && g.groups == 1 && g.Cg >= least && g.Cg % 8 == 0 && g.K % 8 == 0 && (long)g.N * g.OH * g.OW <= 0x7fffffffL && (long)g.N * g.H * g.W <= 0x7fffffffL;
}
static int conv_nchw_half_forw(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, const ccv_nnc_tensor_t* a, const ccv_nnc_tensor_t* w, const ccv_nnc_tensor_t* bias, ccv_nnc_tensor_t* b, ccv_nnc_stream_context_t* const ctx)
{
	if ((flags & CCV_NNC_ACCUMULATE_OUTPUT) || w->info.format != CCV_TENSOR_FORMAT_NCHW || !tensor_contiguous(w)) return CCV_NNC_EXEC_NO_KERNEL;
	int Na, Ca, Pa, Nb, Cb, Pb;
	if (!nchw_dense(a, &Na, &Ca, &Pa) || !nchw_dense(b, &Nb, &Cb, &Pb)) return CCV_NNC_EXEC_NO_KERNEL;
	Image4 ai, bi;
	if (!image4(a, &ai) || !image4(b, &bi)) return CCV_NNC_EXEC_NO_KERNEL;
	int K, kh, kw, Cg;
	if (!weights_shape(w, &K, &kh, &kw, &Cg)) return CCV_NNC_EXEC_NO_KERNEL;
	conv_geom_t g;
	if (!conv_geometry(cmd, hint, ai, bi, 0, &g) || K != g.K || kh != g.kh || kw != g.kw || Cg != g.Cg) return CCV_NNC_EXEC_INVALID;
	if (bias && (bias->info.dim[0] != g.K || !tensor_contiguous(bias))) return CCV_NNC_EXEC_INVALID;
	int ret;
	// Enough reduction channels: the f16 implicit GEMM on the matrix cores between HALF transposes (2 + 2 bytes per activation element around the kernel instead
	// of 6 + 6, and 456-539 TFLOP/s of direct arithmetic against the fp32 Winograd kernels' ~190 direct-equivalent; tools/half_bench.py).  The ReLU a look-ahead
	// may have asked for is left to its own pass there (tl_relu_done stays clear).
	if (conv_nchw_half_f16_ok(g)) {
		const size_t ha = align256(sizeof(half_t) * tensor_count(a->info)), hb = align256(sizeof(half_t) * tensor_count(b->info)), hw = align256(sizeof(half_t) * tensor_count(w->info));
		char* const q = (char*)nnc_staging_of(ctx, ha + hb + hw);
		if (!q) return CCV_NNC_EXEC_OOM;
		if ((ret = transpose_half(a->data.u8, q, Na, Ca, Pa, ctx)) != CCV_NNC_EXEC_SUCCESS) return ret;
		if ((ret = transpose_half(w->data.u8, q + ha + hb, g.K, g.Cg, g.kh * g.kw, ctx)) != CCV_NNC_EXEC_SUCCESS) return ret;
		ccv_nnc_tensor_t at, bt;
		Image4 as, bs;
		dense_nhwc_f32(ai, tensor_nd(a->info.dim) == 4, (float*)q, &at, &as);            // (geometry only: element strides are the same for halves)
		dense_nhwc_f32(bi, tensor_nd(b->info.dim) == 4, (float*)(q + ha), &bt, &bs);
		// the contraction writes the NCHW result itself where it can (groups of four pixels of a plane per store: no NHWC image of b, no pass to re-lay it)
		if (conv_h_planar_ok(g, (long)g.N * g.OH * g.OW, g.Kg, g.kh * g.kw * g.Cg, Pb, b->data.u8)) return conv_forw_h(g, as, q + ha + hb, bias ? bias->data.u8 : 0, bs, flags, ctx, b->data.u8);
		if ((ret = conv_forw_h(g, as, q + ha + hb, bias ? bias->data.u8 : 0, bs, flags, ctx)) != CCV_NNC_EXEC_SUCCESS) return ret;
		return transpose_half(q + ha, b->data.u8, Nb, Pb, Cb, ctx);
	}
	const size_t na = align256(sizeof(float) * tensor_count(a->info)), nb = align256(sizeof(float) * tensor_count(b->info));
	const size_t nw = align256(sizeof(float) * tensor_count(w->info)), nbias = bias ? align256(sizeof(float) * (size_t)g.K) : 0;
	char* const p = (char*)nnc_staging_of(ctx, na + nb + nw + nbias);
	if (!p) return CCV_NNC_EXEC_OOM;
	float* const A = (float*)p; float* const B = (float*)(p + na); float* const W = (float*)(p + na + nb); float* const BI = bias ? (float*)(p + na + nb + nw) : 0;
	if ((ret = transpose_half_to_float(a->data.u8, A, Na, Ca, Pa, ctx)) != CCV_NNC_EXEC_SUCCESS) return ret;
	if ((ret = transpose_half_to_float(w->data.u8, W, g.K, g.Cg, g.kh * g.kw, ctx)) != CCV_NNC_EXEC_SUCCESS) return ret;
	if (bias && (ret = half_to_float(bias->data.u8, BI, (size_t)g.K, ctx)) != CCV_NNC_EXEC_SUCCESS) return ret;
	ccv_nnc_tensor_t at, bt;
	Image4 as, bs;
	dense_nhwc_f32(ai, tensor_nd(a->info.dim) == 4, A, &at, &as);
	dense_nhwc_f32(bi, tensor_nd(b->info.dim) == 4, B, &bt, &bs);
	if ((ret = conv_forw_nhwc(g, as, W, BI, bs, cmd.algorithm, flags, ctx)) != CCV_NNC_EXEC_SUCCESS) return ret;
	return transpose_float_to_half(B, b->data.u8, Nb, Pb, Cb, ctx);
}
static int conv_nchw_half_back(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, const ccv_nnc_tensor_t* gt, const ccv_nnc_tensor_t* a, const ccv_nnc_tensor_t* w, ccv_nnc_tensor_t* h, ccv_nnc_tensor_t* dw, ccv_nnc_tensor_t* dbias, ccv_nnc_stream_context_t* const ctx)
{
	if (flags & CCV_NNC_ACCUMULATE_OUTPUT) return CCV_NNC_EXEC_NO_KERNEL;
	const ccv_nnc_tensor_t* shape_src = a ? a : h;
	const ccv_nnc_tensor_t* wshape = dw ? dw : w;
	if (wshape->info.format != CCV_TENSOR_FORMAT_NCHW || (w && (w->info.format != CCV_TENSOR_FORMAT_NCHW || !tensor_contiguous(w))) || (dw && !tensor_contiguous(dw))) return CCV_NNC_EXEC_NO_KERNEL;
	int Ng, Cgr, Pg, Na, Ca, Pa;
	if (!nchw_dense(gt, &Ng, &Cgr, &Pg) || !nchw_dense(shape_src, &Na, &Ca, &Pa) || (h && !nchw_dense(h, &Na, &Ca, &Pa))) return CCV_NNC_EXEC_NO_KERNEL;
	Image4 gi, ai;
	if (!image4(gt, &gi) || !image4(shape_src, &ai)) return CCV_NNC_EXEC_NO_KERNEL;
	int K, kh, kw, Cg;
	if (!weights_shape(wshape, &K, &kh, &kw, &Cg)) return CCV_NNC_EXEC_NO_KERNEL;
	conv_geom_t g;
	if (!conv_geometry(cmd, hint, ai, gi, 0, &g) || K != g.K || kh != g.kh || kw != g.kw || Cg != g.Cg) return CCV_NNC_EXEC_INVALID;
	if ((dw && !a) || (h && !w)) return CCV_NNC_EXEC_INVALID;
	Image4 hi;
	if (h && (!image4(h, &hi) || hi.h != g.H || hi.w != g.W || hi.c != g.C || hi.n != g.N)) return CCV_NNC_EXEC_INVALID;
	if (dbias && (!tensor_contiguous(dbias) || dbias->info.dim[0] != g.K)) return CCV_NNC_EXEC_INVALID;
	int ret;
	// Enough channels on both sides: the whole backward pass on the f16 implicit-GEMM core between HALF transposes (see conv_nchw_half_forw): since the row-contiguous
	// operands of mfma_gemm_f16.h go through the LDS transpose read the filter gradient runs 430 and the data gradient 550 TFLOP/s of direct arithmetic there
	// (round 2: 118 / 250, which lost to the fp32 Winograd kernels' ~150 - 250 direct-equivalent).  A ReLU-backward mask the look-ahead offers is left to its own pass.
	if (conv_nchw_half_f16_ok(g) && g.Kg >= tune(TUNE_CONV_NCHW_HALF_F16) && tune(TUNE_CONV_NCHW_HALF_F16) > 0) {
		const size_t hg = align256(sizeof(half_t) * tensor_count(gt->info)), ha = dw ? align256(sizeof(half_t) * tensor_count(a->info)) : 0, hh = h ? align256(sizeof(half_t) * tensor_count(h->info)) : 0;
		const size_t hw = align256(sizeof(half_t) * (size_t)g.K * g.kh * g.kw * g.Cg);
		char* const q = (char*)nnc_staging_of(ctx, hg + ha + hh + (h ? hw : 0) + (dw ? hw : 0));
		if (!q) return CCV_NNC_EXEC_OOM;
		char* const G16 = q; char* const A16 = q + hg; char* const H16 = q + hg + ha; char* const W16 = q + hg + ha + hh; char* const DW16 = W16 + (h ? hw : 0);
		// the bias gradient -- per-channel sums of the output gradient -- is taken while the gradient is re-laid: the pass sums every plane row over its 64-pixel tiles,
		// a fold of those partials finishes it (round 6: the separate column-sum pass over the re-laid gradient was 4 - 5 % of the CIFAR trainer's half-precision step)
		bool bias_done = false;
		if (dbias) {
			const long slices = transpose_half_rowsum_slices(Ng, Pg);
			float* const part = (float*)workspace_of(ctx, sizeof(float) * (size_t)(slices + 256) * (size_t)Cgr); // (+ 256 rows: colsum_partials_f16's grouped level)
			if (part && transpose_half_rowsum(gt->data.u8, G16, Ng, Cgr, Pg, part, ctx) == CCV_NNC_EXEC_SUCCESS) {
				if ((ret = colsum_partials_f16(part, slices, g.K, dbias->data.u8, 0, ctx)) != CCV_NNC_EXEC_SUCCESS) return ret;
				bias_done = true;
			}
		}
		if (!bias_done && (ret = transpose_half(gt->data.u8, G16, Ng, Cgr, Pg, ctx)) != CCV_NNC_EXEC_SUCCESS) return ret;
		ccv_nnc_tensor_t g16, a16, h16;
		Image4 g16i, a16i, h16i;
		dense_nhwc_f32(gi, tensor_nd(gt->info.dim) == 4, (float*)G16, &g16, &g16i); // (geometry only: element strides are the same for halves)
		if (dw) {
			if ((ret = transpose_half(a->data.u8, A16, Na, Ca, Pa, ctx)) != CCV_NNC_EXEC_SUCCESS) return ret;
			dense_nhwc_f32(ai, tensor_nd(a->info.dim) == 4, (float*)A16, &a16, &a16i);
			if ((ret = conv_wgrad_h(g, g16i, a16i, DW16, 0, ctx)) != CCV_NNC_EXEC_SUCCESS) return ret;
			if ((ret = transpose_half(DW16, dw->data.u8, g.K, g.kh * g.kw, g.Cg, ctx)) != CCV_NNC_EXEC_SUCCESS) return ret; // [K][khkw][C] -> [K][C][khkw]
		}
		if (dbias && !bias_done && (ret = colsum_f16(g16i.p, (long)g.N * g.OH * g.OW, g.K, g16i.sw, dbias->data.u8, 0, ctx)) != CCV_NNC_EXEC_SUCCESS) return ret;
		if (h) {
			if ((ret = transpose_half(w->data.u8, W16, g.K, g.Cg, g.kh * g.kw, ctx)) != CCV_NNC_EXEC_SUCCESS) return ret;
			dense_nhwc_f32(hi, tensor_nd(h->info.dim) == 4, (float*)H16, &h16, &h16i);
			if (conv_h_planar_ok(g, (long)g.N * g.H * g.W, g.Cg, g.kh * g.kw * g.Kg, Pa, h->data.u8)) {
				if ((ret = conv_dgrad_h(g, g16i, W16, h16i, 0, ctx, h->data.u8)) != CCV_NNC_EXEC_SUCCESS) return ret;
			} else {
				if ((ret = conv_dgrad_h(g, g16i, W16, h16i, 0, ctx)) != CCV_NNC_EXEC_SUCCESS) return ret;
				if ((ret = transpose_half(H16, h->data.u8, Na, Pa, Ca, ctx)) != CCV_NNC_EXEC_SUCCESS) return ret;
			}
		}
		return CCV_NNC_EXEC_SUCCESS;
	}
	const size_t wbytes = align256(sizeof(float) * (size_t)g.K * g.kh * g.kw * g.Cg);
	const size_t ng = align256(sizeof(float) * tensor_count(gt->info));
	const size_t na = dw ? align256(sizeof(float) * tensor_count(a->info)) : 0, nh = h ? align256(sizeof(float) * tensor_count(h->info)) : 0;
	const size_t nw = h ? wbytes : 0, ndw = dw ? wbytes : 0, ndb = dbias ? align256(sizeof(float) * (size_t)g.K) : 0;
	char* const p = (char*)nnc_staging_of(ctx, ng + na + nh + nw + ndw + ndb);
	if (!p) return CCV_NNC_EXEC_OOM;
	float* const G = (float*)p; float* const A = (float*)(p + ng); float* const Hh = (float*)(p + ng + na);
	float* const W = (float*)(p + ng + na + nh); float* const DW = (float*)(p + ng + na + nh + nw); float* const DB = dbias ? (float*)(p + ng + na + nh + nw + ndw) : 0;
	if ((ret = transpose_half_to_float(gt->data.u8, G, Ng, Cgr, Pg, ctx)) != CCV_NNC_EXEC_SUCCESS) return ret;
	ccv_nnc_tensor_t gs, as, hs;
	Image4 gim, aim, him;
	dense_nhwc_f32(gi, tensor_nd(gt->info.dim) == 4, G, &gs, &gim);
	bool bias_done = false;
	if (dw) {
		if ((ret = transpose_half_to_float(a->data.u8, A, Na, Ca, Pa, ctx)) != CCV_NNC_EXEC_SUCCESS) return ret;
		dense_nhwc_f32(ai, tensor_nd(a->info.dim) == 4, A, &as, &aim);
		if ((ret = conv_wgrad_nhwc(g, gim, aim, DW, DB, &bias_done, cmd.algorithm, flags, ctx)) != CCV_NNC_EXEC_SUCCESS) return ret;
		if ((ret = transpose_float_to_half(DW, dw->data.u8, g.K, g.kh * g.kw, g.Cg, ctx)) != CCV_NNC_EXEC_SUCCESS) return ret; // [K][khkw][C] -> [K][C][khkw]
	}
	if (dbias) {
		if (!bias_done && (ret = colsum_f32(gim.p, (long)g.N * g.OH * g.OW, g.K, gim.sw, DB, 0, ctx)) != CCV_NNC_EXEC_SUCCESS) return ret;
		if ((ret = float_to_half(DB, dbias->data.u8, (size_t)g.K, ctx)) != CCV_NNC_EXEC_SUCCESS) return ret;
	}
	if (h) {
		if ((ret = transpose_half_to_float(w->data.u8, W, g.K, g.Cg, g.kh * g.kw, ctx)) != CCV_NNC_EXEC_SUCCESS) return ret;
		dense_nhwc_f32(hi, tensor_nd(h->info.dim) == 4, Hh, &hs, &him);
		if ((ret = conv_dgrad_nhwc(g, gim, W, him, cmd.algorithm, flags, ctx)) != CCV_NNC_EXEC_SUCCESS) return ret;
		if ((ret = transpose_float_to_half(Hh, h->data.u8, Na, Pa, Ca, ctx)) != CCV_NNC_EXEC_SUCCESS) return ret;
	}
	return CCV_NNC_EXEC_SUCCESS;
}

static int _conv_forw_half(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	if (input_size < 2 || output_size < 1 || !inputs[0] || !inputs[1] || !outputs[0]) return CCV_NNC_EXEC_INVALID;
	const ccv_nnc_tensor_t* a = inputs[0];
	const ccv_nnc_tensor_t* w = inputs[1];
	const ccv_nnc_tensor_t* bias = input_size > 2 ? inputs[2] : 0;
	ccv_nnc_tensor_t* b = outputs[0];
	if (a->info.format == CCV_TENSOR_FORMAT_NCHW && b->info.format == CCV_TENSOR_FORMAT_NCHW) {
		const int r = conv1x1_nchw_forw<half_t>(cmd, hint, flags, a, w, bias, b, stream_context);
		return r != CCV_NNC_EXEC_NO_KERNEL ? r : conv_nchw_half_forw(cmd, hint, flags, a, w, bias, b, stream_context);
	}
	if (a->info.format != CCV_TENSOR_FORMAT_NHWC || b->info.format != CCV_TENSOR_FORMAT_NHWC || w->info.format != CCV_TENSOR_FORMAT_NHWC) return CCV_NNC_EXEC_NO_KERNEL;
	Image4 ai, bi;
	if (!image4(a, &ai) || !image4(b, &bi)) return CCV_NNC_EXEC_NO_KERNEL;
	int K, kh, kw, Cg;
	if (!weights_shape(w, &K, &kh, &kw, &Cg)) return CCV_NNC_EXEC_NO_KERNEL;
	conv_geom_t g;
	if (!conv_geometry(cmd, hint, ai, bi, 0, &g) || K != g.K || kh != g.kh || kw != g.kw || Cg != g.Cg) return CCV_NNC_EXEC_INVALID;
	if (bias && (bias->info.dim[0] != g.K || !tensor_contiguous(bias))) return CCV_NNC_EXEC_INVALID;
	if (g.Cg % 4 || !half_image_ok(ai) || !pixel_linear(bi) || !aligned8(w->data.u8) || (long)g.N * g.OH * g.OW > 0x7fffffffL) return CCV_NNC_EXEC_NO_KERNEL;
	return conv_forw_h(g, ai, w->data.u8, bias ? bias->data.u8 : 0, bi, flags, stream_context);
}

static int _conv_back_half(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	if (input_size < 2 || output_size < 1 || !inputs[0]) return CCV_NNC_EXEC_INVALID;
	const ccv_nnc_tensor_t* gt = inputs[0];
	const ccv_nnc_tensor_t* a = inputs[1];
	const ccv_nnc_tensor_t* w = input_size > 2 ? inputs[2] : 0;
	ccv_nnc_tensor_t* h = outputs[0];
	ccv_nnc_tensor_t* dw = output_size > 1 ? outputs[1] : 0;
	ccv_nnc_tensor_t* dbias = output_size > 2 ? outputs[2] : 0;
	const ccv_nnc_tensor_t* shape_src = a ? a : h;
	const ccv_nnc_tensor_t* wshape = dw ? dw : w;
	if (!shape_src || !wshape) return CCV_NNC_EXEC_INVALID;
	if (gt->info.format == CCV_TENSOR_FORMAT_NCHW) {
		const int r = conv1x1_nchw_back<half_t>(cmd, hint, flags, gt, a, w, h, dw, dbias, stream_context);
		return r != CCV_NNC_EXEC_NO_KERNEL ? r : conv_nchw_half_back(cmd, hint, flags, gt, a, w, h, dw, dbias, stream_context);
	}
	if (gt->info.format != CCV_TENSOR_FORMAT_NHWC || shape_src->info.format != CCV_TENSOR_FORMAT_NHWC || wshape->info.format != CCV_TENSOR_FORMAT_NHWC || (w && w->info.format != CCV_TENSOR_FORMAT_NHWC) || (h && h->info.format != CCV_TENSOR_FORMAT_NHWC)) return CCV_NNC_EXEC_NO_KERNEL;
	Image4 gi, ai, hi;
	if (!image4(gt, &gi) || !image4(shape_src, &ai)) return CCV_NNC_EXEC_NO_KERNEL;
	int K, kh, kw, Cg;
	if (!weights_shape(wshape, &K, &kh, &kw, &Cg)) return CCV_NNC_EXEC_NO_KERNEL;
	conv_geom_t g;
	if (!conv_geometry(cmd, hint, ai, gi, 0, &g) || K != g.K || kh != g.kh || kw != g.kw || Cg != g.Cg) return CCV_NNC_EXEC_INVALID;
	if (dw && !a) return CCV_NNC_EXEC_INVALID;
	if (h && (!w || !tensor_contiguous(w))) return CCV_NNC_EXEC_INVALID;
	if (h && (!image4(h, &hi) || hi.h != g.H || hi.w != g.W || hi.c != g.C || hi.n != g.N)) return CCV_NNC_EXEC_INVALID;
	if (dbias && (!tensor_contiguous(dbias) || dbias->info.dim[0] != g.K)) return CCV_NNC_EXEC_INVALID;
	if (g.Cg % 4 || g.Kg % 4 || !half_image_ok(gi) || !pixel_linear(gi) || (long)g.N * g.OH * g.OW > 0x7fffffffL || (long)g.N * g.H * g.W > 0x7fffffffL) return CCV_NNC_EXEC_NO_KERNEL;
	if (dw && (!half_image_ok(ai) || !aligned8(dw->data.u8))) return CCV_NNC_EXEC_NO_KERNEL;
	if (h && (!pixel_linear(hi) || !aligned8(w->data.u8) || !aligned8(hi.p))) return CCV_NNC_EXEC_NO_KERNEL;
	int ret;
	if (dw && (ret = conv_wgrad_h(g, gi, ai, dw->data.u8, flags, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
	if (dbias && (ret = colsum_f16(gi.p, (long)g.N * g.OH * g.OW, g.K, gi.sw, dbias->data.u8, (flags & CCV_NNC_ACCUMULATE_OUTPUT) ? 1 : 0, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
	if (h && (ret = conv_dgrad_h(g, gi, w->data.u8, hi, flags, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
	return CCV_NNC_EXEC_SUCCESS;
}

static bool all_half(ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size)
{
	for (int i = 0; i < input_size + output_size; i++) {
		const ccv_nnc_tensor_t* t = i < input_size ? inputs[i] : outputs[i - input_size];
		if (t && CCV_GET_DATA_TYPE(t->info.datatype) != CCV_16F) return false;
	}
	return true;
}

// The registered exec functions: fp32 tensors -> the fp32 paths above; half precision throughout and chunk-readable -> the
// half-precision core; any other command with a half tensor -> the fp32 paths on fp32 images.
static int conv_forw_dispatch(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	if (!any_half_tensor(inputs, input_size, outputs, output_size)) return _conv_forw(cmd, hint, flags, inputs, input_size, outputs, output_size, stream_context);
	if (all_half(inputs, input_size, outputs, output_size)) {
		const int r = _conv_forw_half(cmd, hint, flags, inputs, input_size, outputs, output_size, stream_context);
		if (r != CCV_NNC_EXEC_NO_KERNEL) return r;
	}
	return half_staged_exec(_conv_forw, cmd, hint, flags, inputs, input_size, outputs, output_size, stream_context);
}
static int conv_forw_entry(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context);
// The registered entry: a convolution whose like has run before is recorded, not launched -- the in-place RELU_FORWARD the reference's
// graphs issue next folds into it (peephole.cpp); anything else on the stream launches it as it is.
static int _conv_forw_any(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	uint64_t sig;
	if (const int e = deferred_take_error(stream_context)) return e; // a recorded command failed when a flush launched it (peephole.cpp)
	if (deferred_try(_conv_forw_any, DEFER_CONV_FORWARD, cmd, hint, flags, inputs, input_size, outputs, output_size, stream_context, &sig)) return CCV_NNC_EXEC_SUCCESS;
	const int r = conv_forw_entry(cmd, hint, flags, inputs, input_size, outputs, output_size, stream_context);
	if (r == CCV_NNC_EXEC_SUCCESS) deferred_mark_good(sig);
	return r;
}
static int conv_forw_entry(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	MarkerScope marker(cmd.cmd);
	// Opt-in fusion (a caller that knows the convolution's only consumer is a RELU_FORWARD): algorithm = FUSE_RELU | (0..2, or 0xff
	// for the backend's choice).  The host's autotuner never produces such a value (it walks 0 .. algorithms - 1).
	if (cmd.algorithm < 0 || !(cmd.algorithm & NNC_MI355X_CONV_ALGO_FUSE_RELU)) return conv_forw_dispatch(cmd, hint, flags, inputs, input_size, outputs, output_size, stream_context);
	if (flags & CCV_NNC_ACCUMULATE_OUTPUT) return CCV_NNC_EXEC_INVALID;
	ccv_nnc_cmd_t plain = cmd;
	plain.algorithm = (cmd.algorithm & 0xff) == 0xff ? -1 : (cmd.algorithm & 0xff);
	tl_relu_want = 1; tl_relu_done = 0;
	const int r = conv_forw_dispatch(plain, hint, flags, inputs, input_size, outputs, output_size, stream_context);
	const int done = tl_relu_done;
	tl_relu_want = 0; tl_relu_done = 0;
	if (r != CCV_NNC_EXEC_SUCCESS || done) return r;
	return relu_inplace(outputs[0], stream_context); // the path taken has no fused epilogue (implicit GEMM, half core, ...): one more pass
}
static int conv_back_dispatch(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context);
// NNC_MI355X_CONV_ALGO_FUSE_RELU on the backward command: h = a > 0 ? (data gradient) : 0 -- the RELU_BACKWARD of the map a that would
// run on h next.  Masked where the data gradient is written by the Winograd kernels; one in-place pass behind the others.
static int conv_back_entry(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context);
static int _conv_back_any(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	uint64_t sig; // (recorded like the forward: the RELU_BACKWARD of the map this command read may follow on the gradient it writes)
	if (const int e = deferred_take_error(stream_context)) return e; // a recorded command failed when a flush launched it (peephole.cpp)
	if (deferred_try(_conv_back_any, DEFER_CONV_BACKWARD, cmd, hint, flags, inputs, input_size, outputs, output_size, stream_context, &sig)) return CCV_NNC_EXEC_SUCCESS;
	const int r = conv_back_entry(cmd, hint, flags, inputs, input_size, outputs, output_size, stream_context);
	if (r == CCV_NNC_EXEC_SUCCESS) deferred_mark_good(sig);
	return r;
}
// the weight / bias gradients of a backward command that has just been enqueued: the overlapped all-reduce of deployment (b) starts behind THEM, not behind
// the rest of the backward pass (cmd_comm.cpp "Overlap"); an accumulating command is one of several writers: stream order for that gradient
static int conv_back_report(const int r, const int flags, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	if (r == CCV_NNC_EXEC_SUCCESS && g_comm_overlap_on.load(std::memory_order_relaxed))
	{
		if (flags & CCV_NNC_ACCUMULATE_OUTPUT) { for (int i = 1; i < output_size && i < 3; i++) comm_gradient_touched(outputs[i]); }
		else if (output_size > 1) comm_gradients_written(outputs + 1, output_size > 3 ? 2 : output_size - 1, stream_context);
	}
	return r;
}
static int conv_back_entry(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	MarkerScope marker(cmd.cmd);
	if (cmd.algorithm < 0 || !(cmd.algorithm & NNC_MI355X_CONV_ALGO_FUSE_RELU)) return conv_back_report(conv_back_dispatch(cmd, hint, flags, inputs, input_size, outputs, output_size, stream_context), flags, outputs, output_size, stream_context);
	ccv_nnc_cmd_t plain = cmd;
	plain.algorithm = (cmd.algorithm & 0xff) == 0xff ? -1 : (cmd.algorithm & 0xff);
	ccv_nnc_tensor_t* const h = output_size > 0 ? outputs[0] : 0;
	const ccv_nnc_tensor_t* const a = input_size > 1 ? inputs[1] : 0;
	if (!h) return conv_back_report(conv_back_dispatch(plain, hint, flags, inputs, input_size, outputs, output_size, stream_context), flags, outputs, output_size, stream_context); // no data gradient asked for: nothing to mask
	if ((flags & CCV_NNC_ACCUMULATE_OUTPUT) || !a || !tensor_contiguous(h) || !tensor_contiguous(a) || h->info.datatype != a->info.datatype || h->info.format != a->info.format || tensor_count(h->info) != tensor_count(a->info)) return CCV_NNC_EXEC_INVALID;
	tl_mask_want = 1; tl_mask_done = 0;
	const int r = conv_back_report(conv_back_dispatch(plain, hint, flags, inputs, input_size, outputs, output_size, stream_context), flags, outputs, output_size, stream_context);
	const int done = tl_mask_done;
	tl_mask_want = 0; tl_mask_done = 0; tl_mask.p = 0;
	if (r != CCV_NNC_EXEC_SUCCESS || done) return r;
	return relu_back_inplace(h, a, stream_context);
}
static int conv_back_dispatch(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	if (!any_half_tensor(inputs, input_size, outputs, output_size)) return _conv_back(cmd, hint, flags, inputs, input_size, outputs, output_size, stream_context);
	if (all_half(inputs, input_size, outputs, output_size)) {
		const int r = _conv_back_half(cmd, hint, flags, inputs, input_size, outputs, output_size, stream_context);
		if (r != CCV_NNC_EXEC_NO_KERNEL) return r;
	}
	return half_staged_exec(_conv_back, cmd, hint, flags, inputs, input_size, outputs, output_size, stream_context);
}

// autotune (ccv_nnc.h:323; what lib/nnc/cmd/convolution/gpu/ccv_nnc_conv_gpu_cudnn.cu:116-202 does with cudnnFind*): run the
// command under every algorithm on the caller's tensors, HIP-event timed on its stream, and return the fastest.  Outputs
// are overwritten with the same values each time (the host autotunes before the first real execution, ccv_nnc_cmd.c:344-578).
static int _conv_autotune(const ccv_nnc_cmd_t cmd, const size_t max_workspace_size, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	(void)max_workspace_size;
	if (any_half_tensor(inputs, input_size, outputs, output_size)) return -1; // half precision: the backend's own choice (the trials below time the fp32 kernels)
	if (any_palettized(inputs, input_size)) return -1; // palettized filters: the trials below would read the byte stream as floats; the backend's own choice
	{ // NNC_MI355X_CONV_AUTOTUNE=0: no timed trials, the backend's own choice -- runs that must be reproducible bit for bit from process to process (the timing of
	  // near-equal algorithms flips with the machine's load: two emulator processes side by side picked differently, tests/test_via_host.py's captured 2-device step)
		static int trials = -1;
		if (trials < 0) { const char* e = getenv("NNC_MI355X_CONV_AUTOTUNE"); trials = (e && *e == '0') ? 0 : 1; }
		if (!trials) return -1;
	}
	const bool fwd = cmd.cmd == CCV_NNC_CONVOLUTION_FORWARD;
	hipStream_t stream = stream_of(stream_context);
	hipEvent_t e0, e1;
	HIP_ENFORCE(hipEventCreate(&e0));
	HIP_ENFORCE(hipEventCreate(&e1));
	int best = 0, have_best = 0;
	float best_ms = 0;
	// trials must not leave their mark: CCV_NNC_ACCUMULATE_OUTPUT would add dw / dbias once per trial, so the trials overwrite
	// (the host autotunes before the first real execution and the real execution follows with the caller's flags)
	const int trial_flags = flags & ~CCV_NNC_ACCUMULATE_OUTPUT;
	for (int algo = 0; algo < CONV_ALGO_COUNT; algo++) {
		ccv_nnc_cmd_t c = cmd;
		c.algorithm = algo;
		float ms = 0;
		int ok = 1;
		for (int trial = 0; trial < 2 && ok; trial++) { // first trial warms the workspace up
			HIP_ENFORCE(hipEventRecord(e0, stream));
			const int ret = fwd ? _conv_forw(c, hint, trial_flags, inputs, input_size, outputs, output_size, stream_context) : _conv_back(c, hint, trial_flags, inputs, input_size, outputs, output_size, stream_context);
			HIP_ENFORCE(hipEventRecord(e1, stream));
			HIP_ENFORCE(hipEventSynchronize(e1));
			if (ret != CCV_NNC_EXEC_SUCCESS) ok = 0;
			else HIP_ENFORCE(hipEventElapsedTime(&ms, e0, e1));
		}
		if (ok && (!have_best || ms < best_ms)) { best = algo; best_ms = ms; have_best = 1; }
	}
	HIP_ENFORCE(hipEventDestroy(e0));
	HIP_ENFORCE(hipEventDestroy(e1));
	return best;
}

} // namespace

extern "C" void _register_command_CCV_NNC_CONVOLUTION_FORWARD_backend_CCV_NNC_BACKEND_GPU_CUDNN(ccv_nnc_cmd_backend_registry_t* const registry)
{
	registry->tensor_formats = CCV_TENSOR_FORMAT_NCHW | CCV_TENSOR_FORMAT_NHWC;
	registry->tensor_datatypes = CCV_32F | CCV_16F;
	registry->tensor_memory = CCV_TENSOR_GPU_MEMORY;
	registry->algorithms = CONV_ALGO_COUNT;
	registry->exec = _conv_forw_any;
	registry->autotune = _conv_autotune;
	NNC_DEPALETTIZED(registry, _conv_forw_any); // palettized filters (ccv_nnc_conv_gpu_cudnn.cu:79-95)
}

extern "C" void _register_command_CCV_NNC_CONVOLUTION_BACKWARD_backend_CCV_NNC_BACKEND_GPU_CUDNN(ccv_nnc_cmd_backend_registry_t* const registry)
{
	registry->tensor_formats = CCV_TENSOR_FORMAT_NCHW | CCV_TENSOR_FORMAT_NHWC;
	registry->tensor_datatypes = CCV_32F | CCV_16F;
	registry->tensor_memory = CCV_TENSOR_GPU_MEMORY;
	registry->algorithms = CONV_ALGO_COUNT; // one choice for both gradients
	registry->exec = _conv_back_any;
	registry->autotune = _conv_autotune;
	NNC_DEPALETTIZED(registry, _conv_back_any); // palettized filters under the data gradient (ccv_nnc_conv_gpu_cudnn.cu:328-345)
}

extern "C" void _register_command_CCV_NNC_CONVOLUTION_TRANSPOSE_FORWARD_backend_CCV_NNC_BACKEND_GPU_CUDNN(ccv_nnc_cmd_backend_registry_t* const registry)
{
	registry->tensor_formats = CCV_TENSOR_FORMAT_NCHW | CCV_TENSOR_FORMAT_NHWC;
	registry->tensor_datatypes = CCV_32F;
	registry->tensor_memory = CCV_TENSOR_GPU_MEMORY;
	registry->algorithms = 1;
	registry->exec = _conv_transpose_forw;
	NNC_HALF_STAGED(registry, _conv_transpose_forw);
	NNC_DEPALETTIZED(registry, nnc::half_staged<_conv_transpose_forw>); // palettized filters (ccv_nnc_conv_transpose_gpu_cudnn.cu:72-90)
}
