// CCV_NNC_CONVOLUTION_FORWARD / BACKWARD on gfx950: implicit-GEMM on the fp32 MFMA contraction core.
// Semantics follow the reference CPU backend (the oracle):
//   forward   lib/nnc/cmd/convolution/ccv_nnc_conv_cpu_ref.c:13-172   b = bias + sum_{i,j,c} w[k,i,j,c] * a[n, y*s-p+i*d, x*s-p+j*d, g*Cg+c]
//   backward  lib/nnc/cmd/convolution/ccv_nnc_conv_cpu_ref.c:174-345  inputs (g, a, w) -> outputs (h, dw, dbias); CCV_NNC_ACCUMULATE_OUTPUT
//             accumulates into dw / dbias (:186-192); h is always overwritten (:286)
// and replace the cuDNN calls of lib/nnc/cmd/convolution/gpu/ccv_nnc_conv_gpu_cudnn.cu:24-114, 204-367.
// Layout on device: NHWC activations (n, y, x, c) with c innermost, weights [K][kh][kw][Cg]; tensor views are accepted for
// the inputs as long as the channel stride is 1.  NCHW tensors are routed through the layout kernels of cmd_util (workspace).
#include "gemm_launch.h"

using namespace nnc;

namespace {

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

static int conv_forw_nhwc(const conv_geom_t& g, const Image4& a, const float* w, const float* bias, const Image4& b, const int flags, ccv_nnc_stream_context_t* const ctx)
{
	if (a.sc != 1 || !pixel_linear(b) || !image_fits_int(a)) return CCV_NNC_EXEC_INVALID;
	const long M = (long)g.N * g.OH * g.OW;
	const int Kred = g.kh * g.kw * g.Cg;
	if (M > 0x7fffffffL) return CCV_NNC_EXEC_INVALID;
	const bool vec = (g.Cg % 4 == 0) && aligned16(a.p) && aligned16(w) && a.sw % 4 == 0 && a.sh % 4 == 0 && (a.n == 1 || a.sn % 4 == 0);
	GemmOut out = { b.p, b.sw, 1, bias, 1.f, 0 };
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
static int conv_dgrad_nhwc(const conv_geom_t& g, const Image4& gr, const float* w, const Image4& h, const int flags, ccv_nnc_stream_context_t* const ctx)
{
	if (gr.sc != 1 || !pixel_linear(h) || !image_fits_int(gr)) return CCV_NNC_EXEC_INVALID;
	const long M = (long)g.N * g.H * g.W;
	const int Kred = g.kh * g.kw * g.Kg;
	if (M > 0x7fffffffL) return CCV_NNC_EXEC_INVALID;
	const bool vec = (g.Kg % 4 == 0) && (g.Cg % 4 == 0) && aligned16(gr.p) && aligned16(w) && gr.sw % 4 == 0 && gr.sh % 4 == 0 && (gr.n == 1 || gr.sn % 4 == 0);
	GemmOut out = { h.p, h.sw, 1, 0, 1.f, 0 };
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
static int conv_wgrad_nhwc(const conv_geom_t& g, const Image4& gr, const Image4& a, float* dw, const int flags, ccv_nnc_stream_context_t* const ctx)
{
	if (a.sc != 1 || !pixel_linear(gr) || !image_fits_int(a)) return CCV_NNC_EXEC_INVALID;
	const long P = (long)g.N * g.OH * g.OW;
	if (P > 0x7fffffffL) return CCV_NNC_EXEC_INVALID;
	const int NN = g.kh * g.kw * g.Cg;
	const bool vec = (g.Cg % 4 == 0) && (g.Kg % 4 == 0) && aligned16(a.p) && aligned16(gr.p) && a.sw % 4 == 0 && a.sh % 4 == 0 && (a.n == 1 || a.sn % 4 == 0) && gr.sw % 4 == 0;
	GemmOut out = { dw, (long)NN, 1, 0, 1.f, (flags & CCV_NNC_ACCUMULATE_OUTPUT) ? 1 : 0 };
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

// ---- layout staging ---------------------------------------------------------------------------------------------------
// The kernels read NHWC activations and [K][kh][kw][Cg] weights.  NCHW activations (the reference's ResNet trainer) and
// NCHW-format weights [K][Cg][kh][kw] (what the reference's GPU tests hand over, test/int/nnc/cudnn.tests.c:50,65) are
// re-laid-out through the stream workspace by the tiled transpose of cmd_util.cpp: one extra read + write of the tensor,
// HBM-bound, instead of a second family of gather kernels whose channel-strided loads could not be 16-byte vectors.
static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

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
	Image4 ai, bi;
	if (!image4(a, &ai) || !image4(b, &bi)) return CCV_NNC_EXEC_INVALID;
	int K, kh, kw, Cg;
	if (!weights_shape(w, &K, &kh, &kw, &Cg)) return CCV_NNC_EXEC_INVALID;
	conv_geom_t g;
	if (!conv_geometry(cmd, hint, ai, bi, 0, &g) || K != g.K || kh != g.kh || kw != g.kw || Cg != g.Cg) return CCV_NNC_EXEC_INVALID;
	if (bias && (bias->info.dim[0] != g.K || !tensor_contiguous(bias))) return CCV_NNC_EXEC_INVALID;
	const bool stage_io = a->info.format == CCV_TENSOR_FORMAT_NCHW, stage_w = w->info.format == CCV_TENSOR_FORMAT_NCHW;
	if (!stage_io && !stage_w) return conv_forw_nhwc(g, ai, w->data.f32, bias ? bias->data.f32 : 0, bi, flags, stream_context);
	const size_t na = stage_io ? align256(sizeof(float) * tensor_count(a->info)) : 0, nb = stage_io ? align256(sizeof(float) * tensor_count(b->info)) : 0;
	const size_t nw = stage_w ? align256(sizeof(float) * tensor_count(w->info)) : 0;
	WorkspaceScope ws(stream_context, na + nb + nw, gemm_workspace_bound((long)g.N * g.OH * g.OW, g.Kg, (long)g.kh * g.kw * g.Cg));
	char* p = (char*)ws.prefix();
	if (!p) return CCV_NNC_EXEC_OOM;
	int ret;
	const float* wp = w->data.f32;
	if (stage_w) {
		if ((ret = weights_nchw_to_nhwc(w->data.f32, (float*)(p + na + nb), g.K, g.Cg, g.kh * g.kw, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
		wp = (const float*)(p + na + nb);
	}
	if (!stage_io) return conv_forw_nhwc(g, ai, wp, bias ? bias->data.f32 : 0, bi, flags, stream_context);
	ccv_nnc_tensor_t at, bt;
	dense_nhwc_like(a, ai, (float*)p, &at);
	dense_nhwc_like(b, bi, (float*)(p + na), &bt);
	if ((ret = format_transform(a, &at, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
	Image4 as, bs;
	image4(&at, &as); image4(&bt, &bs);
	if ((ret = conv_forw_nhwc(g, as, wp, bias ? bias->data.f32 : 0, bs, flags, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
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
	if (dw) {
		float* dwp = dw->data.f32;
		if (stage_w) {
			dwp = (float*)(p + ng + na + nh + nw);
			if (acc && (ret = weights_nchw_to_nhwc(dw->data.f32, dwp, g.K, g.Cg, g.kh * g.kw, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
		}
		if ((ret = conv_wgrad_nhwc(g, gim, aim, dwp, flags, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
		if (stage_w && (ret = weights_nhwc_to_nchw(dwp, dw->data.f32, g.K, g.Cg, g.kh * g.kw, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
	}
	if (dbias) {
		if (!pixel_linear(gim)) return CCV_NNC_EXEC_INVALID;
		if ((ret = colsum_f32(gim.p, P, g.K, gim.sw, dbias->data.f32, acc, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
	}
	if (h) {
		const float* wp = w->data.f32;
		if (stage_w) {
			if ((ret = weights_nchw_to_nhwc(w->data.f32, (float*)(p + ng + na + nh), g.K, g.Cg, g.kh * g.kw, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
			wp = (const float*)(p + ng + na + nh);
		}
		if ((ret = conv_dgrad_nhwc(g, gim, wp, him, flags, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
		if (stage_io && (ret = format_transform(&hs, h, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
	}
	return CCV_NNC_EXEC_SUCCESS;
}

} // namespace

extern "C" void _register_command_CCV_NNC_CONVOLUTION_FORWARD_backend_CCV_NNC_BACKEND_GPU_CUDNN(ccv_nnc_cmd_backend_registry_t* const registry)
{
	registry->tensor_formats = CCV_TENSOR_FORMAT_NCHW | CCV_TENSOR_FORMAT_NHWC;
	registry->tensor_datatypes = CCV_32F;
	registry->tensor_memory = CCV_TENSOR_GPU_MEMORY;
	registry->algorithms = 1;
	registry->exec = _conv_forw;
}

extern "C" void _register_command_CCV_NNC_CONVOLUTION_BACKWARD_backend_CCV_NNC_BACKEND_GPU_CUDNN(ccv_nnc_cmd_backend_registry_t* const registry)
{
	registry->tensor_formats = CCV_TENSOR_FORMAT_NCHW | CCV_TENSOR_FORMAT_NHWC;
	registry->tensor_datatypes = CCV_32F;
	registry->tensor_memory = CCV_TENSOR_GPU_MEMORY;
	registry->algorithms = 1;
	registry->exec = _conv_back;
}
