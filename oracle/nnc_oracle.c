/*
 * nnc_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C CPU restatement of the reference's algorithm for every command on the hot path,
 * written from the semantics of the reference's CPU_REF backend; each function cites the reference file:line it follows.
 * It is the checker the parity tests fall back to where the reference's own build (oracle/_ref/libccv_ref.so) cannot
 * travel, and it is itself pinned against that build and against the reference tests' known answers by
 * tests/test_oracle_pin.py (parity is therefore PINNED, see DESIGN.md "Oracle").
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.  The product
 * (ccv_amd/csrc, libnnc_mi355x.so) never links, loads or calls it.
 *
 * Interface: nnc_oracle_cmd_exec() has the signature of ccv_nnc_cmd_exec (lib/nnc/ccv_nnc_cmd.c:651) over the ABI
 * mirrors of include/nnc_mi355x.h; tensors are dense CPU tensors (tensor views are accepted when contiguous).
 * Loops are deliberately naive: sequential float accumulation in the reference's summation order wherever the order is
 * observable in the last bits (pool / relu / sgd / softmax are compared bit-exactly or within a few ulp).
 *
 * Unlike the reference's CPU pool / CPU_OPT conv kernels (which only process image 0 of a 4-d batch),
 * every function here processes the whole batch -- the semantics the GPU backend being replaced has.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "../include/nnc_mi355x.h"

static int nd_of(const int* dim)
{
	int i;
	for (i = 0; i < CCV_NNC_MAX_DIM_ALLOC && dim[i] > 0; i++) ;
	return i;
}
static size_t count_of(const ccv_nnc_tensor_t* t)
{
	size_t c = 1;
	int i;
	if (t->info.dim[0] == 0) return 0;
	for (i = 0; i < CCV_NNC_MAX_DIM_ALLOC && t->info.dim[i] > 0; i++) c *= (size_t)t->info.dim[i];
	return c;
}
/* (n, h, w, c) extents and element strides of a 3-d / 4-d image tensor in either layout (ccv_nnc_internal.h:44-55). */
typedef struct { float* p; int n, h, w, c; long sn, sh, sw, sc; } img_t;
static int img_of(const ccv_nnc_tensor_t* t, img_t* o)
{
	const int nd = nd_of(t->info.dim), b = nd == 4;
	const int* d = t->info.dim;
	if (nd != 3 && nd != 4) return 0;
	o->p = t->data.f32;
	o->n = b ? d[0] : 1;
	if (t->info.format == CCV_TENSOR_FORMAT_NHWC) {
		o->h = d[b]; o->w = d[b + 1]; o->c = d[b + 2];
		o->sc = 1; o->sw = o->c; o->sh = (long)o->w * o->c;
	} else if (t->info.format == CCV_TENSOR_FORMAT_NCHW) {
		o->c = d[b]; o->h = d[b + 1]; o->w = d[b + 2];
		o->sw = 1; o->sh = o->w; o->sc = (long)o->h * o->w;
	} else return 0;
	o->sn = (long)o->h * o->w * o->c;
	return 1;
}
#define AT(im, n_, y_, x_, c_) ((im).p[(n_) * (im).sn + (y_) * (im).sh + (x_) * (im).sw + (c_) * (im).sc])

/* ------------------------------------------------------------------------------------------------ convolution */
/* Forward: lib/nnc/cmd/convolution/ccv_nnc_conv_cpu_ref.c:13-172.  b[n,y,x,k] = bias[k] + sum_{i,j,c} w[k,i,j,c] *
 * a[n, y*s - p + i*d, x*s - p + j*d, g*Cg + c], taps that fall outside the image are skipped (border clipping,
 * ccv_nnc_internal.h:209-213).  Weights: NHWC [K][kh][kw][Cg] (:44-106), NCHW [K][Cg][kh][kw] (:107-170). */
static long w_index(int format, int k, int i, int j, int c, int kh, int kw, int cg)
{
	return format == CCV_TENSOR_FORMAT_NCHW ? (((long)k * cg + c) * kh + i) * kw + j : (((long)k * kh + i) * kw + j) * cg + c;
}
static int conv_forw(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, ccv_nnc_tensor_t* const* in, int nin, ccv_nnc_tensor_t* const* out)
{
	img_t a, b;
	const ccv_nnc_tensor_t* w = in[1];
	const float* bias = nin > 2 && in[2] ? in[2]->data.f32 : 0;
	const int groups = cmd.info.convolution.groups > 0 ? cmd.info.convolution.groups : 1;
	const int kh = cmd.info.size.dim[0], kw = cmd.info.size.dim[1];
	const int sy = hint.stride.dim[0] > 0 ? hint.stride.dim[0] : 1, sx = hint.stride.dim[1] > 0 ? hint.stride.dim[1] : 1;
	const int py = hint.border.begin[0], px = hint.border.begin[1];
	const int dy = cmd.info.convolution.dilation[0] > 1 ? cmd.info.convolution.dilation[0] : 1, dx = cmd.info.convolution.dilation[1] > 1 ? cmd.info.convolution.dilation[1] : 1;
	int n, y, x, k, i, j, c;
	if (!img_of(in[0], &a) || !img_of(out[0], &b) || a.n != b.n) return CCV_NNC_EXEC_INVALID;
	const int cg = a.c / groups, kg = b.c / groups;
	for (n = 0; n < a.n; n++)
		for (y = 0; y < b.h; y++)
			for (x = 0; x < b.w; x++)
				for (k = 0; k < b.c; k++) {
					const int g = k / kg;
					float v = bias ? bias[k] : 0.f;
					for (i = 0; i < kh; i++) {
						const int iy = y * sy - py + i * dy;
						if (iy < 0 || iy >= a.h) continue;
						for (j = 0; j < kw; j++) {
							const int ix = x * sx - px + j * dx;
							if (ix < 0 || ix >= a.w) continue;
							for (c = 0; c < cg; c++)
								v += w->data.f32[w_index(w->info.format, k, i, j, c, kh, kw, cg)] * AT(a, n, iy, ix, g * cg + c);
						}
					}
					AT(b, n, y, x, k) = v;
				}
	return CCV_NNC_EXEC_SUCCESS;
}
/* Backward: ccv_nnc_conv_cpu_ref.c:174-345 (NHWC only, :356-363).  inputs (g, a, w), outputs (h, dw, dbias);
 * dw[k,i,j,c] (+)= sum g*a, dbias[k] = sum g, h = sum_k g*w (zeroed first, :286).  CCV_NNC_ACCUMULATE_OUTPUT makes dw
 * accumulate (:186-192); dbias is OVERWRITTEN even then (:262-263) -- tests add the GPU-backend accumulate on top. */
static int conv_back(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, int flags, ccv_nnc_tensor_t* const* in, int nin, ccv_nnc_tensor_t* const* out, int nout)
{
	img_t g, a, h;
	const ccv_nnc_tensor_t* w = nin > 2 ? in[2] : 0;
	ccv_nnc_tensor_t* ht = out[0];
	ccv_nnc_tensor_t* dw = nout > 1 ? out[1] : 0;
	ccv_nnc_tensor_t* dbias = nout > 2 ? out[2] : 0;
	const int groups = cmd.info.convolution.groups > 0 ? cmd.info.convolution.groups : 1;
	const int kh = cmd.info.size.dim[0], kw = cmd.info.size.dim[1];
	const int sy = hint.stride.dim[0] > 0 ? hint.stride.dim[0] : 1, sx = hint.stride.dim[1] > 0 ? hint.stride.dim[1] : 1;
	const int py = hint.border.begin[0], px = hint.border.begin[1];
	const int dy = cmd.info.convolution.dilation[0] > 1 ? cmd.info.convolution.dilation[0] : 1, dx = cmd.info.convolution.dilation[1] > 1 ? cmd.info.convolution.dilation[1] : 1;
	int n, y, x, k, i, j, c;
	if (!img_of(in[0], &g)) return CCV_NNC_EXEC_INVALID;
	if (!img_of(in[1] ? in[1] : ht, &a)) return CCV_NNC_EXEC_INVALID;
	const int cg = a.c / groups, kg = g.c / groups;
	if (dw && !(flags & CCV_NNC_ACCUMULATE_OUTPUT)) memset(dw->data.f32, 0, sizeof(float) * count_of(dw));
	if (dbias) memset(dbias->data.f32, 0, sizeof(float) * count_of(dbias));
	if (ht) { if (!img_of(ht, &h)) return CCV_NNC_EXEC_INVALID; memset(h.p, 0, sizeof(float) * count_of(ht)); }
	for (n = 0; n < g.n; n++)
		for (y = 0; y < g.h; y++)
			for (x = 0; x < g.w; x++)
				for (k = 0; k < g.c; k++) {
					const float v = AT(g, n, y, x, k);
					const int gr = k / kg;
					if (dbias) dbias->data.f32[k] += v;
					for (i = 0; i < kh; i++) {
						const int iy = y * sy - py + i * dy;
						if (iy < 0 || iy >= a.h) continue;
						for (j = 0; j < kw; j++) {
							const int ix = x * sx - px + j * dx;
							if (ix < 0 || ix >= a.w) continue;
							for (c = 0; c < cg; c++) {
								const long wi = w_index(CCV_TENSOR_FORMAT_NHWC, k, i, j, c, kh, kw, cg);
								if (dw) dw->data.f32[wi] += v * AT(a, n, iy, ix, gr * cg + c);
								if (ht) AT(h, n, iy, ix, gr * cg + c) += v * w->data.f32[wi];
							}
						}
					}
				}
	return CCV_NNC_EXEC_SUCCESS;
}

/* ------------------------------------------------------------------------------------------------------- GEMM */
/* Matrix view of a tensor: trailing two dims = matrix, dim[nd-3] = batch, optional transpose of the two
 * (lib/nnc/ccv_nnc_easy.h:421-444 ccv_nnc_tensor_get_matrix_params). */
typedef struct { float* p; int batch, rows, cols; long binc, rinc, cinc; } mat_t;
static int mat_of(const ccv_nnc_tensor_t* t, const int* tr, mat_t* m)
{
	const int nd = nd_of(t->info.dim);
	const int* d = t->info.dim;
	if (nd < 1 || nd > 3) return 0;
	m->p = t->data.f32;
	m->batch = nd == 3 ? d[0] : 1;
	m->rows = nd == 1 ? 1 : d[nd - 2];
	m->cols = d[nd - 1];
	m->cinc = 1; m->rinc = m->cols; m->binc = nd == 3 ? (long)m->rows * m->cols : 0;
	if (tr && tr[0] != tr[1]) { int ti = m->rows; long tl = m->rinc; m->rows = m->cols; m->cols = ti; m->rinc = m->cinc; m->cinc = tl; }
	return 1;
}
#define MAT(m, b_, r_, c_) ((m).p[(b_) * (m).binc + (r_) * (m).rinc + (c_) * (m).cinc])
/* Forward: lib/nnc/cmd/blas/ccv_nnc_gemm_cpu_ref.c:110-184.  b = op(a) . op(w) + bias, batches broadcast when 1. */
static int gemm_forw(const ccv_nnc_cmd_t cmd, ccv_nnc_tensor_t* const* in, int nin, ccv_nnc_tensor_t* const* out)
{
	mat_t a, w, b, bias;
	int z, i, j, k;
	if (!mat_of(in[0], cmd.info.blas.transpose_a, &a) || !mat_of(in[1], cmd.info.blas.transpose_b, &w) || !mat_of(out[0], 0, &b)) return CCV_NNC_EXEC_INVALID;
	const int has_bias = nin > 2 && in[2] && mat_of(in[2], 0, &bias);
	if (a.cols != w.rows || a.rows != b.rows || w.cols != b.cols) return CCV_NNC_EXEC_INVALID;
	for (z = 0; z < b.batch; z++)
		for (i = 0; i < b.rows; i++)
			for (j = 0; j < b.cols; j++) {
				float v = has_bias ? MAT(bias, bias.batch > 1 ? z : 0, 0, j) : 0.f;
				for (k = 0; k < a.cols; k++) v += MAT(a, a.batch > 1 ? z : 0, i, k) * MAT(w, w.batch > 1 ? z : 0, k, j);
				MAT(b, z, i, j) = v;
			}
	return CCV_NNC_EXEC_SUCCESS;
}
/* Backward: ccv_nnc_gemm_cpu_ref.c:318-466 (dbias :186-221, dw :223-267, h :269-316).  inputs (g, a, w), outputs
 * (h, dw, dbias); a batch-1 output fed by a batched g sums over the batch; ACCUMULATE_OUTPUT adds into old contents. */
static int gemm_back(const ccv_nnc_cmd_t cmd, int flags, ccv_nnc_tensor_t* const* in, int nin, ccv_nnc_tensor_t* const* out, int nout)
{
	mat_t g, a, w, h, dw, db;
	const int acc = flags & CCV_NNC_ACCUMULATE_OUTPUT;
	int z, i, j, k;
	if (!mat_of(in[0], 0, &g)) return CCV_NNC_EXEC_INVALID;
	if (nout > 2 && out[2] && mat_of(out[2], 0, &db)) {
		if (!acc) memset(db.p, 0, sizeof(float) * count_of(out[2]));
		for (z = 0; z < g.batch; z++) for (i = 0; i < g.rows; i++) for (j = 0; j < g.cols; j++) MAT(db, db.batch > 1 ? z : 0, 0, j) += MAT(g, z, i, j);
	}
	if (nout > 1 && out[1]) {
		if (!mat_of(in[1], cmd.info.blas.transpose_a, &a) || !mat_of(out[1], cmd.info.blas.transpose_b, &dw)) return CCV_NNC_EXEC_INVALID;
		if (!acc) memset(dw.p, 0, sizeof(float) * count_of(out[1]));
		for (z = 0; z < g.batch; z++) for (k = 0; k < dw.rows; k++) for (j = 0; j < dw.cols; j++) {
			float v = 0.f;
			for (i = 0; i < g.rows; i++) v += MAT(a, a.batch > 1 ? z : 0, i, k) * MAT(g, z, i, j);
			MAT(dw, dw.batch > 1 ? z : 0, k, j) += v;
		}
	}
	if (out[0]) {
		if (nin < 3 || !mat_of(in[2], cmd.info.blas.transpose_b, &w) || !mat_of(out[0], cmd.info.blas.transpose_a, &h)) return CCV_NNC_EXEC_INVALID;
		if (!acc) memset(h.p, 0, sizeof(float) * count_of(out[0]));
		for (z = 0; z < g.batch; z++) for (i = 0; i < h.rows; i++) for (k = 0; k < h.cols; k++) {
			float v = 0.f;
			for (j = 0; j < g.cols; j++) v += MAT(g, z, i, j) * MAT(w, w.batch > 1 ? z : 0, k, j);
			MAT(h, h.batch > 1 ? z : 0, i, k) += v;
		}
	}
	return CCV_NNC_EXEC_SUCCESS;
}

/* ---------------------------------------------------------------------------------------------------- pooling */
/* Window of output (y, x): rows [y*s - p, y*s - p + k) clipped to the image -- the window SHRINKS at the borders, there
 * is no -inf / zero padding (SET_BORDER_OFFSET_SIZE_FOR, ccv_nnc_internal.h:209-213).  size 0,0 = whole map. */
static void pool_window(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const img_t* a, int y, int x, int* y0, int* y1, int* x0, int* x1)
{
	const int kh = cmd.info.size.dim[0] > 0 ? cmd.info.size.dim[0] : a->h, kw = cmd.info.size.dim[1] > 0 ? cmd.info.size.dim[1] : a->w;
	const int sy = hint.stride.dim[0] > 0 ? hint.stride.dim[0] : 1, sx = hint.stride.dim[1] > 0 ? hint.stride.dim[1] : 1;
	*y0 = y * sy - hint.border.begin[0]; *y1 = *y0 + kh;
	*x0 = x * sx - hint.border.begin[1]; *x1 = *x0 + kw;
	if (*y0 < 0) *y0 = 0;
	if (*x0 < 0) *x0 = 0;
	if (*y1 > a->h) *y1 = a->h;
	if (*x1 > a->w) *x1 = a->w;
}
/* max forward lib/nnc/cmd/pool/ccv_nnc_max_pool_cpu_ref.c:13-61; average forward ccv_nnc_avg_pool_cpu_ref.c:13-60
 * (divides by the CLIPPED element count, :46-53). */
static int pool_forw(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, int is_max, ccv_nnc_tensor_t* const* in, ccv_nnc_tensor_t* const* out)
{
	img_t a, b;
	int n, y, x, c, i, j, y0, y1, x0, x1;
	if (!img_of(in[0], &a) || !img_of(out[0], &b)) return CCV_NNC_EXEC_INVALID;
	for (n = 0; n < a.n; n++) for (y = 0; y < b.h; y++) for (x = 0; x < b.w; x++) {
		pool_window(cmd, hint, &a, y, x, &y0, &y1, &x0, &x1);
		for (c = 0; c < a.c; c++) {
			float v = is_max ? AT(a, n, y0, x0, c) : 0.f;
			for (i = y0; i < y1; i++) for (j = x0; j < x1; j++) {
				const float u = AT(a, n, i, j, c);
				if (is_max) { if (u > v) v = u; } else v += u;
			}
			AT(b, n, y, x, c) = is_max ? v : v / (float)((y1 - y0) * (x1 - x0));
		}
	}
	return CCV_NNC_EXEC_SUCCESS;
}
/* max backward ccv_nnc_max_pool_cpu_ref.c:63-141: inputs (g, a, b); h[p] += g[o] for EVERY p in the window with
 * a[p] == b[o] (:125-132), accumulated over overlapping windows in output raster order.
 * average backward ccv_nnc_avg_pool_cpu_ref.c:62-109: h[p] += g[o] / clipped count. */
static int pool_back(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, int is_max, ccv_nnc_tensor_t* const* in, ccv_nnc_tensor_t* const* out)
{
	img_t g, a, b, h;
	int n, y, x, c, i, j, y0, y1, x0, x1;
	if (!img_of(in[0], &g) || !img_of(out[0], &h)) return CCV_NNC_EXEC_INVALID;
	if (is_max && (!img_of(in[1], &a) || !img_of(in[2], &b))) return CCV_NNC_EXEC_INVALID;
	memset(h.p, 0, sizeof(float) * count_of(out[0]));
	for (n = 0; n < h.n; n++) for (y = 0; y < g.h; y++) for (x = 0; x < g.w; x++) {
		pool_window(cmd, hint, &h, y, x, &y0, &y1, &x0, &x1);
		for (c = 0; c < h.c; c++) {
			const float gv = AT(g, n, y, x, c);
			if (is_max) {
				const float bv = AT(b, n, y, x, c);
				for (i = y0; i < y1; i++) for (j = x0; j < x1; j++) if (AT(a, n, i, j, c) == bv) AT(h, n, i, j, c) += gv;
			} else {
				const float u = gv / (float)((y1 - y0) * (x1 - x0));
				for (i = y0; i < y1; i++) for (j = x0; j < x1; j++) AT(h, n, i, j, c) += u;
			}
		}
	}
	return CCV_NNC_EXEC_SUCCESS;
}

/* ----------------------------------------------------------------------------------------------- element-wise */
/* relu lib/nnc/cmd/relu/ccv_nnc_relu_cpu_ref.c:13-31 / :33-55: b = max(a, 0); inputs (g, _, b) -> h = b > 0 ? g : 0. */
static int relu_forw(ccv_nnc_tensor_t* const* in, ccv_nnc_tensor_t* const* out)
{
	size_t i, n = count_of(in[0]);
	for (i = 0; i < n; i++) out[0]->data.f32[i] = in[0]->data.f32[i] > 0 ? in[0]->data.f32[i] : 0;
	return CCV_NNC_EXEC_SUCCESS;
}
static int relu_back(ccv_nnc_tensor_t* const* in, ccv_nnc_tensor_t* const* out)
{
	size_t i, n = count_of(in[2]);
	for (i = 0; i < n; i++) out[0]->data.f32[i] = in[2]->data.f32[i] > 0 ? (in[0] ? in[0]->data.f32[i] : 1.f) : 0;
	return CCV_NNC_EXEC_SUCCESS;
}
/* ewsum lib/nnc/cmd/ew/ccv_nnc_ew_cpu_ref.c:15-233: left fold ((in0 + in1) + in2) + ... */
static int ewsum_forw(ccv_nnc_tensor_t* const* in, int nin, ccv_nnc_tensor_t* const* out)
{
	size_t i, n = count_of(out[0]);
	int k;
	for (i = 0; i < n; i++) {
		float v = in[0]->data.f32[i];
		for (k = 1; k < nin; k++) v += in[k]->data.f32[i];
		out[0]->data.f32[i] = v;
	}
	return CCV_NNC_EXEC_SUCCESS;
}
/* scalar mul lib/nnc/cmd/blas/ccv_nnc_mul_cpu_ref.c:417-430: b = a[0] * x. */
static int scalar_mul_forw(const ccv_nnc_cmd_t cmd, ccv_nnc_tensor_t* const* in, ccv_nnc_tensor_t* const* out)
{
	size_t i, n = count_of(in[0]);
	for (i = 0; i < n; i++) out[0]->data.f32[i] = cmd.info.blas.a[0] * in[0]->data.f32[i];
	return CCV_NNC_EXEC_SUCCESS;
}
/* add lib/nnc/cmd/blas/ccv_nnc_add_cpu_ref.c:16-300: c = p*a + q*b with numpy-style broadcast of size-1 dims
 * (a NULL b means c = p*a).  mul ccv_nnc_mul_cpu_ref.c:16-300: c = p * a * b, same broadcast. */
static int addmul_forw(const ccv_nnc_cmd_t cmd, int is_mul, ccv_nnc_tensor_t* const* in, int nin, ccv_nnc_tensor_t* const* out)
{
	const ccv_nnc_tensor_t* a = in[0];
	const ccv_nnc_tensor_t* b = nin > 1 ? in[1] : 0;
	ccv_nnc_tensor_t* c = out[0];
	const float p = cmd.info.blas.a[0], q = cmd.info.blas.a[1];
	const int nd = nd_of(c->info.dim);
	int ad[CCV_NNC_MAX_DIM_ALLOC], bd[CCV_NNC_MAX_DIM_ALLOC], idx[CCV_NNC_MAX_DIM_ALLOC] = { 0 };
	const int and_ = nd_of(a->info.dim), bnd = b ? nd_of(b->info.dim) : 0;
	size_t i, n = count_of(c);
	int k;
	for (k = 0; k < nd; k++) { /* right-align the operand shapes against c */
		ad[k] = k - (nd - and_) >= 0 ? a->info.dim[k - (nd - and_)] : 1;
		bd[k] = b && k - (nd - bnd) >= 0 ? b->info.dim[k - (nd - bnd)] : 1;
	}
	for (i = 0; i < n; i++) {
		size_t ai = 0, bi = 0;
		for (k = 0; k < nd; k++) { ai = ai * ad[k] + (ad[k] == 1 ? 0 : idx[k]); bi = bi * bd[k] + (bd[k] == 1 ? 0 : idx[k]); }
		if (is_mul) c->data.f32[i] = p * a->data.f32[ai] * (b ? b->data.f32[bi] : 1.f);
		else c->data.f32[i] = b ? p * a->data.f32[ai] + q * b->data.f32[bi] : p * a->data.f32[ai];
		for (k = nd - 1; k >= 0; k--) { if (++idx[k] < c->info.dim[k]) break; idx[k] = 0; }
	}
	return CCV_NNC_EXEC_SUCCESS;
}
/* add backward ccv_nnc_add_cpu_ref.c:200-300: da = p * g, db = q * g, each summed over the axes it was broadcast along;
 * mul backward ccv_nnc_mul_cpu_ref.c:192-415: inputs (g, a, b): da = p * g * b, db = p * g * a, summed likewise.
 * A NULL g stands for ones of the broadcast shape of (a, b). */
static int addmul_back(const ccv_nnc_cmd_t cmd, int is_mul, ccv_nnc_tensor_t* const* in, int nin, ccv_nnc_tensor_t* const* out, int nout)
{
	const ccv_nnc_tensor_t* g = in[0];
	int full[4], od[2][4], xd[2][4], idx[4];
	int k, o;
	const ccv_nnc_tensor_t* ops[2] = { nin > 1 ? in[1] : 0, nin > 2 ? in[2] : 0 };
	for (k = 0; k < 4; k++) full[k] = 1;
	for (o = 0; o < 2; o++) {
		const ccv_nnc_tensor_t* shp = (o < nout && out[o]) ? out[o] : ops[o];
		const int nd = shp ? nd_of(shp->info.dim) : 0;
		for (k = 0; k < 4; k++) { od[o][k] = shp && k - (4 - nd) >= 0 ? shp->info.dim[k - (4 - nd)] : 1; if (od[o][k] > full[k]) full[k] = od[o][k]; }
		(void)xd;
	}
	if (g) { const int nd = nd_of(g->info.dim); for (k = 0; k < 4; k++) { const int d = k - (4 - nd) >= 0 ? g->info.dim[k - (4 - nd)] : 1; if (d > full[k]) full[k] = d; } }
	for (o = 0; o < 2 && o < nout; o++) {
		const float p = is_mul ? cmd.info.blas.a[0] : cmd.info.blas.a[o];
		const ccv_nnc_tensor_t* other = ops[1 - o];
		size_t n = 1, i;
		if (!out[o]) continue;
		memset(out[o]->data.f32, 0, sizeof(float) * count_of(out[o]));
		for (k = 0; k < 4; k++) { n *= full[k]; idx[k] = 0; }
		for (i = 0; i < n; i++) {
			size_t oi = 0, xi = 0;
			for (k = 0; k < 4; k++) { oi = oi * od[o][k] + (od[o][k] == 1 ? 0 : idx[k]); xi = xi * od[1 - o][k] + (od[1 - o][k] == 1 ? 0 : idx[k]); }
			out[o]->data.f32[oi] += p * (g ? g->data.f32[i] : 1.f) * (is_mul ? other->data.f32[xi] : 1.f);
			for (k = 3; k >= 0; k--) { if (++idx[k] < full[k]) break; idx[k] = 0; }
		}
	}
	return CCV_NNC_EXEC_SUCCESS;
}
/* set / data transfer lib/nnc/cmd/util/ccv_nnc_util_cpu_ref.c:637-664 / :596-617. */
static int set_forw(const ccv_nnc_cmd_t cmd, ccv_nnc_tensor_t* const* out, int nout)
{
	int k;
	for (k = 0; k < nout; k++) if (out[k]) { size_t i, n = count_of(out[k]); for (i = 0; i < n; i++) out[k]->data.f32[i] = cmd.info.blas.a[0]; }
	return CCV_NNC_EXEC_SUCCESS;
}
static int transfer_forw(ccv_nnc_tensor_t* const* in, int nin, ccv_nnc_tensor_t* const* out, int nout)
{
	int k;
	for (k = 0; k < nin && k < nout; k++) if (in[k] && out[k] && in[k]->data.u8 != out[k]->data.u8) memcpy(out[k]->data.u8, in[k]->data.u8, sizeof(float) * count_of(in[k]));
	return CCV_NNC_EXEC_SUCCESS;
}
/* sgd lib/nnc/cmd/sgd/ccv_nnc_sgd_cpu_ref.c:16-126: inputs (g, a, m), outputs (b, n).
 *   plain:    n = mu*m + (1 - dampening)*(scale*g + decay*a);  b = a - rate*n
 *   nesterov: grad = scale*g; n = mu*m + grad + decay*a; b = a - rate*(grad + mu*n)      (:81-84) */
static int sgd_forw(const ccv_nnc_cmd_t cmd, ccv_nnc_tensor_t* const* in, ccv_nnc_tensor_t* const* out)
{
	const float rate = cmd.info.sgd.rate, scale = cmd.info.sgd.scale, decay = cmd.info.sgd.decay, mu = cmd.info.sgd.momentum, invd = 1.f - cmd.info.sgd.dampening;
	size_t i, n = count_of(in[0]);
	for (i = 0; i < n; i++) {
		const float a = in[1]->data.f32[i];
		if (cmd.info.sgd.nesterov) {
			float grad = scale * in[0]->data.f32[i];
			const float mom = mu * in[2]->data.f32[i] + grad + decay * a;
			out[1]->data.f32[i] = mom;
			grad += mu * mom;
			out[0]->data.f32[i] = a - rate * grad;
		} else {
			const float mom = mu * in[2]->data.f32[i] + invd * (scale * in[0]->data.f32[i] + decay * a);
			out[1]->data.f32[i] = mom;
			out[0]->data.f32[i] = a - rate * mom;
		}
	}
	return CCV_NNC_EXEC_SUCCESS;
}

/* ------------------------------------------------------------------------------------- softmax cross-entropy */
/* Forward lib/nnc/cmd/softmax_loss/ccv_nnc_softmax_crossentropy_cpu_ref.c:13-181: inputs (a [N x C], label), outputs
 * (loss [N] optional, softmax [N x C]).  The "loss" is sum_j t_j * (max - a_j) with t = one-hot(label) (fp32 index
 * rounded +0.5, or int32), smoothed one-hot (trim0 off-label, trim1 on-label) or a dense [N x C] target (:55,:129);
 * softmax = exp(a - max) / sum, the sum accumulated in double (:56-61). */
static int smce_forw(const ccv_nnc_cmd_t cmd, ccv_nnc_tensor_t* const* in, ccv_nnc_tensor_t* const* out)
{
	const ccv_nnc_tensor_t* a = in[0];
	const ccv_nnc_tensor_t* b = in[1];
	ccv_nnc_tensor_t* c = out[0];
	ccv_nnc_tensor_t* d = out[1];
	const int batch = nd_of(a->info.dim) < 2 ? 1 : a->info.dim[0];
	const int count = (int)(count_of(a) / batch);
	const float t0 = cmd.info.label_smoothing.trim0, t1 = cmd.info.label_smoothing.trim1;
	int i, j;
	for (i = 0; i < batch; i++) {
		const float* ap = a->data.f32 + (size_t)i * count;
		float* dp = d->data.f32 + (size_t)i * count;
		double maxval = ap[0], sum = 0;
		for (j = 1; j < count; j++) if (ap[j] > maxval) maxval = ap[j];
		if (c) {
			const int is_int = CCV_GET_DATA_TYPE(b->info.datatype) == CCV_32S;
			const int dense = !is_int && count_of(b) == count_of(a) && count > 1;
			if (dense) {
				float p = 0;
				for (j = 0; j < count; j++) p += b->data.f32[(size_t)i * count + j] * (float)(maxval - ap[j]);
				c->data.f32[i] = p;
			} else {
				const int label = is_int ? b->data.i32[i] : (int)(b->data.f32[i] + 0.5);
				if (t0 == 0 && t1 == 1) c->data.f32[i] = (float)(maxval - ap[label]);
				else {
					float p = 0;
					for (j = 0; j < count; j++) p += (j == label ? t1 : t0) * (float)(maxval - ap[j]);
					c->data.f32[i] = p;
				}
			}
		}
		for (j = 0; j < count; j++) sum += (dp[j] = expf(ap[j] - (float)maxval));
		sum = 1.0 / sum;
		for (j = 0; j < count; j++) dp[j] = (float)(dp[j] * sum);
	}
	return CCV_NNC_EXEC_SUCCESS;
}
/* Backward :183-360: inputs (g [N] or NULL => ones, _, _, label, _, softmax d), output h = g * (d - target). */
static int smce_back(const ccv_nnc_cmd_t cmd, ccv_nnc_tensor_t* const* in, int nin, ccv_nnc_tensor_t* const* out)
{
	const ccv_nnc_tensor_t* g = in[0];
	const ccv_nnc_tensor_t* b = in[3];
	const ccv_nnc_tensor_t* d = in[5];
	ccv_nnc_tensor_t* h = out[0];
	const int batch = nd_of(d->info.dim) < 2 ? 1 : d->info.dim[0];
	const int count = (int)(count_of(d) / batch);
	const float t0 = cmd.info.label_smoothing.trim0, t1 = cmd.info.label_smoothing.trim1;
	const int is_int = CCV_GET_DATA_TYPE(b->info.datatype) == CCV_32S;
	const int dense = !is_int && count_of(b) == count_of(d) && count > 1;
	int i, j;
	for (i = 0; i < batch; i++) {
		const float gv = g ? g->data.f32[i] : 1.f;
		const int label = dense ? -1 : is_int ? b->data.i32[i] : (int)(b->data.f32[i] + 0.5);
		for (j = 0; j < count; j++) {
			const float t = dense ? b->data.f32[(size_t)i * count + j] : (j == label ? t1 : t0);
			h->data.f32[(size_t)i * count + j] = gv * (d->data.f32[(size_t)i * count + j] - t);
		}
	}
	(void)nin;
	return CCV_NNC_EXEC_SUCCESS;
}

/* ------------------------------------------------------------------------------------------------- batch norm */
/* Statistics tensors have extent 1 on every reduced axis.  Index of the statistic that element `idx` of x maps to. */
static size_t bn_stat_index(const ccv_nnc_tensor_t* x, const ccv_nnc_tensor_t* s, size_t idx)
{
	const int nd = nd_of(x->info.dim);
	size_t r = 0, mul = 1;
	int k;
	for (k = nd - 1; k >= 0; k--) {
		const size_t i = idx % x->info.dim[k];
		idx /= x->info.dim[k];
		if (s->info.dim[k] != 1) { r += i * mul; mul *= s->info.dim[k]; }
	}
	return r;
}
/* Forward lib/nnc/cmd/norm/ccv_nnc_batch_norm_cpu_ref.c:16-297: inputs (x, scale, bias, mean, var), outputs
 * (y, mean, var, saved_mean, saved_inv_std) with running mean / var updated IN PLACE (ccv_nnc_norm.c:19-26).
 * train (:44-232): mean_b = sum x / B; var_b = sum (x - mean_b)^2 / B (biased); running = m*running + (1-m)*batch;
 * inv_std = 1/sqrt(var_b + eps); y = x*(inv_std*scale) + (bias - mean_b*inv_std*scale).  test (:233-): the same affine
 * form with the running statistics, except that the reference divides by (sqrt(var) + eps), eps outside the root (:277). */
static int bnorm_forw(const ccv_nnc_cmd_t cmd, ccv_nnc_tensor_t* const* in, ccv_nnc_tensor_t* const* out, int nout)
{
	const ccv_nnc_tensor_t* x = in[0];
	const float* scale = in[1]->data.f32;
	const float* bias = in[2]->data.f32;
	float* mean = in[3]->data.f32;
	float* var = in[4]->data.f32;
	float* y = out[0]->data.f32;
	const size_t n = count_of(x), rc = count_of(in[1]);
	const float eps = cmd.info.bnorm.epsilon, m = cmd.info.bnorm.momentum;
	size_t i;
	float* nscale = (float*)malloc(sizeof(float) * rc * 2);
	float* nbias = nscale + rc;
	if (!cmd.info.bnorm.is_test) {
		float* smean = out[3]->data.f32;
		float* sistd = out[4]->data.f32;
		const float inv_b = 1.f / (float)(n / rc);
		if (nout < 5) { free(nscale); return CCV_NNC_EXEC_INVALID; }
		for (i = 0; i < rc; i++) smean[i] = 0, sistd[i] = 0;
		for (i = 0; i < n; i++) smean[bn_stat_index(x, in[1], i)] += x->data.f32[i];
		for (i = 0; i < rc; i++) { smean[i] = inv_b * smean[i]; mean[i] = m * mean[i] + (1.f - m) * smean[i]; }
		for (i = 0; i < n; i++) { const size_t r = bn_stat_index(x, in[1], i); const float w = x->data.f32[i] - smean[r]; sistd[r] += w * w; }
		for (i = 0; i < rc; i++) { sistd[i] = inv_b * sistd[i]; var[i] = m * var[i] + (1.f - m) * sistd[i]; sistd[i] = 1.f / sqrtf(sistd[i] + eps); }
		for (i = 0; i < rc; i++) { nscale[i] = sistd[i] * scale[i]; nbias[i] = bias[i] - smean[i] * nscale[i]; }
	} else
		for (i = 0; i < rc; i++) { nscale[i] = scale[i] / (sqrtf(var[i]) + eps); nbias[i] = bias[i] - mean[i] * nscale[i]; } /* sic: eps OUTSIDE the root in test mode (:277) */
	for (i = 0; i < n; i++) { const size_t r = bn_stat_index(x, in[1], i); y[i] = x->data.f32[i] * nscale[r] + nbias[r]; }
	free(nscale);
	return CCV_NNC_EXEC_SUCCESS;
}
/* Backward :300-470: inputs 0 (g), 5 (x), 6 (scale), 13 (saved_mean), 14 (saved_inv_std) of 15; outputs (h, dscale, dbias).
 * dbias = sum g; xhat = (x - mean)*inv_std; dscale = sum xhat*g; h = (scale*inv_std/B) * (B*g - dbias - xhat*dscale). */
static int bnorm_back(ccv_nnc_tensor_t* const* in, ccv_nnc_tensor_t* const* out)
{
	const ccv_nnc_tensor_t* g = in[0];
	const ccv_nnc_tensor_t* x = in[5];
	const float* scale = in[6]->data.f32;
	const float* smean = in[13]->data.f32;
	const float* sistd = in[14]->data.f32;
	float* h = out[0]->data.f32;
	float* dscale = out[1]->data.f32;
	float* dbias = out[2]->data.f32;
	const size_t n = count_of(x), rc = count_of(in[6]);
	const float B = (float)(n / rc);
	size_t i;
	for (i = 0; i < rc; i++) dscale[i] = 0, dbias[i] = 0;
	for (i = 0; i < n; i++) dbias[bn_stat_index(x, in[6], i)] += g->data.f32[i];
	for (i = 0; i < n; i++) { const size_t r = bn_stat_index(x, in[6], i); dscale[r] += (x->data.f32[i] - smean[r]) * sistd[r] * g->data.f32[i]; }
	for (i = 0; i < n; i++) {
		const size_t r = bn_stat_index(x, in[6], i);
		const float xhat = (x->data.f32[i] - smean[r]) * sistd[r];
		h[i] = (1.f / B * scale[r] * sistd[r]) * (B * g->data.f32[i] - dbias[r] - xhat * dscale[r]);
	}
	return CCV_NNC_EXEC_SUCCESS;
}

/* --------------------------------------------------------------------------------------------------- dispatch */
int nnc_oracle_cmd_exec(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	(void)stream_context;
	switch (cmd.cmd) {
		case CCV_NNC_NOOP: return CCV_NNC_EXEC_SUCCESS;
		case CCV_NNC_CONVOLUTION_FORWARD: return conv_forw(cmd, hint, inputs, input_size, outputs);
		case CCV_NNC_CONVOLUTION_BACKWARD: return conv_back(cmd, hint, flags, inputs, input_size, outputs, output_size);
		case CCV_NNC_GEMM_FORWARD: return gemm_forw(cmd, inputs, input_size, outputs);
		case CCV_NNC_GEMM_BACKWARD: return gemm_back(cmd, flags, inputs, input_size, outputs, output_size);
		case CCV_NNC_MAX_POOL_FORWARD: return pool_forw(cmd, hint, 1, inputs, outputs);
		case CCV_NNC_AVERAGE_POOL_FORWARD: return pool_forw(cmd, hint, 0, inputs, outputs);
		case CCV_NNC_MAX_POOL_BACKWARD: return pool_back(cmd, hint, 1, inputs, outputs);
		case CCV_NNC_AVERAGE_POOL_BACKWARD: return pool_back(cmd, hint, 0, inputs, outputs);
		case CCV_NNC_RELU_FORWARD: return relu_forw(inputs, outputs);
		case CCV_NNC_RELU_BACKWARD: return relu_back(inputs, outputs);
		case CCV_NNC_EWSUM_FORWARD: return ewsum_forw(inputs, input_size, outputs);
		case CCV_NNC_SCALAR_MUL_FORWARD: return scalar_mul_forw(cmd, inputs, outputs);
		case CCV_NNC_ADD_FORWARD: return addmul_forw(cmd, 0, inputs, input_size, outputs);
		case CCV_NNC_MUL_FORWARD: return addmul_forw(cmd, 1, inputs, input_size, outputs);
		case CCV_NNC_ADD_BACKWARD: return addmul_back(cmd, 0, inputs, input_size, outputs, output_size);
		case CCV_NNC_MUL_BACKWARD: return addmul_back(cmd, 1, inputs, input_size, outputs, output_size);
		case CCV_NNC_SET_FORWARD: case CCV_NNC_SET_BACKWARD: return set_forw(cmd, outputs, output_size);
		case CCV_NNC_DATA_TRANSFER_FORWARD: case CCV_NNC_DATA_TRANSFER_BACKWARD: return transfer_forw(inputs, input_size, outputs, output_size);
		case CCV_NNC_SGD_FORWARD: return sgd_forw(cmd, inputs, outputs);
		case CCV_NNC_SOFTMAX_CROSSENTROPY_FORWARD: return smce_forw(cmd, inputs, outputs);
		case CCV_NNC_SOFTMAX_CROSSENTROPY_BACKWARD: return smce_back(cmd, inputs, input_size, outputs);
		case CCV_NNC_BATCH_NORM_FORWARD: return bnorm_forw(cmd, inputs, outputs, output_size);
		case CCV_NNC_BATCH_NORM_BACKWARD: return bnorm_back(inputs, outputs);
	}
	return CCV_NNC_EXEC_NO_KERNEL;
}
const char* nnc_oracle_version(void) { return "nnc-oracle 0.1 (plain C restatement of lib/nnc CPU_REF; test infrastructure)"; }
