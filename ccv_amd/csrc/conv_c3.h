// First-layer convolution: 3x3, THREE input channels (VGG-D conv1_1, BASELINE config 1; round 4: any stride -- ResNet-50's stride-2 stem layer), forward and filter gradient.
// On the general contraction kernel its 27-deep reduction is padded to a 32-deep K-step on 32x32x2 MFMAs and the wide
// output tile is written behind a long epilogue: 19 TFLOP/s, 2.3 ms for a 3.26 GB store at batch 256 (1.4 TB/s).  These kernels
// are shaped by the tensor instead:
//   * the 27 inputs of an output pixel are three runs of 9 contiguous floats (3 pixels x 3 channels of NHWC rows), so the
//     A fragment of v_mfma_f32_16x16x4_f32 -- lane (pixel = l & 15, k = 4 s + (l >> 4)) -- is seven scalar loads per lane out
//     of lines the neighbouring lanes touch as well (L1), no staging;
//   * the filters (27 x K floats) live in registers for the whole kernel, as B fragments with output channel N n + j in
//     column n of channel tile j, so that a lane ends up with N CONSECUTIVE channels of a pixel and stores 16 bytes;
//   * a wave turns 16 consecutive pixels of an output row into 16 x K outputs with 7 (k steps) x K / 16 MFMAs and four 16-byte
//     stores per lane: the kernel is bound by the output store (forward) or the read of the output gradient (filter gradient).
#pragma once
#include "mfma_gemm.h"

namespace nnc {

struct ConvC3Args {
	const float* a;   // [N][H][W][3]
	const float* w;   // [K][3][3][3]
	const float* bias;
	float* b;         // forward: output [N][OH][OW][K];  filter gradient: the output GRADIENT (read)
	long a_sn, a_sh, b_sn, b_sh, b_sw; // element strides (a: pixel stride 3)
	int N, H, W, OH, OW, K;
	int pad_y, pad_x;
	int sy, sx;                 // stride (1 or more): the patch of output pixel (oy, ox) starts at input (oy * sy - pad_y, ox * sx - pad_x)
	int groups_per_row, groups; // 16-pixel groups per output row, total
	int relu;                   // forward: write max(0, .) (NNC_MI355X_CONV_ALGO_FUSE_RELU)
	unsigned b_image_bytes;     // forward: span of one output image in bytes (range of the per-image store descriptor; host-checked < 2^31)
	FastDiv d_gpr, d_oh;        // group -> (row, position), row -> (image, oy) without hardware division (two per 28-MFMA group otherwise)
};

// element k (0..26; 27 = padding) of the patch of output pixel (oy, ox): row dy = k / 9, offset k % 9 into the 9-float run
__device__ __forceinline__ float convc3_patch(const float* const img, const ConvC3Args& g, const int oy, const int ox, const int k)
{
	const int dy = k / 9, r9 = k - dy * 9, dx = r9 / 3;
	const int iy = oy * g.sy - g.pad_y + dy, ix = ox * g.sx - g.pad_x + dx;
	const bool ok = (k < 27) & (iy >= 0) & (iy < g.H) & (ix >= 0) & (ix < g.W);
	return ok ? img[(long)iy * g.a_sh + (ox * g.sx - g.pad_x) * 3 + r9] : 0.f;
}

// NT = K / 16 (1, 2 or 4 channel tiles).  grid: as many workgroups as fit the chip; waves stride over the pixel groups.
template <int NT>
static __global__ void __launch_bounds__(256) conv3x3_c3_fwd_kernel(const ConvC3Args g)
{
	const int lane = threadIdx.x & 63, n = lane & 15, kq = lane >> 4;
	float wf[7][NT]; // B fragments: k = 4 s + kq, channel NT * n + j
#pragma unroll
	for (int s = 0; s < 7; s++)
#pragma unroll
		for (int j = 0; j < NT; j++) {
			const int k = 4 * s + kq;
			wf[s][j] = k < 27 ? g.w[(long)(NT * n + j) * 27 + k] : 0.f;
		}
	float bv[NT];
#pragma unroll
	for (int j = 0; j < NT; j++) bv[j] = g.bias ? g.bias[NT * n + j] : 0.f;
	const int waves = (gridDim.x * blockDim.x) >> 6;
	// MFMA row m of the group is pixel 4 (m & 3) + (m >> 2): D rows 4 kq + r are then pixels 4 r + kq, so ONE store instruction
	// (fixed r, kq = 0..3, 16 lanes x 16 bytes each) writes four CONSECUTIVE pixels = 1 KB contiguous
	const int pix = 4 * (n & 3) + (n >> 2);
	// the lane's seven patch elements k = 4 s + kq: row dy = k / 9, column dx = (k % 9) / 3, offset dy * row pitch + k % 9 from the
	// patch's first float -- fixed per lane, computed once
	int koff[7], kdy[7], kdx[7];
#pragma unroll
	for (int s = 0; s < 7; s++) {
		const int k = 4 * s + kq, dy = k / 9, r9 = k - dy * 9;
		kdy[s] = k < 27 ? dy : (1 << 28); // element 27 is padding: fails the row test
		kdx[s] = r9 / 3;
		koff[s] = dy * (int)g.a_sh + r9;
	}
	// The loop is bound by instruction ISSUE (knock-out on the MI355X: 0.39 of 0.94 ms with loads, MFMAs and stores all removed),
	// so what is not per-lane work is kept off the vector unit and small: the group index is wave-uniform (SALU), a group's
	// (image, row, position) advances incrementally by the wave stride instead of being divided out, the patch loads are
	// unconditional (address 0 of the image when the element is padding; the value is masked) -- no branch per load.
	int grp = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
	struct Pos { int img, oy, gx; };
	Pos cur;
	{
		const int row = g.d_gpr.div(grp);
		cur.gx = grp - row * g.groups_per_row; cur.img = g.d_oh.div(row); cur.oy = row - cur.img * g.OH;
	}
	// stride of one loop trip, decomposed the same way (wave-uniform constants)
	const int st_row = g.d_gpr.div(waves), st_gx = waves - st_row * g.groups_per_row;
	const int st_img = g.d_oh.div(st_row), st_oy = st_row - st_img * g.OH;
	auto advance = [&](Pos p) {
		p.gx += st_gx;
		const int c1 = p.gx >= g.groups_per_row ? 1 : 0;
		p.gx -= c1 * g.groups_per_row;
		p.oy += st_oy + c1;
		const int c2 = p.oy >= g.OH ? 1 : 0; // st_oy + c1 <= OH: one wrap at most
		p.oy -= c2 * g.OH;
		p.img += st_img + c2;
		return p;
	};
	// raw loads + a bit mask of the elements that exist; the mask is applied where the values are first USED (a select right
	// behind the load would make hipcc wait for the load on the spot)
	auto fetch = [&](const Pos& p, float (&v)[7], unsigned& okmask) {
		const float* const ap = g.a + (long)p.img * g.a_sn;
		const int iy0 = p.oy * g.sy - g.pad_y, ix0 = (p.gx * 16 + pix) * g.sx - g.pad_x;
		const int base = iy0 * (int)g.a_sh + ix0 * 3; // (one image spans < 2^31 floats: host-checked)
		okmask = 0;
#pragma unroll
		for (int s = 0; s < 7; s++) {
			const bool ok = ((unsigned)(iy0 + kdy[s]) < (unsigned)g.H) & ((unsigned)(ix0 + kdx[s]) < (unsigned)g.W);
			okmask |= ok ? (1u << s) : 0u;
			v[s] = ap[ok ? base + koff[s] : 0];
		}
	};
	// Software pipeline: the NEXT group's patch loads are issued before this group's stores.  The memory counter retires in order,
	// so a wave that loads after its stores waits for those stores' write acknowledgements before it can use the loads.
	float av[7];
	unsigned avm = 0;
	if (grp < g.groups) fetch(cur, av, avm);
#pragma unroll
	for (int s = 0; s < 7; s++) av[s] = (avm >> s) & 1 ? av[s] : 0.f;
	// Everything loaded so far (the filter fragments above all) has landed before the loop starts: otherwise hipcc's wait-count
	// pass carries "the filter loads may still be in flight" around the back edge and puts an s_waitcnt vmcnt(0) into EVERY
	// iteration -- which waits for the stores and the prefetch as well, i.e. undoes the pipeline.
	NNC_WAIT_VM0_ONLY();
	for (; grp < g.groups; grp += waves) {
		const Pos nxt = advance(cur);
		float nx[7];
		unsigned nxm = 0;
		const bool more = grp + waves < g.groups;
		if (more) fetch(nxt, nx, nxm);
		floatx4 acc[NT];
#pragma unroll
		for (int j = 0; j < NT; j++) acc[j] = floatx4{ 0.f, 0.f, 0.f, 0.f };
#pragma unroll
		for (int s = 0; s < 7; s++)
#pragma unroll
			for (int j = 0; j < NT; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], wf[s][j], acc[j], 0, 0, 0);
		// D: column n (channels NT n .. NT n + NT - 1 across the tiles), rows 4 kq + r = pixels 4 r + kq.  Stores go through a buffer
		// descriptor over the image: a pixel past the row's end gets an out-of-range offset and the hardware drops the store --
		// no branch, so exactly four stores sit between the prefetch above and its first use and hipcc can wait with vmcnt(4)
		// (behind branches it cannot count them and waits for everything, stores included).
		typedef unsigned int u4 __attribute__((ext_vector_type(4)));
		typedef unsigned int u2 __attribute__((ext_vector_type(2)));
		const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(g.b + (long)cur.img * g.b_sn), 0, g.b_image_bytes, 0x00020000);
		const int ox0 = cur.gx * 16;
		const unsigned row_off = (unsigned)(cur.oy * (int)g.b_sh + NT * n) * 4u;
		auto outv = [&](const int j, const int r) { const float o = acc[j][r] + bv[j]; return __float_as_uint(g.relu ? fmaxf(o, 0.f) : o); };
#pragma unroll
		for (int r = 0; r < 4; r++) {
			const int ox = ox0 + 4 * r + kq;
			const unsigned voff = ox < g.OW ? row_off + (unsigned)(ox * (int)g.b_sw) * 4u : 0x7ffff000u;
			if (NT == 4) __builtin_amdgcn_raw_buffer_store_b128(u4{ outv(0, r), outv(NT > 1 ? 1 : 0, r), outv(NT > 2 ? 2 : 0, r), outv(NT > 3 ? 3 : 0, r) }, rs, voff, 0, 0);
			else if (NT == 2) __builtin_amdgcn_raw_buffer_store_b64(u2{ outv(0, r), outv(NT > 1 ? 1 : 0, r) }, rs, voff, 0, 0);
			else __builtin_amdgcn_raw_buffer_store_b32(outv(0, r), rs, voff, 0, 0);
		}
		if (more) {
#pragma unroll
			for (int s = 0; s < 7; s++) av[s] = (nxm >> s) & 1 ? nx[s] : 0.f;
		}
		cur = nxt;
	}
}

// Filter gradient: dw[k][27] = sum over pixels g[pixel][k] * patch[pixel][27], and -- column 27 of the B operand being the
// constant 1 -- the bias gradient sum over pixels g[pixel][k] in the same contraction.  A = g (k x pixel): lane (m, kq) loads
// g[pixel 4 step + kq][MT m .. MT m + MT - 1] (16 bytes for MT = 4: output channel MT m + i in row m of channel tile i);
// B = patch (pixel x 32): two tiles.  Every wave accumulates MT x 2 tiles over its share of the pixels and writes them to
// part[workgroup][K][32] (its four waves summed through LDS); convc3_wgrad_fold adds the workgroups up in a fixed order (deterministic) and applies accumulate.
template <int MT>
static __global__ void __launch_bounds__(256) conv3x3_c3_wgrad_kernel(const ConvC3Args g, float* const part)
{
	const int lane = threadIdx.x & 63, n = lane & 15, kq = lane >> 4;
	floatx4 acc[MT][2];
#pragma unroll
	for (int i = 0; i < MT; i++) { acc[i][0] = floatx4{ 0.f, 0.f, 0.f, 0.f }; acc[i][1] = floatx4{ 0.f, 0.f, 0.f, 0.f }; }
	const int waves = (gridDim.x * blockDim.x) >> 6, wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	for (int grp = wid; grp < g.groups; grp += waves) {
		const int row = g.d_gpr.div(grp), gx = grp - row * g.groups_per_row;
		const int img = g.d_oh.div(row), oy = row - img * g.OH;
		const int ox0 = gx * 16;
		const float* const ap = g.a + (long)img * g.a_sn;
		const float* const gp = g.b + (long)img * g.b_sn + (long)oy * g.b_sh;
#pragma unroll
		for (int s = 0; s < 4; s++) { // four k-steps of four pixels
			const int ox = ox0 + 4 * s + kq;
			const bool live = ox < g.OW;
			float ga[MT];
			if (MT == 4) {
				const float4 v = live ? *(const float4*)(gp + (long)ox * g.b_sw + 4 * n) : make_float4(0.f, 0.f, 0.f, 0.f);
				ga[0] = v.x; ga[MT > 1 ? 1 : 0] = v.y; ga[MT > 2 ? 2 : 0] = v.z; ga[MT > 3 ? 3 : 0] = v.w;
			} else {
#pragma unroll
				for (int i = 0; i < MT; i++) ga[i] = live ? gp[(long)ox * g.b_sw + MT * n + i] : 0.f;
			}
			// B: lane (column n of tile t, pixel kq): patch element 16 t + n of pixel ox; element 27 = 1 (bias gradient), 28..31 = 0
			const float p0 = convc3_patch(ap, g, oy, ox, n);
			const float p1 = (16 + n) == 27 ? 1.f : convc3_patch(ap, g, oy, ox, 16 + n);
#pragma unroll
			for (int i = 0; i < MT; i++) {
				acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[i], live ? p0 : 0.f, acc[i][0], 0, 0, 0);
				acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[i], live ? p1 : 0.f, acc[i][1], 0, 0, 0);
			}
		}
	}
	// D of tile (i, t): row 4 kq + r = output channel MT (4 kq + r) + i, column n = patch element 16 t + n.  The workgroup's four waves meet in LDS first
	// ((w0 + w1) + (w2 + w3), a fixed order): one [K][32] partial per WORKGROUP -- a quarter of what the fold pass has to read (it was 148 us of the DawnNet's
	// 5.6 ms step for a 45 us kernel: 4096 per-wave partials of 8 KB through 64 workgroups)
	__shared__ float red[4][MT * 16 * 32];
	float* const mine = red[threadIdx.x >> 6];
#pragma unroll
	for (int i = 0; i < MT; i++)
#pragma unroll
		for (int t = 0; t < 2; t++)
#pragma unroll
			for (int r = 0; r < 4; r++) mine[(MT * (4 * kq + r) + i) * 32 + 16 * t + n] = acc[i][t][r];
	__syncthreads();
	float* const out = part + (long)blockIdx.x * g.K * 32;
	for (int i = threadIdx.x; i < MT * 16 * 32; i += 256) out[i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}

// dw[k][0..26] (+)= sum_w part[w][k][0..26], dbias[k] (+)= sum_w part[w][k][27].  One workgroup per output channel k: thread
// (slice = t >> 5, column = t & 31) adds waves slice, slice + 8, ... and the eight slices are folded through LDS in a fixed
// order (deterministic).  (One thread per (k, column) over all waves -- the first version -- read 33 MB through 8 workgroups: 1.2 ms.)
static __global__ void __launch_bounds__(256) convc3_wgrad_fold(const float* const part, const int waves, const int K, float* const dw, float* const dbias, const int accumulate)
{
	__shared__ float red[8][32];
	const int k = blockIdx.x, col = threadIdx.x & 31, slice = threadIdx.x >> 5;
	float s = 0.f;
	for (int w = slice; w < waves; w += 8) s += part[((long)w * K + k) * 32 + col];
	red[slice][col] = s;
	__syncthreads();
	if (slice != 0 || col > 27) return;
#pragma unroll
	for (int i = 1; i < 8; i++) s += red[i][col];
	if (col < 27) { float* const o = dw + (long)k * 27 + col; *o = accumulate ? *o + s : s; }
	else if (dbias) dbias[k] = accumulate ? dbias[k] + s : s;
}

} // namespace nnc
