// Perf probe (not part of the library): does a second wave per SIMD hide the fused Winograd kernel's LDS-DMA issue and transform work behind the MFMAs?
// A synthetic loop with the instruction mix of ccv_amd/csrc/wino_fused.h per 8-channel chunk and CU -- 576 MFMAs (16x16x4 fp32), 80 one-KB LDS-DMA
// pieces (44 patch-like: 16 bytes per lane at a 256-byte pitch; 36 U-like: linear), 36 ds_read_b128 + 72 ds_read_b64 and ~288 VALU per 144 MFMAs, one
// wait + barrier per chunk -- run as 4 waves (one per SIMD: 144 MFMAs, 20 pieces each) and as 8 waves (two per SIMD: 72 MFMAs, 10 pieces each; the
// frequency positions split between the pair, the patches read by both).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ccv_amd/csrc tools/occ2_probe.cpp -o tools/bin/occ2_probe ; tools/bin/occ2_probe
#include "wino_fused.h"
#include <cstdio>
using namespace nnc;

// WAVES: 4 or 8.  NZ = positions per wave (36 or 18).  PIECES per wave and chunk.  PREADS = patch ds_read_b64 per position.  VPK = packed VALU per position.
template <int WAVES, int NZ, int PIECES, int PREADS, int VPK, int DBG, int RUN = 1>
static __global__ void __launch_bounds__(WAVES * 64) occ2_kernel(const float* __restrict__ src, const float* __restrict__ uf, float* __restrict__ out, const int chunks, const unsigned src_bytes, const unsigned uf_bytes)
{
	extern __shared__ float lds[]; // 160 KB: [2][4 x 11 KB patches][36 KB U]
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const int simd = wave & 3; // the pair (wave, wave + 4) shares a SIMD's patches
	const wf_rsrc_t rs_src = wf_make_rsrc(src + (size_t)blockIdx.x * 65536, src_bytes), rs_u = wf_make_rsrc(uf, uf_bytes);
	const unsigned lds0 = wf_lds_addr(lds);
	// RUN = lanes fetching one pixel's contiguous bytes (16 * RUN bytes per pixel, pixels 256 bytes apart); RUN = 64: the piece is one linear KB
	const unsigned pvoff = (unsigned)(lane / RUN) * 256u + (unsigned)(lane % RUN) * 16u, uvoff = (unsigned)lane * 16u;
	floatx4 acc[16];
#pragma unroll
	for (int i = 0; i < 16; i++) acc[i] = floatx4{ 0.f, 0.f, 0.f, 0.f };
	typedef float f2v __attribute__((ext_vector_type(2)));
	f2v t0 = { 1.f, 2.f }, t1 = { 0.5f, 0.25f };
	for (int c = 0; c < chunks; c++) {
		const unsigned buf = (unsigned)(c & 1) * 81920u;
		const unsigned p_lds = lds0 + buf + (unsigned)simd * 11264u, u_lds = lds0 + buf + 45056u;
		const unsigned sp = (unsigned)c * 32u, su = (unsigned)(c & 7) * 36864u;
		wf_static_for<NZ>([&](auto zc) {
			constexpr int z = decltype(zc)::value;
			if constexpr ((DBG & 1) == 0) {
				// pieces spread evenly over the positions
				if constexpr ((z * PIECES) / NZ != ((z + 1) * PIECES) / NZ) {
					constexpr int q = (z * PIECES) / NZ;
					if (q & 1) wf_dma16(rs_u, lds, u_lds + (unsigned)((wave * PIECES + q) % 36) * 1024u, uvoff, su + (unsigned)((wave * PIECES + q) % 36) * 1024u);
					else wf_dma16(rs_src, lds, p_lds + (unsigned)(q % 11) * 1024u, pvoff + (unsigned)q * 16384u, sp);
				}
			}
			floatx4 u = { 1.f, 1.f, 1.f, 1.f };
			f2v p = { 1.f, 1.f };
			if constexpr ((DBG & 2) == 0) {
				u = *(const floatx4*)((const char*)lds + (buf ^ 81920u) + 45056u + z * 1024 + lane * 16);
#pragma unroll
				for (int r = 0; r < PREADS; r++) {
					const f2v v = *(const f2v*)((const char*)lds + (buf ^ 81920u) + simd * 11264u + ((z * PREADS + r) % 40) * 256 + (lane & 31) * 8);
					p = p + v;
				}
			}
			if constexpr ((DBG & 4) == 0) {
#pragma unroll
				for (int r = 0; r < VPK; r++) {
					asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(t0) : "v"(t1), "v"(p));
				}
			}
			if constexpr ((DBG & 8) == 0) {
				const float a0 = t0.x + p.x, a1 = t0.y + p.y;
				WF_MFMA(acc[(z * 4 + 0) & 15], a0, u.x, false);
				WF_MFMA(acc[(z * 4 + 1) & 15], a0, u.y, false);
				WF_MFMA(acc[(z * 4 + 2) & 15], a1, u.z, false);
				WF_MFMA(acc[(z * 4 + 3) & 15], a1, u.w, false);
			}
		});
		if constexpr ((DBG & 16) == 0) {
			WF_WAIT_VMCNT(0);
			__syncthreads();
		}
	}
	floatx4 s = acc[0];
#pragma unroll
	for (int i = 1; i < 16; i++) s += acc[i];
	if (s[0] + s[1] + s[2] + s[3] + t0.x == 12345.678f) out[threadIdx.x] = s[0];
}

template <int WAVES, int NZ, int PIECES, int PREADS, int VPK, int DBG, int RUN = 1>
static void run(const float* src, const float* uf, float* out, const char* what)
{
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	const int chunks = 256, grid = 1024, reps = 3;
	auto k = occ2_kernel<WAVES, NZ, PIECES, PREADS, VPK, DBG, RUN>;
	hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
	hipLaunchKernelGGL(k, dim3(grid), dim3(WAVES * 64), 163840, 0, src, uf, out, chunks, 64u << 20, 8u * 36864u);
	hipEventRecord(e0, 0);
	for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k, dim3(grid), dim3(WAVES * 64), 163840, 0, src, uf, out, chunks, 64u << 20, 8u * 36864u);
	hipEventRecord(e1, 0);
	hipEventSynchronize(e1);
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	const double flops = (double)grid * chunks * WAVES * NZ * 4 * 2048.0 * reps;
	printf("%d waves  run %2d  DBG=%2d  %8.3f ms  %6.1f TFLOP/s = %.2f of the fp32 MFMA peak  %s%s\n", WAVES, RUN, DBG, ms / reps, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 157.3e12, what,
		hipGetLastError() == hipSuccess ? "" : "  (launch error)");
}

int main()
{
	float *src, *uf, *out;
	hipMalloc(&src, (64u << 20) + 1024u * 65536u * 4u); hipMalloc(&uf, 8 * 36864); hipMalloc(&out, 4096);
	hipMemset(src, 0, (64u << 20) + 1024u * 65536u * 4u); hipMemset(uf, 0, 8 * 36864);
#define BOTH(DBG, WHAT) run<4, 36, 20, 2, 8, DBG>(src, uf, out, WHAT); run<8, 18, 10, 4, 8, DBG>(src, uf, out, WHAT)
	BOTH(0, "everything");
	BOTH(1, "no DMA");
	BOTH(2, "no LDS reads");
	BOTH(4, "no VALU");
	BOTH(1 + 2 + 4, "MFMAs + barrier only");
	BOTH(8, "no MFMAs");
	BOTH(2 + 4 + 8, "DMA + barrier only");
	BOTH(16, "no wait + barrier");
	// two waves per SIMD with the single-wave work split (each does half of everything, patches NOT re-read): the upper bound of the split
	run<8, 18, 10, 2, 4, 0>(src, uf, out, "8 waves, nothing duplicated");
	run<8, 18, 10, 2, 4, 1>(src, uf, out, "8 waves, nothing duplicated, no DMA");
	// the patch-like pieces with 16 x RUN contiguous bytes per pixel (U-like pieces unchanged): what a piece costs the texture path as a function of the lines it touches
#define PAT(RUN) run<4, 36, 20, 2, 8, 14, RUN>(src, uf, out, "DMA + barrier only"); run<8, 18, 10, 4, 8, 14, RUN>(src, uf, out, "DMA + barrier only"); run<8, 18, 10, 4, 8, 2, RUN>(src, uf, out, "no LDS reads")
	PAT(2); PAT(4); PAT(8); PAT(64);
	return 0;
}
