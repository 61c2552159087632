// Perf probe (not part of the library): do fp32 VALU / LDS reads / LDS-DMA of ONE wave overlap with the fp32 MFMAs of ANOTHER wave on the same SIMD?
// 8 waves per workgroup (two per SIMD, 160 KB of LDS: one workgroup per CU).  Waves 0-3 issue MFMAs only; waves 4-7 run one of: nothing, scalar fp32 VALU,
// packed fp32 VALU, integer VALU, ds_read_b128, LDS-DMA pieces.  If the MFMA waves' time does not move, the other pipe is free behind them.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ccv_amd/csrc tools/coissue2_probe.cpp -o tools/bin/coissue2_probe
#include "wino_fused.h"
#include <cstdio>
using namespace nnc;
typedef float f2v __attribute__((ext_vector_type(2)));
// OTHER: 7 MFMAs on both waves, 8 both waves MFMA + PER packed VALU behind each, 9 both MFMA + ds_read_b128; 0 idle, 1 v_fma_f32, 2 v_pk_fma_f32, 3 v_add_u32, 4 ds_read_b128, 5 LDS-DMA (linear KB pieces), 6 v_fma_f32 on the SAME wave as the MFMAs (4 waves only)
template <int OTHER, int PER_MFMA4>
static __global__ void __launch_bounds__(512) coissue2_kernel(const float* __restrict__ src, float* __restrict__ out, const int iters, long long* __restrict__ cycles)
{
	extern __shared__ float lds[];
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	floatx4 acc[16];
#pragma unroll
	for (int i = 0; i < 16; i++) acc[i] = floatx4{ 0.f, 0.f, 0.f, 0.f };
	float a = 1.f + lane, b = 2.f, c = 0.5f;
	f2v p = { 1.f, 2.f }, q = { 0.5f, 0.25f };
	int n = lane;
	floatx4 u = { 0.f, 0.f, 0.f, 0.f };
	const wf_rsrc_t rs = wf_make_rsrc(src + (size_t)blockIdx.x * 16384, 1u << 20);
	const unsigned lds0 = wf_lds_addr(lds) + (unsigned)wave * 8192u;
	const long long t0 = __builtin_readcyclecounter();
	float cs[8] = { 1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f };
	f2v ps[8];
	for (int i = 0; i < 8; i++) ps[i] = f2v{ 1.f + i, 2.f };
	if (wave < 4 || OTHER == 6 || (OTHER >= 7 && OTHER <= 9)) {
		if (wave < 4 || (OTHER >= 7 && OTHER <= 9))
		for (int i = 0; i < iters; i++) {
#pragma unroll
			for (int z = 0; z < 16; z++) {
				WF_MFMA(acc[z], a, b, false);
				if constexpr (OTHER == 10) {
#pragma unroll
					for (int r = 0; r < PER_MFMA4; r++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(cs[(z * PER_MFMA4 + r) & 7]) : "v"(a), "v"(b));
				}
				if constexpr (OTHER == 11) {
#pragma unroll
					for (int r = 0; r < PER_MFMA4; r++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(ps[(z * PER_MFMA4 + r) & 7]) : "v"(q), "v"(q));
				}
				if constexpr (OTHER == 8) {
#pragma unroll
					for (int r = 0; r < PER_MFMA4; r++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p) : "v"(q), "v"(q));
				}
				if constexpr (OTHER == 9) {
#pragma unroll
					for (int r = 0; r < PER_MFMA4; r++) { asm volatile("ds_read_b128 %0, %1" : "=v"(u) : "v"((unsigned)(lane * 16 + wave * 8192)) : "memory"); }
				}
				if constexpr (OTHER == 6) {
#pragma unroll
					for (int r = 0; r < PER_MFMA4; r++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
				}
			}
		}
	} else if (OTHER < 10) {
		for (int i = 0; i < iters; i++) {
#pragma unroll
			for (int z = 0; z < 16; z++) {
#pragma unroll
				for (int r = 0; r < PER_MFMA4; r++) {
					if constexpr (OTHER == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
					else if constexpr (OTHER == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p) : "v"(q), "v"(q));
					else if constexpr (OTHER == 3) asm volatile("v_add_u32 %0, %1, %0" : "+v"(n) : "v"(lane));
					else if constexpr (OTHER == 4) { asm volatile("ds_read_b128 %0, %1" : "=v"(u) : "v"((unsigned)(lane * 16 + wave * 8192)) : "memory"); }
					else if constexpr (OTHER == 5) wf_dma16(rs, lds, lds0 + (unsigned)((z * PER_MFMA4 + r) & 7) * 1024u, (unsigned)lane * 16u, (unsigned)(((i * 16 + z) * PER_MFMA4 + r) & 63) * 1024u);
				}
			}
			if constexpr (OTHER == 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
			if constexpr (OTHER == 5) WF_WAIT_VMCNT(0);
		}
	}
	const long long t1 = __builtin_readcyclecounter();
	if (blockIdx.x == 0 && lane == 0) cycles[wave] = t1 - t0;
	floatx4 s = acc[0];
#pragma unroll
	for (int i = 1; i < 16; i++) s += acc[i];
	for (int i = 0; i < 8; i++) { c += cs[i]; p += ps[i]; }
	if (s[0] + c + p.x + (float)n + u[0] == 12345.678f) out[threadIdx.x] = s[0];
}
template <int OTHER, int PER>
static void run(const float* src, float* out, long long* cyc, const char* what)
{
	const int iters = 2000, grid = 256;
	auto k = coissue2_kernel<OTHER, PER>;
	hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL(k, dim3(grid), dim3(512), 163840, 0, src, out, iters, cyc);
	hipEventRecord(e0, 0);
	hipLaunchKernelGGL(k, dim3(grid), dim3(512), 163840, 0, src, out, iters, cyc);
	hipEventRecord(e1, 0);
	hipEventSynchronize(e1);
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	long long h[8];
	hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
	printf("%-44s x%d per MFMA: %7.3f ms; MFMA wave %5.1f clocks per MFMA (32 = pipe-bound), other wave %5.1f clocks per MFMA slot\n", what, PER, ms, (double)h[0] / (iters * 16.0), (double)h[4] / (iters * 16.0));
}
int main()
{
	float *src, *out; long long* cyc;
	hipMalloc(&src, 64u << 20); hipMalloc(&out, 4096); hipMalloc(&cyc, 64);
	hipMemset(src, 0, 64u << 20);
	run<0, 1>(src, out, cyc, "MFMA waves alone");
	run<1, 2>(src, out, cyc, "+ v_fma_f32 on the sibling wave"); run<1, 4>(src, out, cyc, "+ v_fma_f32 on the sibling wave"); run<1, 7>(src, out, cyc, "+ v_fma_f32 on the sibling wave");
	run<2, 2>(src, out, cyc, "+ v_pk_fma_f32 on the sibling wave"); run<2, 4>(src, out, cyc, "+ v_pk_fma_f32 on the sibling wave");
	run<3, 4>(src, out, cyc, "+ v_add_u32 on the sibling wave"); run<3, 7>(src, out, cyc, "+ v_add_u32 on the sibling wave");
	run<4, 1>(src, out, cyc, "+ ds_read_b128 on the sibling wave"); run<4, 2>(src, out, cyc, "+ ds_read_b128 on the sibling wave");
	run<5, 1>(src, out, cyc, "+ LDS-DMA KB pieces on the sibling wave");
	run<6, 2>(src, out, cyc, "v_fma_f32 on the SAME wave"); run<6, 4>(src, out, cyc, "v_fma_f32 on the SAME wave"); run<6, 7>(src, out, cyc, "v_fma_f32 on the SAME wave");
	run<10, 1>(src, out, cyc, "one wave: MFMA + independent v_fma_f32"); run<10, 2>(src, out, cyc, "one wave: MFMA + independent v_fma_f32"); run<10, 4>(src, out, cyc, "one wave: MFMA + independent v_fma_f32");
	run<11, 1>(src, out, cyc, "one wave: MFMA + independent v_pk_fma_f32"); run<11, 2>(src, out, cyc, "one wave: MFMA + independent v_pk_fma_f32"); run<11, 4>(src, out, cyc, "one wave: MFMA + independent v_pk_fma_f32");
	run<7, 1>(src, out, cyc, "MFMAs on BOTH waves of the SIMD");
	run<8, 1>(src, out, cyc, "both waves: MFMA + v_pk_fma_f32"); run<8, 2>(src, out, cyc, "both waves: MFMA + v_pk_fma_f32"); run<8, 4>(src, out, cyc, "both waves: MFMA + v_pk_fma_f32");
	run<9, 1>(src, out, cyc, "both waves: MFMA + ds_read_b128 (not waited for)");
	return 0;
}
