// Perf probe (not part of the library): times the two-waves-per-SIMD fused Winograd kernel (ccv_amd/csrc/wino_fused2.h) next to the one-wave kernel
// (wino_fused.h) on one layer shape, with parts of its loop knocked out (DBG template bits).  Built HERE (hipcc cross-compiles) into tools/bin/:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ccv_amd/csrc -I tools tools/wf2_probe.cpp -o tools/bin/wf2_probe
//   tools/bin/wf2_probe [batch] [hw] [C] [K]
#include "wino_fused2.h"
#include <cstdio>
#include <vector>
#define CHECK(e) do { hipError_t s_ = (e); if (s_ != hipSuccess) { printf("HIP error %d at %d\n", (int)s_, __LINE__); return 1; } } while (0)
using namespace nnc;

template <int DBG>
static void run2(const WinoFusedArgs& a, unsigned grid, double flops, const char* what)
{
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	const int reps = 5;
	for (int i = 0; i < 2; i++) hipLaunchKernelGGL(HIP_KERNEL_NAME(wino_fused2_kernel<4, 4, DBG, false>), dim3(grid), dim3(512), 0, 0, a);
	hipEventRecord(e0, 0);
	for (int i = 0; i < reps; i++) hipLaunchKernelGGL(HIP_KERNEL_NAME(wino_fused2_kernel<4, 4, DBG, false>), dim3(grid), dim3(512), 0, 0, a);
	hipEventRecord(e1, 0);
	hipEventSynchronize(e1);
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	printf("8 waves DBG=%4d  %8.3f ms  %6.1f MFMA-TFLOP/s-equivalent = %.2f of the peak  %s%s\n", DBG, ms / reps, flops * reps / (ms * 1e-3) / 1e12, flops * reps / (ms * 1e-3) / 157.3e12, what, hipGetLastError() == hipSuccess ? "" : "  (launch error)");
}
template <int DBG, int SCHED = 0>
static void run(const WinoFusedArgs& a, unsigned grid, double flops, const char* what)
{
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	const int reps = 5;
	for (int i = 0; i < 2; i++) hipLaunchKernelGGL(HIP_KERNEL_NAME(wino_fused_kernel<4, 4, DBG, false, SCHED>), dim3(grid), dim3(256), 0, 0, a);
	hipEventRecord(e0, 0);
	for (int i = 0; i < reps; i++) hipLaunchKernelGGL(HIP_KERNEL_NAME(wino_fused_kernel<4, 4, DBG, false, SCHED>), dim3(grid), dim3(256), 0, 0, a);
	hipEventRecord(e1, 0);
	hipEventSynchronize(e1);
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	printf("SCHED=%d DBG=%4d  %8.3f ms  %6.1f MFMA-TFLOP/s-equivalent  %s%s\n", SCHED, DBG, ms / reps, flops * reps / (ms * 1e-3) / 1e12, what, hipGetLastError() == hipSuccess ? "" : "  (launch error)");
}

int main(int argc, char** argv)
{
	const int NB = argc > 1 ? atoi(argv[1]) : 256, H = argc > 2 ? atoi(argv[2]) : 223, C = argc > 3 ? atoi(argv[3]) : 64, K = argc > 4 ? atoi(argv[4]) : 64;
	const int W = H;
	float *src, *dst, *w, *uf, *bias;
	const size_t ns = (size_t)NB * H * W * C, nd = (size_t)NB * H * W * K;
	CHECK(hipMalloc(&src, sizeof(float) * ns));
	CHECK(hipMalloc(&dst, sizeof(float) * nd));
	CHECK(hipMalloc(&w, sizeof(float) * (size_t)K * 9 * C));
	CHECK(hipMalloc(&bias, sizeof(float) * K));
	const int KB = (K + WF_KT - 1) / WF_KT, CCn = C / WF_CC;
	CHECK(hipMalloc(&uf, sizeof(float) * (size_t)KB * CCn * WF_U_FLOATS));
	{
		std::vector<float> h((size_t)1 << 24);
		for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
		for (size_t o = 0; o < ns; o += h.size()) CHECK(hipMemcpy(src + o, h.data(), sizeof(float) * (ns - o < h.size() ? ns - o : h.size()), hipMemcpyHostToDevice));
		CHECK(hipMemcpy(w, h.data(), sizeof(float) * (size_t)K * 9 * C, hipMemcpyHostToDevice));
		CHECK(hipMemcpy(bias, h.data(), sizeof(float) * K, hipMemcpyHostToDevice));
	}
	hipLaunchKernelGGL(HIP_KERNEL_NAME(wino_weight_frag_kernel<false>), dim3((unsigned)(((size_t)KB * WF_KT * C + 255) / 256)), dim3(256), 0, 0, (const float*)w, uf, K, C, K, C);
	WinoFusedArgs a = {};
	a.src = src; a.dst = dst; a.uf = uf; a.bias = bias;
	a.s_sn = (long)H * W * C; a.s_sh = (long)W * C; a.s_sw = C; a.d_sn = (long)H * W * K; a.d_sh = (long)W * K; a.d_sw = K;
	a.H = H; a.W = W; a.OH = H; a.OW = W; a.pad_y = 1; a.pad_x = 1;
	const int TH = (H + 3) / 4, TW = (W + 3) / 4;
	a.GYn = (TH + 3) / 4; a.GXn = (TW + 3) / 4; a.groups = NB * a.GYn * a.GXn; a.C = C; a.K = K; a.CCn = CCn; a.KB = KB;
	a.dst_image_bytes = (unsigned)(((long)(H - 1) * a.d_sh + (long)(W - 1) * a.d_sw + K) * 4);
	a.src_image_bytes = (unsigned)(((long)(H - 1) * a.s_sh + (long)(W - 1) * a.s_sw + C) * 4);
	a.uf_kb_bytes = (unsigned)((size_t)CCn * WF_U_FLOATS * 4);
	const int items = (a.groups + 3) / 4 * KB;
	const int team = argc > 5 ? atoi(argv[5]) : (KB % 4 == 0 ? 4 : (KB % 2 == 0 ? 2 : 1));
	a.team = team;
	const unsigned grid = 256;
	const double flops = 2.0 * 36.0 * (double)a.groups * 16 * K * C; // issued MFMA work (padded tile groups included)
	printf("fused Winograd 3x3: N=%d %dx%dx%d -> %d; %d work items of %d trips on %d persistent workgroups in teams of %d; MFMA floor %.3f ms\n", NB, H, W, C, K, items, CCn, grid, team, flops / 157.3e12 * 1e3);
	run<0>(a, grid, flops, "one wave per SIMD: everything");
	run<64>(a, grid, flops, "one wave per SIMD: no epilogue");
	run2<0>(a, grid, flops, "everything");
	run2<64>(a, grid, flops, "no epilogue");
	run2<1>(a, grid, flops, "no DMA in the loop");
	run2<1 + 64>(a, grid, flops, "no DMA, no epilogue");
	run2<8>(a, grid, flops, "no transform VALU");
	run2<2>(a, grid, flops, "no patch reads");
	run2<4>(a, grid, flops, "no U reads");
	run2<32>(a, grid, flops, "no wait + barrier in the loop (racy: timing only)");
	run2<16>(a, grid, flops, "no MFMAs");
	run2<16 + 64>(a, grid, flops, "no MFMAs, no epilogue");
	run2<1 + 2 + 4 + 8 + 32 + 64>(a, grid, flops, "MFMAs only");
	return 0;
}
