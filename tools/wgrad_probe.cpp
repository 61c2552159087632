// Perf probe (not part of the library): the fused Winograd filter-gradient kernel (ccv_amd/csrc/wino_wgrad_fused.h) with parts knocked out (DBG bits).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ccv_amd/csrc tools/wgrad_probe.cpp -o tools/bin/wgrad_probe ;  tools/bin/wgrad_probe [batch] [hw] [C] [K]
#include "wino_wgrad_fused.h"
#include <cstdio>
#include <vector>
using namespace nnc;
template <int DBG>
static void run(const WinoWgradFusedArgs& a, unsigned grid, double flops, const char* what)
{
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	const int reps = 3;
	hipLaunchKernelGGL(HIP_KERNEL_NAME(wino_wgrad_fused_kernel<DBG>), dim3(grid), dim3(256), 0, 0, a);
	hipEventRecord(e0, 0);
	for (int i = 0; i < reps; i++) hipLaunchKernelGGL(HIP_KERNEL_NAME(wino_wgrad_fused_kernel<DBG>), dim3(grid), dim3(256), 0, 0, a);
	hipEventRecord(e1, 0);
	hipEventSynchronize(e1);
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	printf("DBG=%3d  %8.3f ms  %6.1f MFMA-TFLOP/s-equivalent  %s%s\n", DBG, ms / reps, flops * reps / (ms * 1e-3) / 1e12, what, hipGetLastError() == hipSuccess ? "" : "  (launch error)");
}
int main(int argc, char** argv)
{
	const int NB = argc > 1 ? atoi(argv[1]) : 256, H = argc > 2 ? atoi(argv[2]) : 223, C = argc > 3 ? atoi(argv[3]) : 64, K = argc > 4 ? atoi(argv[4]) : 64;
	float *a, *g, *part, *bpart;
	const size_t na = (size_t)NB * H * H * C, ng = (size_t)NB * H * H * K;
	hipMalloc(&a, 4 * na); hipMalloc(&g, 4 * ng);
	hipMemset(a, 0, 4 * na); hipMemset(g, 0, 4 * ng);
	WinoWgradFusedArgs p = {};
	p.a = a; p.g = g;
	p.a_sn = (long)H * H * C; p.a_sh = (long)H * C; p.a_sw = C; p.g_sn = (long)H * H * K; p.g_sh = (long)H * K; p.g_sw = K;
	p.H = H; p.W = H; p.OH = H; p.OW = H; p.pad_y = 1; p.pad_x = 1;
	const int TH = (H + 3) / 4;
	p.GYn = (TH + WG_GH - 1) / WG_GH; p.GXn = (TH + WG_GW - 1) / WG_GW; p.groups = NB * p.GYn * p.GXn; p.C = C; p.K = K;
	p.kblocks = K / WG_KB; p.cblocks = C / WG_CB;
	const int nb = p.kblocks * p.cblocks;
	p.slices = (256 / nb) & ~7; if (p.slices < 8) p.slices = 8;
	p.per_slice = (p.groups + p.slices - 1) / p.slices;
	hipMalloc(&part, 4ul * 36 * p.slices * K * C); hipMalloc(&bpart, 4ul * 4 * p.slices * K);
	p.partial = part; p.bias_partial = bpart;
	const unsigned grid = (unsigned)(p.slices * nb);
	const double flops = 2.0 * 36.0 * (double)NB * TH * TH * K * C;
	printf("fused Winograd filter gradient: N=%d %dx%dx%d, K=%d; %d tile groups, %d slices x %d blocks, %d trips per workgroup; MFMA floor %.3f ms\n", NB, H, H, C, K, p.groups, p.slices, nb, p.per_slice, flops / 157.3e12 * 1e3);
	run<0>(p, grid, flops, "everything");
	run<1>(p, grid, flops, "no DMA in the loop");
	run<2>(p, grid, flops, "no LDS reads / transforms");
	run<16>(p, grid, flops, "no MFMAs");
	run<32>(p, grid, flops, "no wait + barrier");
	run<1 + 32>(p, grid, flops, "no DMA, no wait + barrier");
	run<1 + 2 + 32>(p, grid, flops, "MFMAs only");
	run<2 + 16>(p, grid, flops, "DMA + sync only");
	return 0;
}
