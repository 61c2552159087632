// Perf probe (not part of the library), round 5: the fused Winograd kernel (ccv_amd/csrc/wino_fused.h) with its EXPERIMENT bits (DBG 4096 ...) against
// the library's instantiation (DBG 0) on one layer shape -- time, and a checksum of the output tensor (an experiment must not change a value).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ccv_amd/csrc tools/wf5_probe.cpp -o tools/bin/wf5_probe
//   tools/bin/wf5_probe [batch] [hw] [C] [K] [team]
#include "wino_fused.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(e) do { hipError_t s_ = (e); if (s_ != hipSuccess) { printf("HIP error %d at %d\n", (int)s_, __LINE__); return 1; } } while (0)
using namespace nnc;

static float* g_dst = 0; static size_t g_nd = 0;
template <int GH, int GW, int DBG>
static void run(const WinoFusedArgs& a, unsigned grid, double flops, const char* what)
{
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	const int reps = 5;
	hipMemset(g_dst, 0xff, sizeof(float) * g_nd);
	for (int i = 0; i < 2; i++) hipLaunchKernelGGL(HIP_KERNEL_NAME(wino_fused_kernel<GH, GW, DBG, false, 0>), dim3(grid), dim3(256), 0, 0, a);
	hipEventRecord(e0, 0);
	for (int i = 0; i < reps; i++) hipLaunchKernelGGL(HIP_KERNEL_NAME(wino_fused_kernel<GH, GW, DBG, false, 0>), dim3(grid), dim3(256), 0, 0, a);
	hipEventRecord(e1, 0);
	hipEventSynchronize(e1);
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	// checksum over a strided sample of the output (exact: integer view)
	static std::vector<unsigned> h;
	const size_t sample = g_nd < (1u << 24) ? g_nd : (1u << 24);
	h.resize(sample);
	hipMemcpy(h.data(), g_dst + (g_nd - sample) / 2, sizeof(float) * sample, hipMemcpyDeviceToHost);
	unsigned long long x = 0;
	for (size_t i = 0; i < sample; i++) x = x * 1099511628211ull + h[i];
	printf("<%d,%d> DBG=%5d  %8.3f ms  %6.1f MFMA-TFLOP/s-equivalent  checksum %016llx  %s%s\n", GH, GW, DBG, ms / reps, flops * reps / (ms * 1e-3) / 1e12, x, what, hipGetLastError() == hipSuccess ? "" : "  (launch error)");
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
	const int ghh = argc > 6 ? atoi(argv[6]) : 4, gww = 16 / ghh;
	a.GYn = (TH + ghh - 1) / ghh; a.GXn = (TW + gww - 1) / gww; a.groups = NB * a.GYn * a.GXn; a.C = C; a.K = K; a.CCn = CCn; a.KB = KB;
	a.dst_image_bytes = (unsigned)(((long)(H - 1) * a.d_sh + (long)(W - 1) * a.d_sw + K) * 4);
	a.src_image_bytes = (unsigned)(((long)(H - 1) * a.s_sh + (long)(W - 1) * a.s_sw + C) * 4);
	a.uf_kb_bytes = (unsigned)((size_t)CCn * WF_U_FLOATS * 4);
	const int items = (a.groups + 3) / 4 * KB;
	const int team = argc > 5 ? atoi(argv[5]) : (KB % 8 == 0 ? 8 : (KB % 4 == 0 ? 4 : (KB % 2 == 0 ? 2 : 1)));
	a.team = team;
	const unsigned grid = 256;
	g_dst = dst; g_nd = nd;
	const double flops = 2.0 * 36.0 * (double)a.groups * 16 * K * C; // issued MFMA work (padded tile groups included)
	printf("fused Winograd 3x3: N=%d %dx%dx%d -> %d; %d work items of %d trips on %d persistent workgroups in teams of %d; MFMA floor %.3f ms\n", NB, H, W, C, K, items, CCn, grid, team, flops / 157.3e12 * 1e3);
	const int gh = argc > 6 ? atoi(argv[6]) : 4;
	if (gh == 4 && getenv("WF5_POLICY") && !getenv("WF5_NT")) { // the stores' cache policy against the patch lines' life in the XCD's L2 (run under rocprofv3 --pmc TCC_EA0_RDREQ_...)
		run<4, 4, 5 << 14>(a, grid, flops, "stores with the default cache policy (before round 5)");
		run<4, 4, 0>(a, grid, flops, "library: stores nt");
		run<4, 4, 2 << 14>(a, grid, flops, "stores sc1");
		run<4, 4, 3 << 14>(a, grid, flops, "stores sc1 nt");
		run<4, 4, 4 << 14>(a, grid, flops, "stores sc0 sc1");
	} else if (gh == 4 && getenv("WF5_NT")) {
		for (int i = 0; i < 3; i++) { run<4, 4, 5 << 14>(a, grid, flops, "stores with the default cache policy"); run<4, 4, 0>(a, grid, flops, "library: stores nt"); }
	} else if (getenv("WF5_POLICY") || getenv("WF5_NT")) {
		for (int i = 0; i < 3; i++) { run<2, 8, 5 << 14>(a, grid, flops, "stores with the default cache policy"); run<2, 8, 0>(a, grid, flops, "library: stores nt"); }
	} else if (gh == 4) {
		run<4, 4, 0>(a, grid, flops, "library");
		run<4, 4, 8192>(a, grid, flops, "epilogue stores with per-store vector address arithmetic (before round 5)");
		run<4, 4, 0>(a, grid, flops, "library (again)");
		run<4, 4, 8192>(a, grid, flops, "epilogue stores with per-store vector address arithmetic (again)");
	} else {
		run<2, 8, 0>(a, grid, flops, "library");
		run<2, 8, 8192>(a, grid, flops, "epilogue stores with per-store vector address arithmetic (before round 5)");
	}
	return 0;
}
