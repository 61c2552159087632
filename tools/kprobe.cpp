// Perf probe (not part of the library): times the conv-forward instantiation of the contraction kernel on one VGG-D layer
// shape with parts of its steady state knocked out (DBG template bits of mfma_gemm_f32_kernel), to attribute the time.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ccv_amd/csrc tools/kprobe.cpp -o gpurun_out/kprobe && gpurun_out/kprobe
#include "mfma_gemm.h"
#include <cstdio>
#include <vector>
#define CHECK(e) do { hipError_t s_ = (e); if (s_ != hipSuccess) { printf("HIP error %d at %d\n", (int)s_, __LINE__); return 1; } } while (0)
using namespace nnc;

template <int DBG, int WM = 2, int WN = 2>
static float run(Im2colKC<true, false> la, MatLoader<true, true> lb, EpiStore epi, int M, int N, int K, int reps, KOrder ko = KOrder())
{
	const int tiles_m = (M + 64 * WM - 1) / (64 * WM), tiles_n = (N + 64 * WN - 1) / (64 * WN);
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	for (int i = 0; i < 2; i++)
		hipLaunchKernelGGL(HIP_KERNEL_NAME(mfma_gemm_f32_kernel<Im2colKC<true, false>, MatLoader<true, true>, EpiStore, WM, WN, DBG>), dim3(tiles_m * tiles_n), dim3(256), 0, 0, la, lb, epi, tiles_m, tiles_n, K, K, 1, 0L, 0L, 0L, 0L, ko);
	hipEventRecord(e0, 0);
	for (int i = 0; i < reps; i++)
		hipLaunchKernelGGL(HIP_KERNEL_NAME(mfma_gemm_f32_kernel<Im2colKC<true, false>, MatLoader<true, true>, EpiStore, WM, WN, DBG>), dim3(tiles_m * tiles_n), dim3(256), 0, 0, la, lb, epi, tiles_m, tiles_n, K, K, 1, 0L, 0L, 0L, 0L, ko);
	hipEventRecord(e1, 0);
	hipEventSynchronize(e1);
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	const double tf = 2.0 * M * N * (double)K * reps / (ms * 1e-3) / 1e12;
	printf("tile %dx%d DBG=%2d  %8.3f ms/launch  %7.1f TFLOP/s-equivalent%s\n", 64 * WM, 64 * WN, DBG, ms / reps, tf, hipGetLastError() == hipSuccess ? "" : "  (launch error)");
	return ms;
}

int main(int argc, char** argv)
{
	const int NB = argc > 1 ? atoi(argv[1]) : 64, H = argc > 2 ? atoi(argv[2]) : 55, C = argc > 3 ? atoi(argv[3]) : 256, KO = argc > 4 ? atoi(argv[4]) : 256;
	const int W = H, M = NB * H * W, K = 9 * C;
	float *a, *w, *b, *zp;
	CHECK(hipMalloc(&a, sizeof(float) * (size_t)NB * H * W * C));
	CHECK(hipMalloc(&w, sizeof(float) * (size_t)KO * K));
	CHECK(hipMalloc(&b, sizeof(float) * (size_t)M * KO));
	CHECK(hipMalloc(&zp, 256));
	CHECK(hipMemset(zp, 0, 256));
	std::vector<float> h((size_t)NB * H * W * C);
	for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
	CHECK(hipMemcpy(a, h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice));
	CHECK(hipMemcpy(w, h.data(), sizeof(float) * (size_t)KO * K, hipMemcpyHostToDevice));
	Im2colKC<true, false> la;
	la.p = a; la.zoff = zp - a; la.s_n = (long)H * W * C; la.s_h = W * C; la.s_w = C; la.H = H; la.W = W; la.OW = W; la.OHW = H * W; la.M = M; la.C = C; la.KWC = 3 * C; la.K = K;
	la.my = 1; la.mx = 1; la.oy_off = -1; la.ox_off = -1; la.ty = 1; la.tx = 1; la.dv_y = 1; la.dv_x = 1;
	la.finish();
	MatLoader<true, true> lb;
	lb.p = w; lb.zoff = zp - w; lb.ldr = K; lb.ldk = 1; lb.R = KO; lb.K = K;
	EpiStore epi;
	epi.c = b; epi.ldm = KO; epi.ldn = 1; epi.bias = 0; epi.alpha = 1.f; epi.accumulate = 0; epi.M = M; epi.N = KO; epi.bias_ldm = 0;
	printf("conv fwd 3x3 N=%d %dx%dx%d -> %d : M=%d N=%d K=%d\n", NB, H, W, C, KO, M, KO, K);
	const int reps = 10;
	run<0>(la, lb, epi, M, KO, K, reps);
	run<1>(la, lb, epi, M, KO, K, reps);   // no global loads
	run<8>(la, lb, epi, M, KO, K, reps);   // no address prep (re-loads the same addresses)
	run<9>(la, lb, epi, M, KO, K, reps);   // no loads, no prep
	run<11>(la, lb, epi, M, KO, K, reps);  // + no LDS writes
	run<15>(la, lb, epi, M, KO, K, reps);  // + no barrier: MFMAs + fragment ds_reads only
	run<4>(la, lb, epi, M, KO, K, reps);   // everything but the barrier (racy; timing only)
	run<2>(la, lb, epi, M, KO, K, reps);   // no LDS writes only
	run<16>(la, lb, epi, M, KO, K, reps);  // no MFMAs: the memory/LDS/VALU side alone
	run<32>(la, lb, epi, M, KO, K, reps);  // no A (activation gather) loads
	run<64>(la, lb, epi, M, KO, K, reps);  // no B (weight) loads
	{
		KOrder ko; ko.init(9, C);
		printf("channel-chunk-major K order:\n");
		run<0>(la, lb, epi, M, KO, K, reps, ko);
	}
	la.s_n = 0;                            // every image aliases image 0: the activation gather becomes L2-resident
	printf("aliased images (A operand L2-resident):\n");
	run<0>(la, lb, epi, M, KO, K, reps);
	return 0;
}
