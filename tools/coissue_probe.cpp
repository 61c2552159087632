// Micro-probe (not part of the library): how much other work hides behind v_mfma_f32_16x16x4_f32 on gfx950?
// One workgroup per CU, W waves per SIMD; every wave loops over 16 independent accumulators; behind each MFMA it issues F
// independent v_fma_f32 (and optionally one ds_read_b64).  Reports cycles per MFMA per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/coissue_probe.cpp -o tools/bin/coissue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int F, int LDS, int MF>
__global__ void __launch_bounds__(512) probe(float* out, int iters, float seed)
{
	__shared__ float lds[4096];
	const int t = threadIdx.x;
	floatx4 acc[16];
	for (int i = 0; i < 16; i++) acc[i] = floatx4{0, 0, 0, 0};
	float v[8];
	for (int i = 0; i < 8; i++) v[i] = seed + i + t;
	float a = seed * t, b = seed + 1;
	float2 l = make_float2(0, 0);
	lds[t * 2] = seed; lds[t * 2 + 1] = seed;
	__syncthreads();
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int m = 0; m < 16; m++) {
			if (MF) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[m]) : "v"(a), "v"(b));
#pragma unroll
			for (int f = 0; f < F; f++) { v[(m * F + f) & 7] = v[(m * F + f) & 7] * 1.0001f + 0.5f; asm volatile("" : "+v"(v[(m * F + f) & 7])); }
			if (LDS) { const float2 q = *(const float2*)(lds + ((t * 2 + m * 64) & 4094)); l.x += q.x; asm volatile("" : "+v"(l.x)); }
			__builtin_amdgcn_sched_barrier(0);
		}
		if (LDS) { v[0] += l.x; }
	}
	asm volatile("s_nop 15\n\ts_nop 15");
	float s = 0;
	for (int i = 0; i < 16; i++) s += acc[i][0] + acc[i][3];
	for (int i = 0; i < 8; i++) s += v[i];
	out[blockIdx.x * blockDim.x + t] = s;
}

template <int F, int LDS, int MF>
static void run(int waves_per_simd, float* out)
{
	const int iters = 2000, threads = 256 * waves_per_simd;
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL(HIP_KERNEL_NAME(probe<F, LDS, MF>), dim3(256), dim3(threads), 0, 0, out, 10, 1.f);
	hipEventRecord(e0, 0);
	hipLaunchKernelGGL(HIP_KERNEL_NAME(probe<F, LDS, MF>), dim3(256), dim3(threads), 0, 0, out, iters, 1.f);
	hipEventRecord(e1, 0);
	hipEventSynchronize(e1);
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	const double per_slot_ns = ms * 1e6 / ((double)iters * 16 * waves_per_simd);
	printf("waves/SIMD %d  MFMA %d  v_fma per slot %d  ds_read %d : %7.1f ns per slot per SIMD = %6.1f cycles @2.4GHz  (%s)\n", waves_per_simd, MF, F, LDS, per_slot_ns, per_slot_ns * 2.4,
		MF ? "32 = MFMA-bound" : "no MFMA");
}

int main()
{
	float* out;
	hipMalloc(&out, sizeof(float) * 256 * 512);
	for (int w = 1; w <= 2; w++) {
		if (w == 1) {
			run<0, 0, 1>(1, out); run<1, 0, 1>(1, out); run<2, 0, 1>(1, out); run<3, 0, 1>(1, out); run<4, 0, 1>(1, out); run<6, 0, 1>(1, out);
			run<2, 1, 1>(1, out); run<0, 1, 1>(1, out);
			run<1, 0, 0>(1, out); run<2, 0, 0>(1, out); run<4, 0, 0>(1, out); run<6, 0, 0>(1, out); run<2, 1, 0>(1, out);
		} else {
			run<0, 0, 1>(2, out); run<2, 0, 1>(2, out); run<4, 0, 1>(2, out); run<6, 0, 1>(2, out); run<4, 1, 1>(2, out);
			run<4, 0, 0>(2, out);
		}
	}
	return 0;
}
