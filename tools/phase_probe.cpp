// Perf probe (not part of the library): TWO waves per SIMD that each ALTERNATE between a pure fp32-MFMA phase (NM back-to-back v_mfma_f32_16x16x4_f32 on
// independent accumulators, optionally with the ds_read_b64 fragment reads a real contraction issues) and a pure VALU phase (NV independent v_pk_fma_f32,
// optionally with the LDS reads / writes of a transform), the partner in antiphase.  Round 3's coissue2_probe showed that a wave interleaving MFMA + VALU
// instruction by instruction pays ~14 clocks for the first VALU behind an MFMA whether it has a sibling or not, and that a VALU-only sibling does not slow an
// MFMA-only wave down.  The question here: does PHASING the work (what the 8-wave attention kernels of the guide do) let a SIMD's matrix pipe run at 32 clocks
// per MFMA while the VALU of the other wave rides along -- and how long may the VALU phase be?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/phase_probe.cpp -o tools/bin/phase_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));

// WAVES: 8 = two per SIMD (waves w, w + 4 share a SIMD by the dispatch order), 4 = one per SIMD.
// SYNC: 0 none, 1 s_barrier between the phases (all waves), 2 = none but s_setprio 1 inside the MFMA phase, 3 = s_setprio 1 inside the VALU phase
// LDSOPS: 0 none; 1 = one ds_read_b64 per two MFMAs in the MFMA phase + 36 ds_read_b64 and 18 ds_write_b64 in the VALU phase
// PACKED: 1 v_pk_fma_f32, 0 v_fma_f32
template <int WAVES, int NM, int NV, int SYNC, int LDSOPS, int PACKED>
static __global__ void __launch_bounds__(WAVES * 64) phase_kernel(float* __restrict__ out, const int iters, long long* __restrict__ cycles)
{
	extern __shared__ float lds[];
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const int grp = WAVES == 8 ? (wave >> 2) : 0;
	floatx4 acc[36];
#pragma unroll
	for (int i = 0; i < 36; i++) acc[i] = floatx4{ 0.f, 0.f, 0.f, 0.f };
	float a = 1.f + lane, b = 2.f;
	f2v ps[12];
	float cs[12];
#pragma unroll
	for (int i = 0; i < 12; i++) { ps[i] = f2v{ 1.f + i, 2.f }; cs[i] = 1.f + i; }
	const f2v q = { 0.5f, 0.25f };
	float2 u[2] = { make_float2(1.f, 2.f), make_float2(3.f, 4.f) };
	float* const my = lds + wave * 4096 + lane * 2;
	for (int i = threadIdx.x; i < WAVES * 4096; i += WAVES * 64) lds[i] = 1.f;
	__syncthreads();
	auto mfma_phase = [&]() {
		if constexpr (SYNC == 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
		for (int z = 0; z < NM; z++) {
			if constexpr (LDSOPS == 1) {
				if ((z & 1) == 0) u[(z >> 1) & 1 ^ 1] = *(const float2*)(my + ((z >> 1) & 15) * 128);
				acc[z % 36] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, (z & 1) ? u[(z >> 1) & 1].y : u[(z >> 1) & 1].x, acc[z % 36], 0, 0, 0);
			} else {
				asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[z % 36]) : "v"(a), "v"(b));
			}
		}
		if constexpr (SYNC == 2) __builtin_amdgcn_s_setprio(0);
	};
	auto valu_phase = [&]() {
		if constexpr (SYNC == 3) __builtin_amdgcn_s_setprio(1);
		float2 d[6];
#pragma unroll
		for (int v = 0; v < NV; v++) {
			if constexpr (LDSOPS == 1) {
				// 36 reads spread over the phase, 6 at a time; 18 writes at the end
				if (v % (NV / 6) == 0 && v / (NV / 6) < 6) {
#pragma unroll
					for (int r = 0; r < 6; r++) d[r] = *(const float2*)(my + ((v / (NV / 6)) * 6 + r) * 128 + 2048);
#pragma unroll
					for (int r = 0; r < 6; r++) { ps[r].x += d[r].x; ps[r + 6].y += d[r].y; }
				}
			}
			if constexpr (PACKED) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(ps[v % 12]) : "v"(q), "v"(q));
			else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(cs[v % 12]) : "v"(a), "v"(b));
		}
		if constexpr (LDSOPS == 1) {
#pragma unroll
			for (int r = 0; r < 18; r++) *(float2*)(my + r * 128 + 2048) = make_float2(ps[r % 12].x, ps[r % 12].y);
		}
		if constexpr (SYNC == 3) __builtin_amdgcn_s_setprio(0);
	};
	const long long t0 = __builtin_readcyclecounter();
	if (grp == 0) {
		for (int i = 0; i < iters; i++) {
			mfma_phase();
			if constexpr (SYNC == 1) __builtin_amdgcn_s_barrier();
			if constexpr (NV > 0) valu_phase();
			if constexpr (SYNC == 1) __builtin_amdgcn_s_barrier();
		}
	} else {
		for (int i = 0; i < iters; i++) {
			if constexpr (NV > 0) valu_phase();
			if constexpr (SYNC == 1) __builtin_amdgcn_s_barrier();
			mfma_phase();
			if constexpr (SYNC == 1) __builtin_amdgcn_s_barrier();
		}
	}
	const long long t1 = __builtin_readcyclecounter();
	if (blockIdx.x == 0 && lane == 0) cycles[wave] = t1 - t0;
	floatx4 s = acc[0];
#pragma unroll
	for (int i = 1; i < 36; i++) s += acc[i];
	float c = 0.f;
#pragma unroll
	for (int i = 0; i < 12; i++) c += ps[i].x + ps[i].y + cs[i];
	if (s[0] + c + u[0].x + u[1].y == 12345.678f) out[threadIdx.x] = s[0];
}

template <int WAVES, int NM, int NV, int SYNC, int LDSOPS, int PACKED>
static void run(float* out, long long* cyc, const char* what)
{
	const int iters = 400, grid = 256;
	auto k = phase_kernel<WAVES, NM, NV, SYNC, LDSOPS, PACKED>;
	hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL(k, dim3(grid), dim3(WAVES * 64), 163840, 0, out, iters, cyc);
	hipEventRecord(e0, 0);
	hipLaunchKernelGGL(k, dim3(grid), dim3(WAVES * 64), 163840, 0, out, iters, cyc);
	hipEventRecord(e1, 0);
	hipEventSynchronize(e1);
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	long long h[8] = { 0 };
	hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
	const double per_simd = (double)(WAVES / 4) * iters * NM; // MFMAs one SIMD executes
	printf("%d waves/SIMD  NM=%3d NV=%3d %s sync=%d lds=%d : %7.3f ms  wave0 %6.1f clocks per SIMD-MFMA (32 = pipe-bound; one wave alone would need %5.1f + VALU)  wave%d %6.1f   %s\n",
		WAVES / 4, NM, NV, PACKED ? "pk" : "sc", SYNC, LDSOPS, ms, (double)h[0] / per_simd, 32.0, WAVES - 1, (double)h[WAVES - 1] / per_simd, what);
	hipEventDestroy(e0); hipEventDestroy(e1);
}

int main()
{
	float* out; long long* cyc;
	hipMalloc(&out, 8192); hipMalloc(&cyc, 64);
	run<8, 36, 0, 0, 0, 1>(out, cyc, "MFMAs only, both waves");
	run<4, 36, 0, 0, 0, 1>(out, cyc, "MFMAs only, one wave");
	run<4, 36, 75, 0, 0, 1>(out, cyc, "one wave: phases in sequence");
	run<4, 36, 150, 0, 0, 1>(out, cyc, "one wave: phases in sequence");
	run<4, 72, 150, 0, 0, 1>(out, cyc, "one wave: phases in sequence");
	run<8, 36, 36, 0, 0, 1>(out, cyc, "antiphase, free-running");
	run<8, 36, 75, 0, 0, 1>(out, cyc, "antiphase, free-running");
	run<8, 36, 110, 0, 0, 1>(out, cyc, "antiphase, free-running");
	run<8, 36, 150, 0, 0, 1>(out, cyc, "antiphase, free-running");
	run<8, 36, 180, 0, 0, 1>(out, cyc, "antiphase, free-running");
	run<8, 72, 150, 0, 0, 1>(out, cyc, "antiphase, free-running");
	run<8, 72, 250, 0, 0, 1>(out, cyc, "antiphase, free-running");
	run<8, 18, 75, 0, 0, 1>(out, cyc, "antiphase, free-running");
	run<8, 36, 75, 1, 0, 1>(out, cyc, "antiphase, s_barrier between phases");
	run<8, 36, 150, 1, 0, 1>(out, cyc, "antiphase, s_barrier between phases");
	run<8, 72, 150, 1, 0, 1>(out, cyc, "antiphase, s_barrier between phases");
	run<8, 36, 75, 2, 0, 1>(out, cyc, "free-running, setprio 1 in the MFMA phase");
	run<8, 36, 150, 2, 0, 1>(out, cyc, "free-running, setprio 1 in the MFMA phase");
	run<8, 36, 75, 3, 0, 1>(out, cyc, "free-running, setprio 1 in the VALU phase");
	run<8, 36, 150, 3, 0, 1>(out, cyc, "free-running, setprio 1 in the VALU phase");
	run<8, 36, 150, 0, 0, 0>(out, cyc, "free-running, scalar v_fma_f32");
	run<8, 36, 300, 0, 0, 0>(out, cyc, "free-running, scalar v_fma_f32");
	run<8, 36, 72, 0, 1, 1>(out, cyc, "free-running + LDS traffic of a real loop");
	run<8, 36, 150, 0, 1, 1>(out, cyc, "free-running + LDS traffic of a real loop");
	run<8, 72, 150, 0, 1, 1>(out, cyc, "free-running + LDS traffic of a real loop");
	run<8, 36, 150, 1, 1, 1>(out, cyc, "s_barrier + LDS traffic of a real loop");
	run<4, 36, 150, 0, 1, 1>(out, cyc, "one wave + LDS traffic of a real loop");
	return 0;
}
