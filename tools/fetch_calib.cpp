// Counter calibration (not part of the library), round 5: what do FETCH_SIZE and the size-resolved request counters TCC_EA0_RDREQ{,_32B,_64B,_128B}_sum report for
// (A) a wide coalesced streaming read (16 bytes per lane: the case /opt/skills/guides/MI355X_MICROARCH.md calibrates -- FETCH_SIZE shows half the bytes) and
// (B) the fused Winograd kernel's patch pieces: LDS-DMA where lanes 2 i, 2 i + 1 fetch 32 contiguous bytes of pixel i, pixels 256 bytes apart (64 channels),
//     the eight 32-byte chunks of a pixel fetched in eight separate sweeps over the buffer (the kernel fetches them a trip apart)?
// Both read every byte of a 2 GiB buffer exactly once -- far beyond L2 and the Infinity Cache -- so the true HBM read volume is at least 2 GiB in (A), and in (B)
// 2 GiB x (fill granule / 32 bytes) if nothing of a line survives between sweeps.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ccv_amd/csrc tools/fetch_calib.cpp -o tools/bin/fetch_calib
//   rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d out -- tools/bin/fetch_calib
#include "wino_fused.h"
#include <cstdio>
using namespace nnc;

__global__ void __launch_bounds__(256) calib_wide_kernel(const float4* __restrict__ src, float* __restrict__ out, const size_t n16)
{
	float s = 0.f;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const float4 v = src[i]; s += v.x + v.y + v.z + v.w; }
	if (s == 12345.678f) out[threadIdx.x] = s;
}

// one workgroup = 256 lanes; piece = 64 lanes x 16 bytes: lane pair (2 i, 2 i + 1) reads bytes [chunk * 32, chunk * 32 + 32) of pixel p0 + i
__global__ void __launch_bounds__(256) calib_patch_kernel(const float* __restrict__ src, float* __restrict__ out, const size_t pixels, const int chunk)
{
	__shared__ __attribute__((aligned(16))) float lds[4 * 8 * 256]; // 8 pieces in flight per wave
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const unsigned lds0 = __builtin_amdgcn_readfirstlane(wf_lds_addr(lds)) + wave * 8192;
	const size_t per_wave = 32; // pixels per piece
	const size_t waves = (size_t)gridDim.x * 4, w = (size_t)blockIdx.x * 4 + wave;
	for (size_t p0 = w * per_wave * 8; p0 < pixels; p0 += waves * per_wave * 8) {
#pragma unroll
		for (int q = 0; q < 8; q++) {
			const size_t pix = p0 + q * per_wave + (lane >> 1);
			// a descriptor per 1 GiB window would be needed for 32-bit offsets: rebuild the base per piece instead (wave-uniform)
			const wf_rsrc_t rs = wf_make_rsrc(src + (p0 + q * per_wave) * 64, 32 * 256);
			const unsigned voff = pix < pixels ? (unsigned)((lane >> 1) * 256 + chunk * 32 + (lane & 1) * 16) : WF_OOB;
			NNC_ASM_NOPS("s_nop 4");
			wf_dma16(rs, lds, lds0 + q * 1024, voff, 0u);
		}
		WF_WAIT_VMCNT(0);
	}
	if (lds[threadIdx.x] == 12345.678f) out[threadIdx.x] = lds[threadIdx.x + 1];
}

// (B2) the same 32-byte pieces, but the eight chunks of a pixel (= its two 128-byte lines) fetched by the SAME wave: MODE 0 back to back (four requests for one
// line in flight together), MODE 1 with s_waitcnt vmcnt(0) between chunks (a line's later pieces issued after its first landed), MODE 2 two chunks back to
// back, then a wait (the fused kernel's pair schedule: 16 channels per pair of trips)
template <int MODE>
__global__ void __launch_bounds__(256) calib_patch2_kernel(const float* __restrict__ src, float* __restrict__ out, const size_t pixels)
{
	__shared__ __attribute__((aligned(16))) float lds[4 * 8 * 256];
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const unsigned lds0 = __builtin_amdgcn_readfirstlane(wf_lds_addr(lds)) + wave * 8192;
	const size_t waves = (size_t)gridDim.x * 4, w = (size_t)blockIdx.x * 4 + wave;
	for (size_t p0 = w * 32; p0 < pixels; p0 += waves * 32) {
		const wf_rsrc_t rs = wf_make_rsrc(src + p0 * 64, 32 * 256);
#pragma unroll
		for (int chunk = 0; chunk < 8; chunk++) {
			const unsigned voff = p0 + (lane >> 1) < pixels ? (unsigned)((lane >> 1) * 256 + chunk * 32 + (lane & 1) * 16) : WF_OOB;
			NNC_ASM_NOPS("s_nop 4");
			wf_dma16(rs, lds, lds0 + chunk * 1024, voff, 0u);
			if (MODE == 1 || (MODE == 2 && (chunk & 1))) WF_WAIT_VMCNT(0);
		}
		WF_WAIT_VMCNT(0);
	}
	if (lds[threadIdx.x] == 12345.678f) out[threadIdx.x] = lds[threadIdx.x + 1];
}

// (C) the shape of the library's 2 x 2 / 2 NHWC max pool at 64 channels: a lane owns (output pixel, 4 channels), reads its four input pixels 16 bytes each
// (16 lanes = one 256-byte pixel; a load instruction = pixels x0, x0 + 2, x0 + 4, x0 + 6), every input line read by exactly one instruction of one wave.
// STORE = 0: no output written;  ORDER = 1: a load instruction covers 1024 contiguous bytes instead (pixels 4 j .. 4 j + 3), same bytes per wave
template <int STORE, int ORDER>
__global__ void __launch_bounds__(256) calib_pool_kernel(const float* __restrict__ a, float* __restrict__ b, const int N, const int H, const int W)
{
	const int OH = H / 2, OW = W / 2;
	const size_t total = (size_t)N * OH * OW * 16;
	const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (idx >= total) return;
	const int c4 = idx & 15; size_t r = idx >> 4;
	const int ox = r % OW; r /= OW;
	const int oy = r % OH; const int n = r / OH;
	const float* ap = a + ((size_t)n * H + oy * 2) * W * 64 + c4 * 4;
	float4 v0, v1, v2, v3;
	if (ORDER == 0) {
		v0 = *(const float4*)(ap + (ox * 2) * 64); v1 = *(const float4*)(ap + (ox * 2 + 1) * 64);
		v2 = *(const float4*)(ap + (size_t)W * 64 + (ox * 2) * 64); v3 = *(const float4*)(ap + (size_t)W * 64 + (ox * 2 + 1) * 64);
	} else { // the wave's 4 output pixels are ox & ~3 ..: its 8 input pixels per row, read as two runs of 4
		const int xb = (ox & ~3) * 2, j = ox & 3;
		v0 = *(const float4*)(ap + (xb + j) * 64); v1 = *(const float4*)(ap + (xb + 4 + j) * 64);
		v2 = *(const float4*)(ap + (size_t)W * 64 + (xb + j) * 64); v3 = *(const float4*)(ap + (size_t)W * 64 + (xb + 4 + j) * 64);
	}
	float4 m;
	m.x = fmaxf(fmaxf(v0.x, v1.x), fmaxf(v2.x, v3.x)); m.y = fmaxf(fmaxf(v0.y, v1.y), fmaxf(v2.y, v3.y));
	m.z = fmaxf(fmaxf(v0.z, v1.z), fmaxf(v2.z, v3.z)); m.w = fmaxf(fmaxf(v0.w, v1.w), fmaxf(v2.w, v3.w));
	if (STORE) *(float4*)(b + ((size_t)(n * OH + oy) * OW + ox) * 64 + c4 * 4) = m;
	else if (m.x == 12345.678f) b[threadIdx.x] = m.y;
}

int main()
{
	const size_t bytes = (size_t)2 << 30;
	float *src, *out;
	if (hipMalloc(&src, bytes) != hipSuccess || hipMalloc(&out, 4096) != hipSuccess) { printf("allocation failed\n"); return 1; }
	hipMemset(src, 0, bytes);
	hipDeviceSynchronize();
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	float ms = 0;
	hipEventRecord(e0, 0);
	hipLaunchKernelGGL(calib_wide_kernel, dim3(256 * 32), dim3(256), 0, 0, (const float4*)src, out, bytes / 16);
	hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
	printf("calib_wide_kernel: %.1f MB read once, 16 bytes per lane, %.3f ms = %.2f TB/s\n", bytes / 1e6, ms, bytes / (ms * 1e-3) / 1e12);
	const size_t pixels = bytes / 256;
	hipEventRecord(e0, 0);
	for (int chunk = 0; chunk < 8; chunk++) hipLaunchKernelGGL(calib_patch_kernel, dim3(256 * 8), dim3(256), 0, 0, (const float*)src, out, pixels, chunk);
	hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
	printf("calib_patch_kernel x 8 sweeps: %.1f MB read once in total (32 bytes per lane pair per sweep, pixels 256 bytes apart), %.3f ms\n", bytes / 1e6, ms);
#define PATCH2_RUN(M) \
	hipEventRecord(e0, 0); hipLaunchKernelGGL(HIP_KERNEL_NAME(calib_patch2_kernel<M>), dim3(256 * 8), dim3(256), 0, 0, (const float*)src, out, pixels); \
	hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); \
	printf("calib_patch2_kernel<mode %d>: %.1f MB read once (32-byte pieces, a pixel's eight chunks by one wave), %.3f ms\n", M, bytes / 1e6, ms);
	PATCH2_RUN(0) PATCH2_RUN(1) PATCH2_RUN(2)
	{
		const int N = 160, H = 224, W = 224; // 160 x 224 x 224 x 64 x 4 bytes = 2055 MB in, 514 MB out
		float* dst;
		if (hipMalloc(&dst, bytes / 4) != hipSuccess) { printf("allocation failed\n"); return 1; }
		const size_t total = (size_t)N * (H / 2) * (W / 2) * 16;
		const dim3 grid((unsigned)((total + 255) / 256));
		const double in_mb = (double)N * H * W * 64 * 4 / 1e6;
#define POOL_RUN(S, O) \
		hipEventRecord(e0, 0); hipLaunchKernelGGL(HIP_KERNEL_NAME(calib_pool_kernel<S, O>), grid, dim3(256), 0, 0, src, dst, N, H, W); \
		hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); \
		printf("calib_pool_kernel<store %d, order %d>: %.1f MB read once, %.3f ms\n", S, O, in_mb, ms);
		POOL_RUN(1, 0) POOL_RUN(0, 0) POOL_RUN(1, 1) POOL_RUN(0, 1)
	}
	return 0;
}
