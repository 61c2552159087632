// Probe (not part of the library), round 6: what bounds the half-precision 1 x 1 convolutions that WRITE the wide tensor (ResNet-50's 64 -> 256 at 56^2, batch 256:
// 103 MB read, 411 MB written, 0.189 ms = 2.7 TB/s in the contraction kernel against 5.1 TB/s for the same bytes the other way round)?  The memory side of that
// kernel without its arithmetic: every workgroup reads the [C][BP] slice of one image's input planes and writes the [M rows][BP] tile of its output planes, 16 bytes
// per lane, in several launch shapes:
//   mode 0  one workgroup per (image, 128-row block, 128-pixel block), as the contraction kernel is launched (12 800 workgroups, 40 KB of LDS each: three per CU)
//   mode 1  one workgroup per (image, 128-pixel block), ALL rows (the input slice read once)
//   mode 2  persistent: 2 workgroups per CU walk the (image, pixel block) items, the next item's loads in flight while the current one is stored
//   mode 3  mode 2 with nontemporal stores
//   mode 4  mode 0 with 256-pixel blocks (512-byte row segments)
//   mode 5  plain streaming: linear read of the input, linear write of the output (the ceiling)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/store_probe.cpp -o tools/bin/store_probe && tools/bin/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Geo { int N, C, M, P; };

__device__ __forceinline__ u4 mix(u4 a, u4 b) { return u4{ a.x ^ b.y, a.y + b.x, a.z ^ b.w, a.w + b.z }; }

template <int BP, int ROWS, bool NT>
__device__ __forceinline__ void item(const Geo g, const unsigned short* __restrict__ in, unsigned short* __restrict__ out, const int n, const int m0, const int p0, const int t, float* lds)
{
	// read the [C][BP] slice: C * BP / 8 chunks over 256 threads
	constexpr int CPR = BP / 8; // chunks per row
	u4 acc = u4{ 0, 0, 0, 0 };
	const unsigned short* const src = in + ((long)n * g.C) * g.P + p0;
	for (int id = t; id < g.C * CPR; id += 256) {
		const int c = id / CPR, x = (id % CPR) * 8;
		if (p0 + x + 8 <= g.P) acc = mix(acc, *(const u4*)(src + (long)c * g.P + x));
	}
	lds[t] = (float)acc.x; // (keeps the LDS allocation alive)
	unsigned short* const dst = out + ((long)n * g.M + m0) * g.P + p0;
	for (int id = t; id < ROWS * CPR; id += 256) {
		const int r = id / CPR, x = (id % CPR) * 8;
		if (p0 + x + 8 <= g.P && m0 + r < g.M) {
			u4 v = acc; v.x += id;
			if (NT) __builtin_nontemporal_store(v, (u4*)(dst + (long)r * g.P + x));
			else *(u4*)(dst + (long)r * g.P + x) = v;
		}
	}
}

template <int BP, int ROWS>
__global__ void __launch_bounds__(256) tile_kernel(const Geo g, const unsigned short* __restrict__ in, unsigned short* __restrict__ out, const int mblocks, const int pblocks)
{
	__shared__ float lds[10240]; // 40 KB: three workgroups per CU, as the contraction kernel
	const int b = blockIdx.x;
	const int n = b / (mblocks * pblocks), r = b % (mblocks * pblocks);
	item<BP, ROWS, false>(g, in, out, n, (r / pblocks) * ROWS, (r % pblocks) * BP, threadIdx.x, lds);
}

template <int BP, bool NT>
__global__ void __launch_bounds__(256) persistent_kernel(const Geo g, const unsigned short* __restrict__ in, unsigned short* __restrict__ out, const int pblocks, const int items)
{
	__shared__ float lds[16384];
	constexpr int CPR = BP / 8;
	const int t = threadIdx.x;
	// this workgroup's contiguous range of items
	const int per = (items + gridDim.x - 1) / gridDim.x;
	int it = blockIdx.x * per, end = it + per < items ? it + per : items;
	// C <= 64 * 256 * 8 / BP ... keep the chunks of one item in registers: C * CPR / 256 chunks per thread (C = 64, BP = 128: 4)
	constexpr int MAXCH = 8;
	u4 cur[MAXCH], nxt[MAXCH];
	const int nch = g.C * CPR / 256;
	auto load = [&](u4 (&r)[MAXCH], const int item_) {
		const int n = item_ / pblocks, p0 = (item_ % pblocks) * BP;
		const unsigned short* const src = in + ((long)n * g.C) * g.P + p0;
#pragma unroll
		for (int j = 0; j < MAXCH; j++) if (j < nch) {
			const int id = t + 256 * j, c = id / CPR, x = (id % CPR) * 8;
			r[j] = p0 + x + 8 <= g.P ? *(const u4*)(src + (long)c * g.P + x) : u4{ 0, 0, 0, 0 };
		}
	};
	if (it < end) load(cur, it);
	for (; it < end; it++) {
		if (it + 1 < end) load(nxt, it + 1);
		u4 acc = u4{ 0, 0, 0, 0 };
#pragma unroll
		for (int j = 0; j < MAXCH; j++) if (j < nch) acc = mix(acc, cur[j]);
		lds[t] = (float)acc.x;
		const int n = it / pblocks, p0 = (it % pblocks) * BP;
		unsigned short* const dst = out + ((long)n * g.M) * g.P + p0;
		for (int id = t; id < g.M * CPR; id += 256) {
			const int r = id / CPR, x = (id % CPR) * 8;
			if (p0 + x + 8 <= g.P) {
				u4 v = acc; v.x += id;
				if (NT) __builtin_nontemporal_store(v, (u4*)(dst + (long)r * g.P + x));
				else *(u4*)(dst + (long)r * g.P + x) = v;
			}
		}
#pragma unroll
		for (int j = 0; j < MAXCH; j++) cur[j] = nxt[j];
	}
}

__global__ void __launch_bounds__(256) stream_kernel(const u4* __restrict__ in, u4* __restrict__ out, const size_t nin, const size_t nout)
{
	const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
	u4 acc = u4{ 0, 0, 0, 0 };
	for (size_t j = i; j < nin; j += stride) acc = mix(acc, in[j]);
	for (size_t j = i; j < nout; j += stride) { u4 v = acc; v.x += (unsigned)j; out[j] = v; }
}

int main()
{
	const Geo shapes[] = { { 256, 64, 256, 3136 }, { 256, 256, 64, 3136 }, { 256, 128, 512, 784 }, { 256, 256, 1024, 196 } };
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	for (const Geo& g : shapes) {
		const size_t nin = (size_t)g.N * g.C * g.P, nout = (size_t)g.N * g.M * g.P;
		unsigned short *in, *out;
		CHECK(hipMalloc(&in, nin * 2)); CHECK(hipMalloc(&out, nout * 2));
		CHECK(hipMemset(in, 1, nin * 2));
		const double gb = (nin + nout) * 2 / 1e9;
		printf("N %d  C %d -> M %d  P %d   %.0f MB read, %.0f MB written\n", g.N, g.C, g.M, g.P, nin * 2 / 1e6, nout * 2 / 1e6);
		auto timed = [&](const char* name, auto launch) {
			launch();
			CHECK(hipEventRecord(e0, 0));
			for (int i = 0; i < 5; i++) launch();
			CHECK(hipEventRecord(e1, 0));
			CHECK(hipEventSynchronize(e1));
			CHECK(hipGetLastError());
			float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
			printf("  %-58s %7.3f ms  %5.2f TB/s\n", name, ms, gb / ms); fflush(stdout);
		};
		const int pb128 = (g.P + 127) / 128, pb256 = (g.P + 255) / 256, mb = (g.M + 127) / 128;
		timed("0: workgroup per (image, 128 rows, 128 px), 3 per CU", [&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(tile_kernel<128, 128>), dim3(g.N * mb * pb128), dim3(256), 0, 0, g, in, out, mb, pb128); });
		if (g.M <= 1024) timed("1: workgroup per (image, 128 px), all rows", [&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(tile_kernel<128, 1024>), dim3(g.N * pb128), dim3(256), 0, 0, g, in, out, 1, pb128); });
		if (g.C * 16 / 256 <= 8) {
			timed("2: persistent, 512 workgroups, next item's loads in flight", [&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(persistent_kernel<128, false>), dim3(512), dim3(256), 0, 0, g, in, out, pb128, g.N * pb128); });
			timed("3: the same, nontemporal stores", [&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(persistent_kernel<128, true>), dim3(512), dim3(256), 0, 0, g, in, out, pb128, g.N * pb128); });
			timed("2b: persistent, 1024 workgroups", [&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(persistent_kernel<128, false>), dim3(1024), dim3(256), 0, 0, g, in, out, pb128, g.N * pb128); });
		}
		timed("4: workgroup per (image, 128 rows, 256 px)", [&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(tile_kernel<256, 128>), dim3(g.N * mb * pb256), dim3(256), 0, 0, g, in, out, mb, pb256); });
		timed("4b: workgroup per (image, 64 rows, 256 px)", [&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(tile_kernel<256, 64>), dim3(g.N * ((g.M + 63) / 64) * pb256), dim3(256), 0, 0, g, in, out, (g.M + 63) / 64, pb256); });
		timed("5: linear read + linear write (the ceiling)", [&] { hipLaunchKernelGGL(stream_kernel, dim3(256 * 8), dim3(256), 0, 0, (const u4*)in, (u4*)out, nin / 8, nout / 8); });
		CHECK(hipFree(in)); CHECK(hipFree(out));
	}
	return 0;
}
