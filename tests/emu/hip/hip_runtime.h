// TEST INFRASTRUCTURE: a single-threaded, fiber-based emulator of the subset of the HIP device
// language + runtime API that ccv_amd/csrc uses, so the *unmodified* kernel sources can be compiled
// for x86 (clang++ -I tests/emu) and checked against the oracle inside the CPU-only test tier.
// Every workgroup runs as N cooperative fibers (ucontext); __syncthreads / wave-level ops
// (__shfl*, MFMA builtins) are rendezvous points between fibers.  Wavefront = 64 lanes.
// The MFMA builtins are modelled exactly as the CDNA4 guide specifies them: a k-ordered fmaf
// chain with the documented lane->element maps.  This file is never part of the product
// library: libnnc_mi355x.so is built by hipcc against the real <hip/hip_runtime.h>.
#pragma once
#include <cstdint>
#include <cstddef>
#include <cstring>
#include <cstdlib>
#include <cstdio>
#include <cmath>
#include <functional>
#include <algorithm>

#define NNC_HIP_EMULATOR 1
#define __host__
#define __device__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define __constant__
#define HIP_KERNEL_NAME(...) __VA_ARGS__
#define HIP_DYNAMIC_SHARED(type, var) type* var = (type*)emu::dyn_smem();
#define warpSize 64

struct dim3 { unsigned x, y, z; constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
struct uint3_emu { unsigned x, y, z; };
extern uint3_emu threadIdx, blockIdx;
extern dim3 blockDim, gridDim;

struct alignas(8) float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct alignas(8) uint2 { unsigned x, y; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600,
       hipErrorStreamCaptureUnsupported = 900, hipErrorStreamCaptureInvalidated = 901, hipErrorStreamCaptureUnjoined = 904, hipErrorStreamCaptureIsolation = 905, hipErrorStreamCaptureImplicit = 906 };
typedef struct emu_stream_s* hipStream_t;
typedef struct emu_event_s* hipEvent_t;
// stream capture (the subset device_rt.cpp's "HIP-graph capture" uses): a capturing stream RECORDS its launches, copies and memsets instead of running them;
// an event recorded on a capturing stream carries the capture, a stream that waits for such an event joins it; hipGraphLaunch runs the recorded nodes in
// the order they were recorded (a topological order of the graph).  What the real runtime refuses during a capture is refused here too, with its codes:
// synchronising / querying a capturing stream, waiting on an outside event from a capturing stream, ending a capture other streams have not rejoined.
typedef struct emu_graph_s* hipGraph_t;
typedef struct emu_graph_exec_s* hipGraphExec_t;
typedef struct emu_graph_node_s* hipGraphNode_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1, hipStreamCaptureModeRelaxed = 2 };
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone = 0, hipStreamCaptureStatusActive = 1, hipStreamCaptureStatusInvalidated = 2 };
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamDefault = 0, hipStreamNonBlocking = 1, hipEventDefault = 0, hipEventDisableTiming = 2, hipHostMallocDefault = 0, hipHostRegisterDefault = 0 };
typedef void (*hipHostFn_t)(void*);
struct hipDeviceProp_t { char name[256]; char gcnArchName[256]; int multiProcessorCount; size_t totalGlobalMem; int major, minor; size_t sharedMemPerBlock; };

namespace emu {
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
void launch_concurrent(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body); // workgroups that wait for each other: a window of them resident
void spin_yield(); // inside a polling loop
void syncthreads();
void wave_sync();
int lane();
int wave();
void* xbuf(int lane); // 32 B per lane exchange slot of the calling fiber's wave
void* dyn_smem();
int wave_width();
bool capturing(hipStream_t st);                                   // the stream records instead of running
void capture_push(hipStream_t st, std::function<void()> node);    // (only on a capturing stream)
bool legacy_stream_use(const char* what);                         // work on the NULL stream: false (refused) while a blocking stream captures
}

template <typename K, typename... Args>
static inline void hipLaunchKernelGGL(K kernel, dim3 grid, dim3 block, size_t shmem, hipStream_t st, Args... args)
{
	if (st && emu::capturing(st)) { emu::capture_push(st, [=]() { emu::launch(grid, block, shmem, [&]() { kernel(args...); }); }); return; } // a kernel node: the arguments by value
	if (!st && !emu::legacy_stream_use("kernel launch")) { fprintf(stderr, "emu: kernel launch on the NULL stream while a stream captures\n"); abort(); }
	emu::launch(grid, block, shmem, [&]() { kernel(args...); });
}

template <typename K, typename... Args>
static inline void emuLaunchConcurrentKernel(K kernel, dim3 grid, dim3 block, size_t shmem, hipStream_t st, Args... args)
{
	if (st && emu::capturing(st)) { emu::capture_push(st, [=]() { emu::launch_concurrent(grid, block, shmem, [&]() { kernel(args...); }); }); return; }
	if (!st && !emu::legacy_stream_use("kernel launch")) { fprintf(stderr, "emu: kernel launch on the NULL stream while a stream captures\n"); abort(); }
	emu::launch_concurrent(grid, block, shmem, [&]() { kernel(args...); });
}

static inline void __syncthreads() { asm volatile("" ::: "memory"); emu::syncthreads(); asm volatile("" ::: "memory"); }
static inline void __threadfence() {}
static inline void __threadfence_block() {}

template <typename T> static inline T emu_shfl_from(T v, int src)
{
	static_assert(sizeof(T) <= 16, "shfl payload");
	memcpy(emu::xbuf(emu::lane()), &v, sizeof(T));
	emu::wave_sync();
	T r;
	if (src < 0 || src >= emu::wave_width()) src = emu::lane();
	memcpy(&r, emu::xbuf(src), sizeof(T));
	emu::wave_sync();
	return r;
}
template <typename T> static inline T __shfl(T v, int src, int width = 64) { const int l = emu::lane(); return emu_shfl_from(v, (l / width) * width + (src % width)); }
// separately rounded fp32 operations (no contraction into fma)
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
template <typename T> static inline T __shfl_xor(T v, int mask, int width = 64) { return emu_shfl_from(v, emu::lane() ^ mask); }
template <typename T> static inline T __shfl_down(T v, unsigned d, int width = 64) { const int l = emu::lane(); const int s = l + (int)d; return emu_shfl_from(v, ((s / width) == (l / width)) ? s : l); }
template <typename T> static inline T __shfl_up(T v, unsigned d, int width = 64) { const int l = emu::lane(); const int s = l - (int)d; return emu_shfl_from(v, (s >= 0 && (s / width) == (l / width)) ? s : l); }
static inline unsigned long long __ballot(int pred)
{
	unsigned long long r = 0;
	for (int i = 0; i < 64; i++) { int p = emu_shfl_from(pred, i); if (i < emu::wave_width() && p) r |= 1ull << i; }
	return r;
}

typedef float emu_floatx16 __attribute__((ext_vector_type(16)));
typedef float emu_floatx4 __attribute__((ext_vector_type(4)));
// v_mfma_f32_32x32x2_f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]; D: col=l&31, row=(r&3)+8*(r>>2)+4*(l>>5)
static inline emu_floatx16 emu_mfma_f32_32x32x2f32(float a, float b, emu_floatx16 c, int, int, int)
{
	struct ab { float a, b; } mine = { a, b };
	memcpy(emu::xbuf(emu::lane()), &mine, sizeof(mine));
	emu::wave_sync();
	const int l = emu::lane(), j = l & 31, hi = l >> 5;
	for (int r = 0; r < 16; r++) {
		const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
		float acc = c[r];
		for (int k = 0; k < 2; k++) {
			ab A, B;
			memcpy(&A, emu::xbuf(i + 32 * k), sizeof(A));
			memcpy(&B, emu::xbuf(j + 32 * k), sizeof(B));
			acc = fmaf(A.a, B.b, acc);
		}
		c[r] = acc;
	}
	emu::wave_sync();
	return c;
}
// v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; D: col=l&15, row=(l>>4)*4+r
static inline emu_floatx4 emu_mfma_f32_16x16x4f32(float a, float b, emu_floatx4 c, int, int, int)
{
	struct ab { float a, b; } mine = { a, b };
	memcpy(emu::xbuf(emu::lane()), &mine, sizeof(mine));
	emu::wave_sync();
	const int l = emu::lane(), j = l & 15, q = l >> 4;
	for (int r = 0; r < 4; r++) {
		const int i = q * 4 + r;
		float acc = c[r];
		for (int k = 0; k < 4; k++) {
			ab A, B;
			memcpy(&A, emu::xbuf(i + 16 * k), sizeof(A));
			memcpy(&B, emu::xbuf(j + 16 * k), sizeof(B));
			acc = fmaf(A.a, B.b, acc);
		}
		c[r] = acc;
	}
	emu::wave_sync();
	return c;
}
// v_mfma_f32_32x32x16_f16 (gfx950): lane l = (row / column l & 31, k half l >> 5) supplies 8 halves per operand; D as the 32x32x2 form.
// Products of two halves are exact in fp32; the sum runs in k order in fp32 (the hardware's internal order is not specified:
// tests of the half-precision core compare within a tolerance).
typedef _Float16 emu_halfx8 __attribute__((ext_vector_type(8)));
static inline emu_floatx16 emu_mfma_f32_32x32x16_f16(emu_halfx8 a, emu_halfx8 b, emu_floatx16 c)
{
	struct ab { emu_halfx8 a, b; } mine = { a, b };
	static_assert(sizeof(ab) == 32, "exchange slot");
	memcpy(emu::xbuf(emu::lane()), &mine, sizeof(mine));
	emu::wave_sync();
	const int l = emu::lane(), j = l & 31, hi = l >> 5;
	for (int r = 0; r < 16; r++) {
		const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
		float acc = c[r];
		for (int h = 0; h < 2; h++) {
			ab A, B;
			memcpy(&A, emu::xbuf(i + 32 * h), sizeof(A));
			memcpy(&B, emu::xbuf(j + 32 * h), sizeof(B));
			for (int e = 0; e < 8; e++) acc += (float)A.a[e] * (float)B.b[e];
		}
		c[r] = acc;
	}
	emu::wave_sync();
	return c;
}
// The bf16 form on packed operands (two bf16 per dword, element 2i in the low half of dword i): products of two 8-bit significands are exact in fp32.
typedef unsigned int emu_bf16x8 __attribute__((ext_vector_type(4)));
static inline emu_floatx16 emu_mfma_f32_32x32x16_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_floatx16 c)
{
	struct ab { emu_bf16x8 a, b; } mine = { a, b };
	static_assert(sizeof(ab) == 32, "exchange slot");
	memcpy(emu::xbuf(emu::lane()), &mine, sizeof(mine));
	emu::wave_sync();
	const int l = emu::lane(), j = l & 31, hi = l >> 5;
	auto at = [](const emu_bf16x8& v, const int e) { const unsigned u = (e & 1) ? (v[e >> 1] & 0xffff0000u) : (v[e >> 1] << 16); float f; memcpy(&f, &u, 4); return f; };
	for (int r = 0; r < 16; r++) {
		const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
		float acc = c[r];
		for (int h = 0; h < 2; h++) {
			ab A, B;
			memcpy(&A, emu::xbuf(i + 32 * h), sizeof(A));
			memcpy(&B, emu::xbuf(j + 32 * h), sizeof(B));
			for (int e = 0; e < 8; e++) acc += at(A.a, e) * at(B.b, e);
		}
		c[r] = acc;
	}
	emu::wave_sync();
	return c;
}
// LDS-DMA (buffer_load ... lds): every lane copies `size` bytes from base + voffset + soffset + inst_offset to lds + lane * size;
// a lane whose offset reaches past num_records writes zeros (raw-buffer range check).  Synchronous here: the emulator cannot
// show a missing s_waitcnt vmcnt -- only the MI355X tier can.
struct emu_buffer_rsrc { const char* base; unsigned num_records; };
typedef emu_buffer_rsrc __amdgpu_buffer_rsrc_t;
static inline emu_buffer_rsrc __builtin_amdgcn_make_buffer_rsrc(void* p, short, unsigned num_records, int) { return emu_buffer_rsrc{ (const char*)p, num_records }; }
static inline void __builtin_amdgcn_raw_ptr_buffer_load_lds(emu_buffer_rsrc r, __attribute__((address_space(3))) void* lds, unsigned size, unsigned voffset, unsigned soffset, unsigned ioffset, unsigned)
{
	char* const d = (char*)lds + (size_t)emu::lane() * size;
	const unsigned long long off = (unsigned long long)voffset + ioffset;
	if (off + size > r.num_records || off + soffset + size > r.num_records) memset(d, 0, size);
	else memcpy(d, r.base + off + soffset, size);
}
typedef unsigned int emu_u4 __attribute__((ext_vector_type(4)));
static inline void __builtin_amdgcn_raw_buffer_store_b128(emu_u4 v, emu_buffer_rsrc r, unsigned voffset, unsigned soffset, int)
{ // out-of-range lanes are dropped
	if ((unsigned long long)voffset + 16 > r.num_records || (unsigned long long)voffset + soffset + 16 > r.num_records) return;
	memcpy((char*)r.base + voffset + soffset, &v, 16);
}
static inline emu_u4 __builtin_amdgcn_raw_buffer_load_b128(emu_buffer_rsrc r, unsigned voffset, unsigned soffset, int)
{ // raw buffer (stride 0): the range check is per DWORD -- an out-of-range dword reads zero, the others are delivered
	emu_u4 v = { 0u, 0u, 0u, 0u };
	for (int d = 0; d < 4; d++) {
		const unsigned long long off = (unsigned long long)voffset + soffset + 4ull * d;
		if ((unsigned long long)voffset + 4ull * d + 4 > r.num_records || off + 4 > r.num_records) continue;
		unsigned w; memcpy(&w, r.base + off, 4); v[d] = w;
	}
	return v;
}
typedef unsigned int emu_u2 __attribute__((ext_vector_type(2)));
static inline emu_u2 __builtin_amdgcn_raw_buffer_load_b64(emu_buffer_rsrc r, unsigned voffset, unsigned soffset, int)
{ // (range check per dword, as for the 16-byte load)
	emu_u2 v = { 0u, 0u };
	for (int d = 0; d < 2; d++) {
		const unsigned long long off = (unsigned long long)voffset + soffset + 4ull * d;
		if ((unsigned long long)voffset + 4ull * d + 4 > r.num_records || off + 4 > r.num_records) continue;
		unsigned w; memcpy(&w, r.base + off, 4); v[d] = w;
	}
	return v;
}
static inline unsigned __builtin_amdgcn_raw_buffer_load_b32(emu_buffer_rsrc r, unsigned voffset, unsigned soffset, int)
{
	unsigned w = 0;
	if ((unsigned long long)voffset + 4 <= r.num_records && (unsigned long long)voffset + soffset + 4 <= r.num_records) memcpy(&w, r.base + voffset + soffset, 4);
	return w;
}
static inline void __builtin_amdgcn_raw_buffer_store_b64(emu_u2 v, emu_buffer_rsrc r, unsigned voffset, unsigned soffset, int)
{
	if ((unsigned long long)voffset + 8 > r.num_records || (unsigned long long)voffset + soffset + 8 > r.num_records) return;
	memcpy((char*)r.base + voffset + soffset, &v, 8);
}
static inline void __builtin_amdgcn_raw_buffer_store_b32(unsigned v, emu_buffer_rsrc r, unsigned voffset, unsigned soffset, int)
{
	if ((unsigned long long)voffset + 4 > r.num_records || (unsigned long long)voffset + soffset + 4 > r.num_records) return;
	memcpy((char*)r.base + voffset + soffset, &v, 4);
}
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
// (the "memory" clobbers: a workgroup's LDS is a function-local static whose address never leaves the kernel, so without them
// the x86 compiler may move a lane's LDS reads above the rendezvous -- other fibers' writes are invisible to its analysis)
static inline void emu_wave_barrier() { asm volatile("" ::: "memory"); emu::wave_sync(); asm volatile("" ::: "memory"); }
#define __builtin_amdgcn_wave_barrier() emu_wave_barrier()
#define __builtin_amdgcn_s_barrier() __syncthreads()
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_amdgcn_mfma_f32_32x32x2f32 emu_mfma_f32_32x32x2f32
#define __builtin_amdgcn_mfma_f32_16x16x4f32 emu_mfma_f32_16x16x4f32
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(a, b, c) ((void)0)
#define __builtin_nontemporal_load(p) (*(p))
#define __builtin_nontemporal_store(v, p) (*(p) = (v))

static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }
static inline float rsqrtf(float a) { return 1.0f / sqrtf(a); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }

// ---------------------------------------------------------------- runtime API (host memory)
hipError_t hipMalloc(void** p, size_t n);
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t n, unsigned flags = 0);
hipError_t hipHostFree(void* p);
hipError_t hipHostRegister(void* p, size_t n, unsigned flags);
hipError_t hipHostUnregister(void* p);
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st = 0);
hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, hipMemcpyKind k, hipStream_t st = 0);
hipError_t hipMemcpyPeer(void* d, int dd, const void* s, int sd, size_t n);
hipError_t hipMemcpyPeerAsync(void* d, int dd, const void* s, int sd, size_t n, hipStream_t st = 0);
hipError_t hipMemset(void* d, int v, size_t n);
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st = 0);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int* d);
hipError_t hipGetDeviceCount(int* n);
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int d);
hipError_t hipDeviceSynchronize();
hipError_t hipDeviceCanAccessPeer(int* can, int a, int b);
hipError_t hipDeviceEnablePeerAccess(int peer, unsigned flags);
hipError_t hipMemGetInfo(size_t* free_, size_t* total);
hipError_t hipStreamCreate(hipStream_t* s);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags = 0);
hipError_t hipStreamQuery(hipStream_t s);
hipError_t hipLaunchHostFunc(hipStream_t s, hipHostFn_t fn, void* ud);
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = 0);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventQuery(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode mode);
hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t* graph);
hipError_t hipStreamIsCapturing(hipStream_t s, hipStreamCaptureStatus* status);
hipError_t hipGraphInstantiate(hipGraphExec_t* exec, hipGraph_t graph, hipGraphNode_t* error_node, char* log, size_t log_size);
hipError_t hipGraphLaunch(hipGraphExec_t exec, hipStream_t s);
hipError_t hipGraphGetNodes(hipGraph_t graph, hipGraphNode_t* nodes, size_t* count);
hipError_t hipGraphExecDestroy(hipGraphExec_t exec);
hipError_t hipGraphDestroy(hipGraph_t graph);
hipError_t hipGetLastError();
hipError_t hipPeekAtLastError();
const char* hipGetErrorString(hipError_t e);
