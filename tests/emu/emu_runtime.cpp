// TEST INFRASTRUCTURE: runtime half of the HIP emulator (see tests/emu/hip/hip_runtime.h).
#include <hip/hip_runtime.h>
#include <ucontext.h>
#include <sys/mman.h>
#include <time.h>
#include <vector>

uint3_emu threadIdx, blockIdx;
dim3 blockDim, gridDim;

namespace emu {
static const size_t kStack = 256 * 1024;
struct Fiber { ucontext_t ctx; char* stack = nullptr; bool done = false; };
static std::vector<Fiber> fibers;
static ucontext_t main_ctx;
static const std::function<void()>* cur_body = nullptr;
static int cur_tid = 0, nthreads = 0, ndone = 0;
static unsigned block_gen = 0, block_arrived = 0;
static unsigned wave_gen[16], wave_arrived[16], wave_live[16];
static unsigned long progress = 0;
static char xbufs[16][64][32];
static std::vector<char> dyn;
static bool in_kernel = false;

static inline int live() { return nthreads - ndone; }
static void check_block() { if (block_arrived > 0 && (int)block_arrived == live()) { block_arrived = 0; block_gen++; progress++; } }
static void check_wave(int w) { if (wave_arrived[w] > 0 && wave_arrived[w] == wave_live[w]) { wave_arrived[w] = 0; wave_gen[w]++; progress++; } }
static void yield() { const int t = cur_tid; swapcontext(&fibers[t].ctx, &main_ctx); }
static void entry()
{
	(*cur_body)();
	const int t = cur_tid;
	fibers[t].done = true;
	ndone++;
	wave_live[t / 64]--;
	progress++;
	check_block();
	check_wave(t / 64);
	swapcontext(&fibers[t].ctx, &main_ctx);
}
void syncthreads()
{
	const unsigned g = block_gen;
	block_arrived++;
	check_block();
	while (block_gen == g) yield();
}
void wave_sync()
{
	const int w = cur_tid / 64;
	const unsigned g = wave_gen[w];
	wave_arrived[w]++;
	check_wave(w);
	while (wave_gen[w] == g) yield();
}
int lane() { return cur_tid % 64; }
int wave() { return cur_tid / 64; }
int wave_width() { const int w = cur_tid / 64; return std::min(64, nthreads - w * 64); }
void* xbuf(int l) { return xbufs[cur_tid / 64][l]; }
void* dyn_smem() { return dyn.data(); }

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body)
{
	if (in_kernel) { fprintf(stderr, "emu: nested launch\n"); abort(); }
	const int n = (int)(block.x * block.y * block.z);
	if (n <= 0 || n > 1024) { fprintf(stderr, "emu: bad block size %d\n", n); abort(); }
	if ((size_t)grid.x * grid.y * grid.z == 0) return;
	if ((int)fibers.size() < n) fibers.resize(n);
	for (int t = 0; t < n; t++)
		if (!fibers[t].stack) {
			fibers[t].stack = (char*)mmap(0, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
			if (fibers[t].stack == (char*)MAP_FAILED) { perror("mmap"); abort(); }
		}
	dyn.assign(shmem + 64, 0);
	in_kernel = true;
	cur_body = &body;
	blockDim = block; gridDim = grid; nthreads = n;
	for (unsigned bz = 0; bz < grid.z; bz++) for (unsigned by = 0; by < grid.y; by++) for (unsigned bx = 0; bx < grid.x; bx++) {
		blockIdx = uint3_emu{bx, by, bz};
		ndone = 0; block_arrived = 0;
		for (int w = 0; w < 16; w++) { wave_arrived[w] = 0; wave_live[w] = (unsigned)std::max(0, std::min(64, n - w * 64)); }
		for (int t = 0; t < n; t++) {
			fibers[t].done = false;
			getcontext(&fibers[t].ctx);
			fibers[t].ctx.uc_stack.ss_sp = fibers[t].stack;
			fibers[t].ctx.uc_stack.ss_size = kStack;
			fibers[t].ctx.uc_link = &main_ctx;
			makecontext(&fibers[t].ctx, (void (*)())entry, 0);
		}
		int stale = 0;
		while (ndone < n) {
			const unsigned long before = progress;
			for (int t = 0; t < n; t++) {
				if (fibers[t].done) continue;
				cur_tid = t;
				threadIdx = uint3_emu{(unsigned)(t % block.x), (unsigned)((t / block.x) % block.y), (unsigned)(t / (block.x * block.y))};
				swapcontext(&main_ctx, &fibers[t].ctx);
			}
			if (progress == before) { if (++stale > 2) { fprintf(stderr, "emu: deadlock in block (%u,%u,%u): divergent barrier or wave op (%d/%d threads done)\n", bx, by, bz, ndone, n); abort(); } }
			else stale = 0;
		}
	}
	in_kernel = false;
}
} // namespace emu

struct emu_stream_s { int device; };
struct emu_event_s { double t_ms; };
static int g_device = 0;
static int device_count() { const char* e = getenv("NNC_EMU_DEVICE_COUNT"); return e ? atoi(e) : 1; }
static double now_ms() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }

hipError_t hipMalloc(void** p, size_t n) { *p = nullptr; if (posix_memalign(p, 256, n ? n : 256)) return hipErrorOutOfMemory; return hipSuccess; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostRegister(void*, size_t, unsigned) { return hipSuccess; }
hipError_t hipHostUnregister(void*) { return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t) { for (size_t y = 0; y < height; y++) memmove((char*)d + y * dpitch, (const char*)s + y * spitch, width); return hipSuccess; }
hipError_t hipMemcpyPeer(void* d, int, const void* s, int, size_t n) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyPeerAsync(void* d, int, const void* s, int, size_t n, hipStream_t) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
hipError_t hipSetDevice(int d) { if (d < 0 || d >= device_count()) return hipErrorInvalidValue; g_device = d; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = g_device; return hipSuccess; }
hipError_t hipGetDeviceCount(int* n) { *n = device_count(); return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { memset(p, 0, sizeof(*p)); strcpy(p->name, "hip-emulator"); strcpy(p->gcnArchName, "emu"); p->multiProcessorCount = 256; p->totalGlobalMem = 1ull << 34; p->major = 9; p->minor = 5; p->sharedMemPerBlock = 160 * 1024; return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipDeviceCanAccessPeer(int* can, int, int) { *can = 1; return hipSuccess; }
hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = 1ull << 33; *t = 1ull << 34; return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t* s) { *s = new emu_stream_s{g_device}; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { return hipStreamCreate(s); }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipLaunchHostFunc(hipStream_t, hipHostFn_t fn, void* ud) { fn(ud); return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new emu_event_s{0}; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t_ms = now_ms(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t_ms - a->t_ms); return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipPeekAtLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hip-emulator error"; }
