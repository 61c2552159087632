// TEST INFRASTRUCTURE: runtime half of the HIP emulator (see tests/emu/hip/hip_runtime.h).
#include <hip/hip_runtime.h>
#include <ucontext.h>
#include <sys/mman.h>
#include <time.h>
#include <vector>
#include <mutex>
#include <atomic>

// ThreadSanitizer build (make emu-tsan): every lane is a TSAN fiber and every switch a synchronisation -- the lanes of a kernel run one after the other
// on the launching thread, so they are never in a race with each other; what TSAN then sees is the HOST side of the library (peephole.cpp, cmd_comm.cpp,
// device_rt.cpp) called from the tests' threads.
#if defined(__has_feature)
#if __has_feature(thread_sanitizer)
#define EMU_TSAN 1
extern "C" { void* __tsan_get_current_fiber(void); void* __tsan_create_fiber(unsigned flags); void __tsan_destroy_fiber(void* fiber); void __tsan_switch_to_fiber(void* fiber, unsigned flags); }
#endif
#endif

uint3_emu threadIdx, blockIdx;
dim3 blockDim, gridDim;

namespace emu {
static const size_t kStack = 256 * 1024;
struct Fiber { ucontext_t ctx; char* stack = nullptr; bool done = false; void* tsan = nullptr; };
static void* main_tsan = nullptr;
// ONE kernel at a time, whatever thread launches it (a loader thread's conversion kernel next to the training thread's): the device is one device
static std::recursive_mutex launch_mutex;
// Everything one resident workgroup owns.  The ordinary launch runs the grid's workgroups one after the other through ONE of these; launch_concurrent
// (kernels whose workgroups talk to each other inside a launch: spin-waits on flags other workgroups of the grid publish) keeps a WINDOW of them
// resident and interleaves their fibers.
struct Block {
	std::vector<Fiber> fibers;
	int ndone = 0;
	unsigned block_gen = 0, block_arrived = 0;
	unsigned wave_gen[16] = {}, wave_arrived[16] = {}, wave_live[16] = {};
	char xbufs[16][64][32];
	std::vector<char> dyn;
	uint3_emu bidx{0, 0, 0};
	bool active = false;
};
static Block seq_block;
static Block* cur = &seq_block;
static ucontext_t main_ctx;
static const std::function<void()>* cur_body = nullptr;
static int cur_tid = 0, nthreads = 0;
static unsigned long progress = 0, threads_finished = 0;
static bool in_kernel = false;

static inline int live() { return nthreads - cur->ndone; }
static void check_block() { if (cur->block_arrived > 0 && (int)cur->block_arrived == live()) { cur->block_arrived = 0; cur->block_gen++; progress++; } }
static void check_wave(int w) { if (cur->wave_arrived[w] > 0 && cur->wave_arrived[w] == cur->wave_live[w]) { cur->wave_arrived[w] = 0; cur->wave_gen[w]++; progress++; } }
static inline void to_main_tsan()
{
#ifdef EMU_TSAN
	__tsan_switch_to_fiber(main_tsan, 0);
#endif
}
static void yield() { Block* const b = cur; const int t = cur_tid; to_main_tsan(); swapcontext(&b->fibers[t].ctx, &main_ctx); }
static void entry()
{
	(*cur_body)();
	Block* const b = cur;
	const int t = cur_tid;
	b->fibers[t].done = true;
	b->ndone++;
	b->wave_live[t / 64]--;
	progress++;
	threads_finished++;
	check_block();
	check_wave(t / 64);
	to_main_tsan();
	swapcontext(&b->fibers[t].ctx, &main_ctx);
}
void syncthreads()
{
	Block* const b = cur; // (a fiber stays in its block: `cur` is that block again whenever this fiber runs)
	const unsigned g = b->block_gen;
	b->block_arrived++;
	check_block();
	while (b->block_gen == g) yield();
}
void wave_sync()
{
	Block* const b = cur;
	const int w = cur_tid / 64;
	const unsigned g = b->wave_gen[w];
	b->wave_arrived[w]++;
	check_wave(w);
	while (b->wave_gen[w] == g) yield();
}
void spin_yield() { yield(); } // a polling loop's s_sleep: let the other resident workgroups run (no progress of its own)
int lane() { return cur_tid % 64; }
int wave() { return cur_tid / 64; }
int wave_width() { const int w = cur_tid / 64; return std::min(64, nthreads - w * 64); }
void* xbuf(int l) { return cur->xbufs[cur_tid / 64][l]; }
void* dyn_smem() { return cur->dyn.data(); }

static void block_prepare(Block& b, const int n, const size_t shmem, const uint3_emu bidx)
{
	if ((int)b.fibers.size() < n) b.fibers.resize(n);
	for (int t = 0; t < n; t++)
		if (!b.fibers[t].stack) {
			b.fibers[t].stack = (char*)mmap(0, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
			if (b.fibers[t].stack == (char*)MAP_FAILED) { perror("mmap"); abort(); }
		}
	b.dyn.assign(shmem + 64, 0);
	b.bidx = bidx;
	b.ndone = 0; b.block_arrived = 0;
	for (int w = 0; w < 16; w++) { b.wave_arrived[w] = 0; b.wave_live[w] = (unsigned)std::max(0, std::min(64, n - w * 64)); }
	for (int t = 0; t < n; t++) {
		b.fibers[t].done = false;
		getcontext(&b.fibers[t].ctx);
		b.fibers[t].ctx.uc_stack.ss_sp = b.fibers[t].stack;
		b.fibers[t].ctx.uc_stack.ss_size = kStack;
		b.fibers[t].ctx.uc_link = &main_ctx;
		makecontext(&b.fibers[t].ctx, (void (*)())entry, 0);
#ifdef EMU_TSAN
		if (b.fibers[t].tsan) __tsan_destroy_fiber(b.fibers[t].tsan); // (a fresh shadow stack per use: a lane leaves its function by a context switch, never by returning)
		b.fibers[t].tsan = __tsan_create_fiber(0);
#endif
	}
	b.active = true;
}
// one sweep: every unfinished fiber of the block runs until its next rendezvous
static void block_sweep(Block& b, const int n, const dim3 block)
{
	cur = &b;
	blockIdx = b.bidx;
	for (int t = 0; t < n; t++) {
		if (b.fibers[t].done) continue;
		cur_tid = t;
		threadIdx = uint3_emu{(unsigned)(t % block.x), (unsigned)((t / block.x) % block.y), (unsigned)(t / (block.x * block.y))};
#ifdef EMU_TSAN
		main_tsan = __tsan_get_current_fiber();
		__tsan_switch_to_fiber(b.fibers[t].tsan, 0);
#endif
		swapcontext(&main_ctx, &b.fibers[t].ctx);
	}
}
static int launch_begin(dim3 grid, dim3 block, const std::function<void()>& body)
{
	if (in_kernel) { fprintf(stderr, "emu: nested launch\n"); abort(); }
	const int n = (int)(block.x * block.y * block.z);
	if (n <= 0 || n > 1024) { fprintf(stderr, "emu: bad block size %d\n", n); abort(); }
	in_kernel = true;
	cur_body = &body;
	blockDim = block; gridDim = grid; nthreads = n;
	return n;
}

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body)
{
	if ((size_t)grid.x * grid.y * grid.z == 0) return;
	std::lock_guard<std::recursive_mutex> lock(launch_mutex);
	const int n = launch_begin(grid, block, body);
	for (unsigned bz = 0; bz < grid.z; bz++) for (unsigned by = 0; by < grid.y; by++) for (unsigned bx = 0; bx < grid.x; bx++) {
		block_prepare(seq_block, n, shmem, uint3_emu{bx, by, bz});
		int stale = 0;
		while (seq_block.ndone < n) {
			const unsigned long before = progress;
			block_sweep(seq_block, n, block);
			if (progress == before) { if (++stale > 2) { fprintf(stderr, "emu: deadlock in block (%u,%u,%u): divergent barrier or wave op (%d/%d threads done)\n", bx, by, bz, seq_block.ndone, n); abort(); } }
			else stale = 0;
		}
	}
	cur = &seq_block;
	in_kernel = false;
}

// Workgroups that wait for each other inside a launch.  A window of workgroups is resident at a time (NNC_EMU_RESIDENT_BLOCKS, default 32); a finished
// one is replaced by the next of the dispatch order -- forward by default, NNC_EMU_DISPATCH_ORDER=reverse / shuffle to show that a kernel's protocol does not
// depend on it (HIP promises no dispatch order).  Such kernels keep their LDS in the dynamic region: `__shared__` statics are ONE object in this emulator.
void launch_concurrent(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body)
{
	const size_t total = (size_t)grid.x * grid.y * grid.z;
	if (total == 0) return;
	std::lock_guard<std::recursive_mutex> lock(launch_mutex);
	const int n = launch_begin(grid, block, body);
	static std::vector<Block*> window;
	const char* const we = getenv("NNC_EMU_RESIDENT_BLOCKS");
	size_t resident = we ? (size_t)atol(we) : 32;
	if (resident < 1) resident = 1;
	if (resident > total) resident = total;
	while (window.size() < resident) window.push_back(new Block);
	std::vector<size_t> order(total);
	for (size_t i = 0; i < total; i++) order[i] = i;
	const char* const oe = getenv("NNC_EMU_DISPATCH_ORDER");
	if (oe && !strcmp(oe, "reverse")) std::reverse(order.begin(), order.end());
	else if (oe && !strcmp(oe, "shuffle")) { unsigned long long st = 0x9e3779b97f4a7c15ULL; for (size_t i = total; i > 1; i--) { st = st * 6364136223846793005ULL + 1442695040888963407ULL; std::swap(order[i - 1], order[(st >> 33) % i]); } }
	size_t next = 0, finished = 0;
	for (size_t i = 0; i < resident; i++) window[i]->active = false;
	// (a polling loop's wave votes count as `progress`: the deadlock test here is "no THREAD has finished and no workgroup was admitted for 20 000 sweeps")
	int stale = 0;
	while (finished < total) {
		const unsigned long before = threads_finished + next;
		for (size_t i = 0; i < resident; i++) {
			Block& b = *window[i];
			if (!b.active) {
				if (next >= total) continue;
				const size_t id = order[next++];
				block_prepare(b, n, shmem, uint3_emu{(unsigned)(id % grid.x), (unsigned)((id / grid.x) % grid.y), (unsigned)(id / ((size_t)grid.x * grid.y))});
			}
			block_sweep(b, n, block);
			if (b.ndone == n) { b.active = false; finished++; }
		}
		if (threads_finished + next == before) { if (++stale > 20000) { fprintf(stderr, "emu: deadlock in a concurrent launch: %zu of %zu workgroups finished, %zu resident, none makes progress\n", finished, total, resident); abort(); } }
		else stale = 0;
	}
	cur = &seq_block;
	in_kernel = false;
}
} // namespace emu

// ---- streams, events and stream capture -----------------------------------------------------------------------------------------------------------
struct emu_graph_s {
	std::vector<std::function<void()> > nodes; // in the order they were recorded: a topological order
	std::vector<emu_stream_s*> members;        // the origin first, then the streams that joined through an event
	bool active = true;
};
struct emu_graph_exec_s { std::vector<std::function<void()> > nodes; };
struct emu_stream_s {
	int device; emu_graph_s* cap = nullptr; unsigned long seq = 0, joined_seq = 0;
	// ROCm 7.2's bookkeeping, modelled because it has a flaw this library must stay clear of: a NON-origin stream that waits for a captured event is filed under
	// the stream the event was recorded on ("parallel capture streams"), at every wait; hipStreamEndCapture walks those lists recursively.  Two non-origin streams
	// that wait for each other's events file each other -- the walk never ends (stack overflow inside hipStreamEndCapture: the first MI355X run of the
	// reference host's ccv_cnnp_model_fit step, whose schedule's main stream and side streams do exactly that).  Here: the capture is refused with a message.
	std::vector<emu_stream_s*> parallel;
	int walk = 0;
};
struct emu_event_s { double t_ms; emu_graph_s* cap = nullptr; emu_stream_s* on = nullptr; unsigned long rec_seq = 0; };
static std::atomic<int> g_captures_active(0);
namespace emu {
bool capturing(hipStream_t st) { return st && st->cap; }
void capture_push(hipStream_t st, std::function<void()> node) { st->cap->nodes.push_back(std::move(node)); st->seq++; if (getenv("NNC_EMU_CAPTURE_LOG")) fprintf(stderr, "emu-cap: NODE   stream %p seq %lu\n", (void*)st, st->seq); }
bool legacy_stream_use(const char* what)
{ // the streams this library makes are BLOCKING streams (ordered against the NULL stream): NULL-stream work while one of them records would be an implicit
  // dependency on the capture -- the MI355X's runtime refuses it (a blocking hipMemcpy inside a capture aborted the first GPU run of tests/test_capture.py)
	if (g_captures_active.load() > 0) { if (getenv("NNC_EMU_CAPTURE_NOTES")) fprintf(stderr, "emu: NULL-stream %s while a stream captures: refused\n", what); return false; }
	return true;
}
}
static thread_local int g_device = 0; // (HIP: the current device is per host thread -- found by the ThreadSanitizer run of the two-thread tests)
static int device_count() { const char* e = getenv("NNC_EMU_DEVICE_COUNT"); return e ? atoi(e) : 1; }
static double now_ms() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }

// EMU_MALLOC_FAIL_NEXT=<n> in the environment of a test: the next n device allocations fail (the pressure path of nnc_mi355x_malloc)
static std::atomic<int> g_emu_malloc_fail(-1);
hipError_t hipMalloc(void** p, size_t n)
{
	*p = nullptr;
	if (g_emu_malloc_fail.load() < 0) { const char* e = getenv("EMU_MALLOC_FAIL_NEXT"); g_emu_malloc_fail.store(e ? atoi(e) : 0); }
	if (g_emu_malloc_fail.load() > 0) { g_emu_malloc_fail.fetch_sub(1); return hipErrorOutOfMemory; }
	if (posix_memalign(p, 256, n ? n : 256)) return hipErrorOutOfMemory;
	return hipSuccess;
}
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostRegister(void*, size_t, unsigned) { return hipSuccess; }
hipError_t hipHostUnregister(void*) { return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (!emu::legacy_stream_use("blocking copy")) return hipErrorStreamCaptureImplicit; memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t st) { if (emu::capturing(st)) { emu::capture_push(st, [=]() { memmove(d, s, n); }); return hipSuccess; } memmove(d, s, n); return hipSuccess; } // (a captured copy reads its SOURCE POINTER at every replay, as the real node does)
hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t st)
{
	auto run = [=]() { for (size_t y = 0; y < height; y++) memmove((char*)d + y * dpitch, (const char*)s + y * spitch, width); };
	if (emu::capturing(st)) { emu::capture_push(st, run); return hipSuccess; }
	run();
	return hipSuccess;
}
hipError_t hipMemcpyPeer(void* d, int, const void* s, int, size_t n) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyPeerAsync(void* d, int, const void* s, int, size_t n, hipStream_t st) { if (emu::capturing(st)) { emu::capture_push(st, [=]() { memmove(d, s, n); }); return hipSuccess; } memmove(d, s, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { if (!emu::legacy_stream_use("blocking fill")) return hipErrorStreamCaptureImplicit; memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st) { if (emu::capturing(st)) { emu::capture_push(st, [=]() { memset(d, v, n); }); return hipSuccess; } memset(d, v, n); return hipSuccess; }
hipError_t hipSetDevice(int d) { if (d < 0 || d >= device_count()) return hipErrorInvalidValue; g_device = d; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = g_device; return hipSuccess; }
hipError_t hipGetDeviceCount(int* n) { *n = device_count(); return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { memset(p, 0, sizeof(*p)); strcpy(p->name, "hip-emulator"); strcpy(p->gcnArchName, "emu"); p->multiProcessorCount = 256; p->totalGlobalMem = 1ull << 34; p->major = 9; p->minor = 5; p->sharedMemPerBlock = 160 * 1024; return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipDeviceCanAccessPeer(int* can, int, int) { *can = 1; return hipSuccess; }
hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = 1ull << 33; *t = 1ull << 34; return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t* s) { *s = new emu_stream_s; (*s)->device = g_device; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { return hipStreamCreate(s); }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t s) { return emu::capturing(s) ? hipErrorStreamCaptureUnsupported : hipSuccess; }
static bool cap_log() { static int on = -1; if (on < 0) on = getenv("NNC_EMU_CAPTURE_LOG") ? 1 : 0; return on == 1; }
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned)
{
	if (cap_log() && g_captures_active.load() > 0) fprintf(stderr, "emu-cap: WAIT   stream %p (capturing %d, seq %lu) event %p (captured %d, recorded on %p at seq %lu)\n", (void*)s, s && s->cap ? 1 : 0, s ? s->seq : 0ul, (void*)e, e->cap && e->cap->active ? 1 : 0, (void*)e->on, e->rec_seq);
	if (e->cap && e->cap->active) { // a captured event: the waiting stream joins the capture (or already belongs to it)
		if (!s) return hipErrorStreamCaptureImplicit;
		if (s->cap && s->cap != e->cap) return hipErrorStreamCaptureIsolation;
		if (!s->cap) { s->cap = e->cap; e->cap->members.push_back(s); }
		if (s != e->cap->members[0] && e->on && e->on != s) e->on->parallel.push_back(s);
		s->seq++;
		if (e->on && e->rec_seq > e->on->joined_seq) e->on->joined_seq = e->rec_seq; // everything the recording stream had done up to the record is joined to somebody
		return hipSuccess;
	}
	if (s && s->cap) return hipErrorStreamCaptureIsolation; // a capturing stream may not depend on work outside its graph
	return hipSuccess;
}
hipError_t hipStreamQuery(hipStream_t s) { return emu::capturing(s) ? hipErrorStreamCaptureUnsupported : hipSuccess; }
hipError_t hipLaunchHostFunc(hipStream_t st, hipHostFn_t fn, void* ud) { if (emu::capturing(st)) { emu::capture_push(st, [=]() { fn(ud); }); return hipSuccess; } fn(ud); return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new emu_event_s; (*e)->t_ms = 0; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) { if (cap_log() && g_captures_active.load() > 0) fprintf(stderr, "emu-cap: RECORD stream %p (capturing %d, seq %lu) event %p\n", (void*)s, s && s->cap ? 1 : 0, s ? s->seq : 0ul, (void*)e); e->t_ms = now_ms(); e->cap = s ? s->cap : nullptr; e->on = s; e->rec_seq = s ? s->seq : 0; return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t e) { return (e->cap && e->cap->active) ? hipErrorStreamCaptureUnsupported : hipSuccess; }
hipError_t hipEventQuery(hipEvent_t e) { return (e->cap && e->cap->active) ? hipErrorStreamCaptureUnsupported : hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t_ms - a->t_ms); return hipSuccess; }
hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode)
{
	if (!s || s->cap) return hipErrorInvalidValue;
	s->cap = new emu_graph_s;
	s->cap->members.push_back(s);
	s->seq = s->joined_seq = 0;
	g_captures_active.fetch_add(1);
	return hipSuccess;
}
hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t* graph)
{
	*graph = nullptr;
	if (!s || !s->cap || s->cap->members[0] != s) return hipErrorInvalidValue; // (only the origin stream ends a capture)
	emu_graph_s* const g = s->cap;
	// the runtime's recursive walk over the "parallel capture streams": a cycle = endless recursion on the MI355X
	std::function<bool(emu_stream_s*)> cyclic = [&](emu_stream_s* const m) {
		if (m->walk == 1) return true;
		if (m->walk == 2) return false;
		m->walk = 1;
		for (emu_stream_s* c : m->parallel) if (cyclic(c)) return true;
		m->walk = 2;
		return false;
	};
	bool endless = false;
	for (emu_stream_s* m : g->members) if (cyclic(m)) endless = true;
	for (emu_stream_s* m : g->members) { m->parallel.clear(); m->walk = 0; }
	if (endless) {
		fprintf(stderr, "emu: hipStreamEndCapture: two non-origin streams of the capture waited for each other's events -- ROCm 7.2's runtime recurses without end here (stack overflow on the MI355X); refused\n");
		for (emu_stream_s* m : g->members) { m->cap = nullptr; m->seq = m->joined_seq = 0; }
		g->active = false;
		g_captures_active.fetch_sub(1);
		g->nodes.clear();
		return hipErrorStreamCaptureInvalidated;
	}
	bool unjoined = false;
	for (size_t i = 1; i < g->members.size(); i++) if (g->members[i]->seq != g->members[i]->joined_seq) unjoined = true; // a stream that joined holds work (or a dependency) nobody waited for
	for (emu_stream_s* m : g->members) { m->cap = nullptr; m->seq = m->joined_seq = 0; }
	g->active = false;
	g_captures_active.fetch_sub(1);
	if (unjoined) { g->nodes.clear(); return hipErrorStreamCaptureUnjoined; } // (the graph object leaks: events may still point at it)
	*graph = g;
	return hipSuccess;
}
hipError_t hipStreamIsCapturing(hipStream_t s, hipStreamCaptureStatus* status) { *status = emu::capturing(s) ? hipStreamCaptureStatusActive : hipStreamCaptureStatusNone; return hipSuccess; }
hipError_t hipGraphInstantiate(hipGraphExec_t* exec, hipGraph_t graph, hipGraphNode_t*, char*, size_t) { *exec = new emu_graph_exec_s; (*exec)->nodes = graph->nodes; return hipSuccess; }
hipError_t hipGraphLaunch(hipGraphExec_t exec, hipStream_t s)
{
	if (emu::capturing(s)) return hipErrorStreamCaptureUnsupported; // (child graphs: not modelled)
	for (const std::function<void()>& n : exec->nodes) n();
	return hipSuccess;
}
hipError_t hipGraphGetNodes(hipGraph_t graph, hipGraphNode_t*, size_t* count) { *count = graph->nodes.size(); return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t exec) { delete exec; return hipSuccess; }
hipError_t hipGraphDestroy(hipGraph_t graph) { graph->nodes.clear(); return hipSuccess; } // (the shell stays: an event recorded in the capture still points at it)
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipPeekAtLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t e)
{
	switch (e) {
	case hipSuccess: return "hipSuccess";
	case hipErrorStreamCaptureUnsupported: return "hipErrorStreamCaptureUnsupported (emulator: the operation is not permitted on a capturing stream)";
	case hipErrorStreamCaptureUnjoined: return "hipErrorStreamCaptureUnjoined (emulator: a stream that joined the capture was not joined back)";
	case hipErrorStreamCaptureIsolation: return "hipErrorStreamCaptureIsolation (emulator: a capturing stream waited for an event recorded outside its capture)";
	case hipErrorStreamCaptureImplicit: return "hipErrorStreamCaptureImplicit (emulator)";
	default: return "hip-emulator error";
	}
}
