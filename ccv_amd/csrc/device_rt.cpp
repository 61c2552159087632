// Device runtime behind the plugin surface: memory, streams, events, workspace, host callbacks.
// Replaces lib/nnc/gpu/ccv_nnc_compat.cu:101-655 of the reference (CUDA runtime + cuBLAS/cuDNN handle pools);
// here a stream context owns just a HIP stream and a grow-only workspace -- there are no vendor-library handles.
#include "common.h"
#include <atomic>
#include <pthread.h>
#include <dlfcn.h>
#include <vector>
#include <list>
#include <deque>
#include <mutex>
#include <unordered_map>

namespace {

// What a stream context owns per device: a HIP stream and a grow-only scratch buffer (no vendor-library handles).
struct device_local_t {
	hipStream_t stream;
	void* workspace;
	size_t workspace_size;
	int device;
	void* staging;        // second grow-only arena: the fp32 images of half-precision tensors (half_stage.cpp) -- they must survive
	size_t staging_size;  // whatever the command underneath asks of the workspace (a growing workspace is freed and re-allocated)
	void* palette;        // third grow-only arena: the dense images of palettized (CCV_QX) inputs (palette.cpp) -- the command underneath may grow
	size_t palette_size;  // the workspace AND the staging arena (a half-precision GEMM with palettized weights does both)
	void* cluster_sync;   // the words the workgroups of ONE launch on this stream hand each other (nnc::cluster_sync_of): never scratch, never moved
	unsigned cluster_epoch;
	unsigned long cluster_capture; // the capture in which the area was last cleared (cluster_sync_of)
	unsigned long capture_alias;   // the capture (its number) in which this stream's work goes to the recording stream instead (effective_stream)
	unsigned long comm_seen; // the overlapped gradient all-reduces this stream has been ordered behind (cmd_comm.cpp comm_overlap_join)
};
// Layout contract with the reference host (lib/nnc/ccv_nnc_stream.c:15-20, lib/nnc/gpu/ccv_nnc_compat.cu:286-299):
// the host allocates `super` + a {size_t, void*} CPU-workspace tail for EVERY context (CPU contexts included) and, under
// its GPU configuration, routes every workspace request -- CPU memory too -- to ccv_nnc_stream_compat_get_workspace, so
// the CPU scratch lives right behind `super` in both kinds; ccv_nnc_init_stream_context() grows GPU contexts by the rest.
struct stream_cpu_t {
	ccv_nnc_stream_context_s super;
	size_t workspace_size;
	void* workspace;
};
struct stream_gpu_t {
	ccv_nnc_stream_context_s super; // host-allocated base (lib/nnc/_ccv_nnc_stream.h:29-46)
	struct { size_t workspace_size; void* workspace; } cpu;
	device_local_t one;             // contexts created for a fixed device
	device_local_t* any;            // CCV_COMPUTE_DEVICE_ANY contexts: one slot per device, created at first use there
	int any_size;
};
struct signal_gpu_t {
	ccv_nnc_stream_signal_s super;
	hipEvent_t event;
	unsigned long capture_id; // the capture in which it was last emitted (0: outside any), and -- folded form -- the device whose recording stream it was emitted on (-1: none)
	int capture_dev;
};
// stream_context == NULL: the device's default stream + a per-thread, per-device workspace
// (lib/nnc/gpu/ccv_nnc_compat.cu:301-340).
constexpr int MAX_DEVICES = 64;
thread_local device_local_t tl_default[MAX_DEVICES];
thread_local struct { size_t workspace_size; void* workspace; } tl_default_cpu;
thread_local const char* tl_last_kernel = "";

struct mem_pressure_t { int device_id; nnc_mi355x_mem_pressure_f func; void* ctx; };
pthread_mutex_t g_mp_mutex = PTHREAD_MUTEX_INITIALIZER;
std::vector<mem_pressure_t>& g_mp = *new std::vector<mem_pressure_t>; // (never destroyed: a host thread the process does not join may still free memory while the exit handlers run; cmd_comm.cpp g_cliques)

void trigger_mem_pressure()
{
	int device_id = 0;
	HIP_ENFORCE(hipGetDevice(&device_id));
	pthread_mutex_lock(&g_mp_mutex);
	for (size_t i = 0; i < g_mp.size(); i++)
		if (g_mp[i].func && g_mp[i].device_id == device_id) g_mp[i].func(device_id, g_mp[i].ctx);
	pthread_mutex_unlock(&g_mp_mutex);
	HIP_ENFORCE(hipSetDevice(device_id));
}

int current_device()
{
	int d = 0;
	HIP_ENFORCE(hipGetDevice(&d));
	return d;
}

// -- stream capture (the section "HIP-graph capture of a compiled schedule" further down): what the rest of this file asks about it
std::atomic<unsigned long> g_pool_seq(0);          // device allocations so far
std::atomic<int> g_capture_active(0);              // captures in progress (process-wide: a data-parallel step's capture spans the devices)
std::atomic<unsigned long> g_capture_id(0);        // the running capture's number (never 0 while one runs)
std::atomic<unsigned long> g_graph_max_end_seq(0); // the newest allocation any LIVE captured graph may name (0: no graph alive)
hipStream_t g_cap_origin = 0;                      // the stream the running capture began on, its device,
int g_cap_device = 0;
constexpr int CAP_MAX_DEVICES = 64;
hipStream_t g_cap_rep[CAP_MAX_DEVICES];            // folded form: the ONE recording stream per device (the origin on its device; elsewhere the first stream of the device the step reached)
hipEvent_t g_cap_relay[CAP_MAX_DEVICES];           // ... and an event per device for dependencies between two non-origin devices, which are routed through the origin
int g_cap_keep_streams = -1;                       // and the capture's form: 0 = the step's streams of that device are folded into the recording one (default), 1 = kept (-1: environment not read yet)
inline bool pool_pinned(const unsigned long seq) { return g_capture_active.load(std::memory_order_acquire) > 0 || seq <= g_graph_max_end_seq.load(std::memory_order_acquire); }
// is THIS stream recording (a stream of the device that has not joined the capture is an ordinary stream)
inline bool stream_capturing(hipStream_t st)
{
	if (!st || g_capture_active.load(std::memory_order_acquire) <= 0) return false;
	hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
	HIP_ENFORCE(hipStreamIsCapturing(st, &status));
	return status == hipStreamCaptureStatusActive;
}

void release_device_block(void* ptr, bool drained); // memory from nnc_mi355x_malloc goes back the way it came (the caching layer further down)
void pool_stream_created(int device, hipStream_t st);   // ... whose frees are ordered against every stream the library makes: each one is registered
void pool_stream_destroyed(int device, hipStream_t st);

bool is_any(const ccv_nnc_stream_context_t* ctx) { return (ctx->type & CCV_COMPUTE_DEVICE_ANY) == CCV_COMPUTE_DEVICE_ANY; }

// The per-device state a command launched through `ctx` uses right now.  A CCV_COMPUTE_DEVICE_ANY context follows the
// calling thread's current device (lib/nnc/gpu/ccv_nnc_compat.cu:319-340); a fixed-device context makes its device
// current (the host's single scheduler thread relies on that when it walks a multi-GPU graph), but only when it is
// not already -- hipGetDevice is a thread-local read, hipSetDevice per launch would be host overhead.
device_local_t* bind(const ccv_nnc_stream_context_t* ctx)
{
	stream_gpu_t* s = (stream_gpu_t*)ctx;
	if (!is_any(ctx)) {
		if (current_device() != s->one.device) HIP_ENFORCE(hipSetDevice(s->one.device)); // one host thread may drive all 8 GPUs
		return &s->one;
	}
	const int device = current_device();
	if (device >= s->any_size) {
		s->any = (device_local_t*)realloc(s->any, sizeof(device_local_t) * (device + 1));
		memset(s->any + s->any_size, 0, sizeof(device_local_t) * (device + 1 - s->any_size));
		s->any_size = device + 1;
	}
	device_local_t* l = s->any + device;
	if (!l->stream) {
		l->device = device;
		HIP_ENFORCE(hipStreamCreateWithFlags(&l->stream, hipStreamDefault)); // blocking w.r.t. the NULL stream, see ccv_nnc_init_stream_context
		pool_stream_created(device, l->stream);
	}
	return l;
}

// The stream a context's work is enqueued on right now.  While a step is being recorded in the folded form (HIP-graph capture further down) every stream of
// the recording device that the step reaches works on the recording stream: issue order on one stream IS a valid order of the step.
inline hipStream_t effective_stream(device_local_t* const l)
{
	if (g_capture_active.load(std::memory_order_acquire) > 0 && l->capture_alias == g_capture_id.load(std::memory_order_acquire) && l->capture_alias && l->device >= 0 && l->device < CAP_MAX_DEVICES && g_cap_rep[l->device]) return g_cap_rep[l->device];
	return l->stream;
}

} // namespace

namespace nnc {
int g_force_tile = 0;
int g_force_splits = 0;
static const char* const g_tune_names[TUNE_COUNT] = { "WINO_SLICE_KB", "WINO_FUSED_MAX_C", "WINO_FUSED_GRID", "GRID_WG_PER_CU", "WINO_WGRAD_FUSED_MAX", "GEMM_BUFFER_LOADS", "CONV_NCHW_HALF_F16", "BN_SMALL_PLANES", "SDPA_MFMA", "BN_CLUSTER", "GEMM_VEC_EPILOGUE", "POOL_ROWS", "GEMM_HALF_CHUNK8", "LSTM_PERSISTENT", "LSTM_ROWS", "GEMM_BF16X3", "BN_CLUSTER_SLOTS", "GEMM_BATCH_XCD" };
// GRID_WG_PER_CU = 0: grid-stride kernels get one trip per thread.  tools/ew_bw_bench.py: a grid capped at 8 .. 64 workgroups per CU
// striding a 3.3 GB tensor runs at 4.7 - 5.2 TB/s, the same kernel with the whole tensor as its grid at 6.2 TB/s.
static long g_tune_values[TUNE_COUNT] = { 0, 128, 0, 0, 128, 1, 32, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 1 }; // defaults: the measured best on the MI355X
static int g_tune_env_read = 0;
long tune(int key)
{
	if (!g_tune_env_read) { // environment overrides, read once: NNC_MI355X_WINO_SLICE_KB=96 ...
		for (int i = 0; i < TUNE_COUNT; i++) {
			char name[64];
			snprintf(name, sizeof(name), "NNC_MI355X_%s", g_tune_names[i]);
			const char* v = getenv(name);
			if (v && *v) g_tune_values[i] = atol(v);
		}
		g_tune_env_read = 1;
	}
	return key >= 0 && key < TUNE_COUNT ? g_tune_values[key] : 0;
}
}
namespace nnc {

// Gradient all-reduces that went out on the communication stream (cmd_comm.cpp, "overlap"): whatever this stream does next comes behind them.
static inline device_local_t* joined(device_local_t* const l)
{
	if (g_comm_overlap_epoch.load(std::memory_order_acquire) != l->comm_seen) comm_overlap_join(l->stream, &l->comm_seen);
	return l;
}
std::atomic<unsigned long> g_launch_seq(0); // bumped by every stream_of(): "has anything been launched through this library since ...?" (peephole.cpp: SGD batches)
hipStream_t stream_of(const ccv_nnc_stream_context_t* ctx)
{
	g_launch_seq.fetch_add(1, std::memory_order_relaxed);
	if (g_comm_pending) comm_flush(); // a launch is about to be ordered on a stream: recorded collectives go first (cmd_comm.cpp)
	if (g_deferred_live) deferred_flush(ctx); // ... and so does this stream's recorded command (peephole.cpp)
	if (!ctx) return (hipStream_t)0;
	if (CCV_STREAM_GET_CONTEXT(ctx->type) != CCV_STREAM_CONTEXT_GPU) return (hipStream_t)0;
	return effective_stream(joined(bind(ctx)));
}
void stream_registered(const int device, hipStream_t st) { pool_stream_created(device, st); }
// the stream itself, no hooks: for recording "this buffer has been written" behind a command that has just been enqueued (cmd_comm.cpp comm_gradient_written)
hipStream_t stream_peek(const ccv_nnc_stream_context_t* ctx)
{
	if (!ctx || CCV_STREAM_GET_CONTEXT(ctx->type) != CCV_STREAM_CONTEXT_GPU) return (hipStream_t)0;
	return effective_stream(bind(ctx));
}

static thread_local size_t tl_ws_prefix = 0;
static thread_local size_t tl_ws_limit = 0; // inside a WorkspaceScope: the bytes reserved up front (prefix + inner), 0 = no scope

void* workspace_of(const ccv_nnc_stream_context_t* ctx, size_t size)
{
	// Inside a scope the buffer must not move: the scope's owner holds pointers into its prefix (staged layouts).  A request
	// beyond what the scope reserved is refused (the caller's CCV_NNC_EXEC_OOM path: the convolution falls back to an algorithm
	// whose scratch the scope did account for) instead of growing -- growing frees and re-allocates the buffer, and the kernels
	// launched afterwards would read the staged tensors out of freed memory (a GPU memory fault on the DawnNet's NCHW
	// convolutions: the first-layer gradient's per-wave partials were not in the scope's bound).
	if (tl_ws_limit && tl_ws_prefix + size > tl_ws_limit) return 0;
	char* p = (char*)ccv_nnc_stream_compat_get_workspace(ctx, tl_ws_prefix + size, CCV_TENSOR_GPU_MEMORY);
	return p ? p + tl_ws_prefix : 0;
}

WorkspaceScope::WorkspaceScope(const ccv_nnc_stream_context_t* ctx, size_t prefix_bytes, size_t inner_bytes)
{
	prefix_bytes = (prefix_bytes + 255) & ~(size_t)255;
	prev = tl_ws_prefix;
	prev_limit = tl_ws_limit;
	if (tl_ws_limit && tl_ws_prefix + prefix_bytes + inner_bytes > tl_ws_limit) { base = 0; return; } // nested scope beyond the outer reservation
	char* p = (char*)ccv_nnc_stream_compat_get_workspace(ctx, tl_ws_prefix + prefix_bytes + inner_bytes, CCV_TENSOR_GPU_MEMORY);
	base = p ? p + tl_ws_prefix : 0;
	if (!tl_ws_limit) tl_ws_limit = tl_ws_prefix + prefix_bytes + inner_bytes;
	tl_ws_prefix += prefix_bytes;
}
WorkspaceScope::~WorkspaceScope() { tl_ws_prefix = prev; tl_ws_limit = prev_limit; }

// ---- the hand-over words of kernels whose workgroups wait for each other inside ONE launch (cmd_norm.cpp's cluster kernels) -----------------------
// One area per stream (per device for ANY-device contexts, per thread and device for the NULL stream): launches on one stream run one after the other, so
// the area has ONE user at a time.  Layout: cluster_sync_t (a ticket counter and a done counter, both zero between launches -- the last workgroup of a
// launch puts them back) followed by 8-byte {tag, value} granules.  A launch's tags are its EPOCH, a per-area launch count that never repeats a value still
// in the area (32 bits, zero skipped), so the granules are never cleared: nothing but these kernels writes here, and what an older launch left cannot match.
// A polling loop that gives up (CLUSTER_SPIN_LIMIT polls: the workgroups it waits for never became resident) raises g_cluster_timeout in pinned host memory;
// the next synchronise of any stream stops the process with a message -- wrong numbers never travel on in silence.
static unsigned* g_cluster_timeout = 0;
static pthread_mutex_t g_cluster_mutex = PTHREAD_MUTEX_INITIALIZER;
static void cluster_check_timeout()
{
	if (g_cluster_timeout && *(volatile unsigned*)g_cluster_timeout) {
		fprintf(stderr, "[nnc_mi355x] a cluster kernel gave up waiting for its sibling workgroups (code %u): its results are invalid\n", *(volatile unsigned*)g_cluster_timeout);
		abort();
	}
}
// ---- one spinning launch at a time per device (ClusterTurn, common.h)
static struct { pthread_mutex_t mutex; hipEvent_t event; hipStream_t last; int have; } g_cluster_turn[MAX_DEVICES];
static pthread_once_t g_cluster_turn_once = PTHREAD_ONCE_INIT;
static long g_cluster_turns_chained = 0;
static void cluster_turn_init() { for (int i = 0; i < MAX_DEVICES; i++) pthread_mutex_init(&g_cluster_turn[i].mutex, 0); }
long cluster_turns_chained(void) { return __atomic_load_n(&g_cluster_turns_chained, __ATOMIC_RELAXED); }
ClusterTurn::ClusterTurn(const ccv_nnc_stream_context_t* ctx) : stream(stream_of(ctx)), device(current_device())
{
	pthread_once(&g_cluster_turn_once, cluster_turn_init);
	if (device < 0 || device >= MAX_DEVICES) { device = -1; return; }
	auto& t = g_cluster_turn[device];
	pthread_mutex_lock(&t.mutex); // held across the command's launches: the order of the turns is the order in which they were taken
	if (t.have && t.last != stream) {
		// the previous turn was taken on ANOTHER stream: an event behind everything queued there so far (its spinning launches included), and this stream
		// waits for it.  Nothing is recorded while one stream takes turn after turn -- the common case costs no runtime call at all.
		if (!t.event) HIP_ENFORCE(hipEventCreateWithFlags(&t.event, hipEventDisableTiming));
		HIP_ENFORCE(hipEventRecord(t.event, t.last));
		HIP_ENFORCE(hipStreamWaitEvent(stream, t.event, 0));
		__atomic_add_fetch(&g_cluster_turns_chained, 1, __ATOMIC_RELAXED);
	}
}
ClusterTurn::~ClusterTurn()
{
	if (device < 0) return;
	auto& t = g_cluster_turn[device];
	t.last = stream;
	t.have = 1;
	pthread_mutex_unlock(&t.mutex);
}
// a stream is about to be destroyed: it must not be recorded on later (device_rt.cpp's local_release calls this after draining it -- nothing of its turns is left to wait for)
static void cluster_turn_forget(const int device, hipStream_t stream)
{
	pthread_once(&g_cluster_turn_once, cluster_turn_init);
	if (device < 0 || device >= MAX_DEVICES) return;
	auto& t = g_cluster_turn[device];
	pthread_mutex_lock(&t.mutex);
	if (t.have && t.last == stream) t.have = 0;
	pthread_mutex_unlock(&t.mutex);
}

void* cluster_sync_of(const ccv_nnc_stream_context_t* ctx, size_t granule_bytes, unsigned* epoch, unsigned** timeout_word)
{
	if (granule_bytes > CLUSTER_SYNC_BYTES - 256) return 0;
	if (g_deferred_live) deferred_flush(ctx);
	device_local_t* l;
	if (ctx && CCV_STREAM_GET_CONTEXT(ctx->type) == CCV_STREAM_CONTEXT_GPU) l = bind(ctx);
	else {
		const int device = current_device();
		if (device >= MAX_DEVICES) return 0;
		l = &tl_default[device];
		l->device = device;
	}
	if (!g_cluster_timeout) {
		pthread_mutex_lock(&g_cluster_mutex);
		if (!g_cluster_timeout) {
			unsigned* w = 0;
			HIP_ENFORCE(hipHostMalloc((void**)&w, 256, hipHostMallocDefault));
			memset(w, 0, 256);
			g_cluster_timeout = w;
		}
		pthread_mutex_unlock(&g_cluster_mutex);
	}
	if (!l->cluster_sync) {
		void* p = nnc_mi355x_malloc(l->device, CLUSTER_SYNC_BYTES);
		if (!p) return 0;
		HIP_ENFORCE(hipMemset(p, 0, CLUSTER_SYNC_BYTES)); // (synchronous, once per stream and device)
		l->cluster_sync = p;
		l->cluster_epoch = 0;
	}
	// A captured launch carries its epoch as a kernel argument: every replay of the graph presents the SAME tags, and the granules the previous replay left would
	// match them.  The first cluster launch of a stream inside a capture is therefore preceded by a node that clears the area (2 MB, ~ a microsecond of HBM
	// time per replay and stream); behind it the recorded epochs are as fresh at every replay as they were when they were recorded.
	hipStream_t const es = l->stream ? effective_stream(l) : (hipStream_t)0;
	if (es && stream_capturing(es) && l->cluster_capture != g_capture_id.load(std::memory_order_acquire)) {
		HIP_ENFORCE(hipMemsetAsync(l->cluster_sync, 0, CLUSTER_SYNC_BYTES, es));
		l->cluster_capture = g_capture_id.load(std::memory_order_acquire);
	}
	if (++l->cluster_epoch == 0) l->cluster_epoch = 1;
	*epoch = l->cluster_epoch;
	*timeout_word = g_cluster_timeout;
	return l->cluster_sync;
}

const float* zero_page_of(const ccv_nnc_stream_context_t* ctx)
{
	static void* pages[MAX_DEVICES];
	static pthread_mutex_t mutex = PTHREAD_MUTEX_INITIALIZER;
	const int device = (ctx && CCV_STREAM_GET_CONTEXT(ctx->type) == CCV_STREAM_CONTEXT_GPU) ? bind(ctx)->device : current_device();
	assert(device >= 0 && device < MAX_DEVICES);
	if (!pages[device]) {
		pthread_mutex_lock(&mutex);
		if (!pages[device]) {
			const int prev = current_device();
			void* ptr = 0;
			HIP_ENFORCE(hipSetDevice(device));
			if (hipMalloc(&ptr, 256) != hipSuccess) { (void)hipGetLastError(); nnc_mi355x_pool_trim(device); HIP_ENFORCE(hipMalloc(&ptr, 256)); } // (the kept blocks are this process's to give back)
			HIP_ENFORCE(hipMemset(ptr, 0, 256));
			HIP_ENFORCE(hipDeviceSynchronize());
			HIP_ENFORCE(hipSetDevice(prev));
			pages[device] = ptr;
		}
		pthread_mutex_unlock(&mutex);
	}
	return (const float*)pages[device];
}

int device_cu_count(void)
{
	static int cus = 0;
	if (!cus) {
		hipDeviceProp_t prop;
		HIP_ENFORCE(hipGetDeviceProperties(&prop, current_device()));
		cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
	}
	return cus;
}

void note_kernel(const char* name) { tl_last_kernel = name; }

// ---- per-command marker ranges (roctx) ----
static int (*g_roctx_push)(const char*) = 0;
static int (*g_roctx_pop)(void) = 0;
volatile int g_markers_on = -1; // -1 = environment not read yet
void markers_enable(int on)
{
	if (on && !g_roctx_push) {
		void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
		if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
		if (h) {
			g_roctx_push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
			g_roctx_pop = (int (*)(void))dlsym(h, "roctxRangePop");
		}
	}
	g_markers_on = (on && g_roctx_push && g_roctx_pop) ? 1 : 0;
}
static inline int markers_on(void)
{
	if (g_markers_on < 0) { const char* e = getenv("NNC_MI355X_MARKERS"); markers_enable(e && *e && *e != '0'); }
	return g_markers_on;
}
// NNC_MI355X_SYNC_TRACE=1 (debugging aid): every command prints its row on entry and waits for the device on exit, so a GPU
// fault is reported right after the name of the command whose kernels raised it.
static int g_sync_trace = -1;
static inline int sync_trace_on(void)
{
	if (g_sync_trace < 0) { const char* e = getenv("NNC_MI355X_SYNC_TRACE"); g_sync_trace = (e && *e && *e != '0') ? 1 : 0; }
	return g_sync_trace;
}
static std::atomic<long> g_exec_commands(0); // nnc_mi355x_debug_exec_count(): commands that reached an exec function of this library (the host-enqueue measurement)
MarkerScope::MarkerScope(const uint32_t cmd) : active(0)
{
	g_exec_commands.fetch_add(1, std::memory_order_relaxed);
	if (sync_trace_on()) { fprintf(stderr, "[nnc_mi355x] > %s\n", command_row_name(cmd)); active |= 2; }
	if (!markers_on()) return;
	g_roctx_push(command_row_name(cmd));
	active |= 1;
}
MarkerScope::~MarkerScope()
{
	if (active & 1) g_roctx_pop();
	if (active & 2) { const hipError_t e = hipDeviceSynchronize(); fprintf(stderr, "[nnc_mi355x] < %s\n", e == hipSuccess ? "ok" : hipGetErrorString(e)); }
}

struct prof_rec_t { char name[192]; double flops, bytes; int dims[5]; hipEvent_t e0, e1; };
static pthread_mutex_t g_prof_mutex = PTHREAD_MUTEX_INITIALIZER;
static std::vector<prof_rec_t*>& g_prof = *new std::vector<prof_rec_t*>; // (never destroyed, as above)
static volatile int g_prof_on = 0;

// Algorithmic bytes of the NEXT contraction record of this thread (a convolution knows its tensors; the launcher underneath only sees an implicit matrix
// whose M x K counts every input pixel kh * kw times): taken by the next ProfScope that was given bytes < 0, cleared by it.
static thread_local double tl_prof_next_bytes = 0;
void prof_next_bytes(const double bytes) { if (g_prof_on) tl_prof_next_bytes = bytes; }
ProfScope::ProfScope(const char* name, double flops, double bytes, int M, int N, int K, int Z, int S, hipStream_t st) : rec(0), stream(st)
{
	if (!g_prof_on) return;
	if (bytes < 0) { bytes = tl_prof_next_bytes > 0 ? tl_prof_next_bytes : -bytes; tl_prof_next_bytes = 0; }
	prof_rec_t* r = new prof_rec_t;
	snprintf(r->name, sizeof(r->name), "%s", name);
	r->flops = flops; r->bytes = bytes;
	r->dims[0] = M; r->dims[1] = N; r->dims[2] = K; r->dims[3] = Z; r->dims[4] = S;
	HIP_ENFORCE(hipEventCreate(&r->e0));
	HIP_ENFORCE(hipEventCreate(&r->e1));
	HIP_ENFORCE(hipEventRecord(r->e0, st));
	rec = r;
}
ProfScope::~ProfScope()
{
	if (!rec) return;
	prof_rec_t* r = (prof_rec_t*)rec;
	HIP_ENFORCE(hipEventRecord(r->e1, stream));
	pthread_mutex_lock(&g_prof_mutex);
	g_prof.push_back(r);
	pthread_mutex_unlock(&g_prof_mutex);
}

} // namespace nnc

extern "C" {

// ---- device memory behind cumalloc / cufree: a caching layer over hipMalloc ----------------------------------------------------------------------
// The reference's dynamic graphs allocate and free tensors all the time; its own free lists (lib/nnc/ccv_nnc_xpu_alloc.c: size-keyed trees per stream and
// device) catch most of it and call cumalloc / cufree for the rest -- and a hipMalloc of 1 GB costs 33 - 42 ms on this box, a hipFree 0.2 - 0.8 ms
// (profiles/r05_v2_alloc_bench.txt).  Freed blocks are therefore KEPT, per device and exact (rounded) size, and handed out again:
//   free      what hipFree guarantees its caller -- nothing queued on the device can still touch the block -- is kept by draining the device
//             (hipDeviceSynchronize: the wait hipFree itself performs), then the block goes on its size's list instead of back to the driver;
//   allocate  the size is rounded (512 B below 1 MB, 2 MB above) and the newest block of that size is reused; otherwise hipMalloc.  Out of memory: every kept
//             block of the device goes back to the driver, the host's registered pressure callbacks run (curegmp: ccv_nnc_xpu_alloc's drain, the stream
//             contexts' workspaces), one retry.
//   bound     the bytes KEPT per device never exceed a cap -- half the device's memory, NNC_MI355X_POOL_KEEP_MB overrides -- the oldest kept blocks go back to
//             the driver first: a caller whose sizes never repeat (dynamic graphs over ragged batches) cannot grow the layer until allocations fail.
// NNC_MI355X_POOL_ALLOC=0 selects plain hipMalloc / hipFree.
// Round 5 first built this on the runtime's own stream-ordered pool (hipMallocAsync / hipFreeAsync on the legacy stream, release threshold = keep everything)
// and took it out again: with the free only QUEUED the full-size convolution parity sequence read back an output still holding its initial fill; with a device
// drain in front of every hipFreeAsync that sequence passed and the reference's own NCHW convolution int cases failed instead (cudnn.tests.c:130, :473) -- on
// ROCm 7.2 memory from the pool is not interchangeable with hipMalloc's for this library's mix of blocking copies, default-stream kernels and stream-ordered
// work, for reasons not understood.  The layer below involves no runtime feature beyond hipMalloc / hipFree / hipDeviceSynchronize.
static int g_pool_mode = -1;
// Round 6: the free is STREAM-ORDERED, no device drain.  What hipFree guarantees its caller -- nothing queued on the device can still touch the block -- is
// kept by a FENCE taken at the free: one event recorded on every stream of the device that still has work in flight (every stream this library ever makes
// is registered below; the legacy NULL stream counts as one), attached to the kept block.  A block is handed out again when its fence has completed; if
// only unfinished blocks of the size are kept, the allocation waits for the oldest one's events (the wait hipFree would have performed at the free, moved to
// the one place that needs it and almost never reached).  Idle streams are skipped (hipStreamQuery), so a run of frees behind a synchronize records nothing.
// (Why the runtime's own pool misbehaved in round 5: a hipFreeAsync queued on the legacy stream orders the block's reuse against THAT stream only -- and a block
// reused by an allocation whose first user is a blocking copy or another stream is not ordered against the kernels still queued elsewhere.  The fence below
// names every stream.)  Blocks go back to the driver through hipFree, which drains the device itself.
struct pool_fence_t { std::vector<hipEvent_t> events; int refs; bool done; };
struct kept_block_t { void* ptr; size_t size; pool_fence_t* fence; };
struct live_block_t { size_t size; unsigned long seq; };   // seq: the allocation's number, process-wide (the capture section below: which graphs may name the block)
struct parked_block_t { void* ptr; size_t size; unsigned long seq; };
typedef std::list<kept_block_t> kept_list_t;
struct pool_device_t {
	std::mutex mutex;                                                     // per device (one device's allocation no longer stalls the others)
	kept_list_t kept;                                                     // free blocks, oldest first
	std::unordered_map<size_t, std::deque<kept_list_t::iterator> > by_size; // the same blocks by rounded size, newest last
	std::unordered_map<void*, live_block_t> live;                         // blocks handed out (ptr -> rounded size, allocation number)
	std::vector<parked_block_t> parked;                                   // freed while a capture was running or a captured graph that may name them is alive: not reusable yet
	std::vector<hipStream_t> streams;                                     // the library's live streams on this device
	std::vector<hipEvent_t> spare;                                        // events of completed fences
	size_t kept_bytes, keep_cap;                                          // keep_cap 0 = not read yet
};
static pool_device_t* const g_pool = new pool_device_t[MAX_DEVICES](); // (never destroyed: cufree may arrive from a thread that outlives the exit handlers)
static std::atomic<long> g_pool_allocs(0), g_pool_retries(0), g_pool_kept_bytes(0), g_pool_live_bytes(0), g_pool_trimmed(0), g_pool_fence_events(0), g_pool_fence_waits(0), g_pool_parked_bytes(0);
static size_t pool_keep_cap(const int device)
{ // (caller holds the device's mutex; the device is current)
	pool_device_t& d = g_pool[device];
	if (!d.keep_cap) {
		const char* e = getenv("NNC_MI355X_POOL_KEEP_MB");
		size_t free_b = 0, total_b = 0;
		if (e && *e) d.keep_cap = ((size_t)strtoull(e, 0, 10) << 20) + 1; // (+ 1: "0 MB" is a cap too, not "unread")
		else if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b) d.keep_cap = total_b / 4; // a quarter of the device (half starved the process's other allocators)
		else { (void)hipGetLastError(); d.keep_cap = (size_t)32 << 30; }
	}
	return d.keep_cap;
}
static bool pool_on(void)
{
	if (g_pool_mode < 0) {
		const char* e = getenv("NNC_MI355X_POOL_ALLOC");
		g_pool_mode = (e && *e == '0') ? 0 : 1;
	}
	return g_pool_mode != 0;
}
static size_t pool_round(const size_t size)
{
	const size_t g = size < ((size_t)1 << 20) ? 512 : (size_t)2 << 20;
	return (size + g - 1) / g * g;
}
// -- fences (caller holds the device's mutex, the device is current)
static void fence_unref(pool_device_t& d, pool_fence_t* const f)
{
	if (!f || --f->refs > 0) return;
	for (hipEvent_t e : f->events) d.spare.push_back(e); // (an unfinished fence nobody refers to: its events are re-recorded when they are taken again)
	delete f;
}
static bool fence_done(pool_device_t& d, pool_fence_t* const f)
{
	if (!f || f->done) return true;
	while (!f->events.empty()) {
		const hipError_t r = hipEventQuery(f->events.back());
		if (r == hipErrorNotReady) { (void)hipGetLastError(); return false; }
		HIP_ENFORCE(r);
		d.spare.push_back(f->events.back());
		f->events.pop_back();
	}
	f->done = true;
	return true;
}
static void fence_wait(pool_device_t& d, pool_fence_t* const f)
{
	if (!f || f->done) return;
	for (hipEvent_t e : f->events) { HIP_ENFORCE(hipEventSynchronize(e)); d.spare.push_back(e); }
	f->events.clear();
	f->done = true;
	g_pool_fence_waits.fetch_add(1, std::memory_order_relaxed);
}
static pool_fence_t* fence_take(pool_device_t& d)
{ // everything queued on the device so far: an event behind every stream that is not idle.  0 = the device is idle.
	pool_fence_t* f = 0;
	auto behind = [&](hipStream_t st) {
		const hipError_t q = hipStreamQuery(st);
		if (q == hipSuccess) return;
		if (q != hipErrorNotReady) HIP_ENFORCE(q);
		(void)hipGetLastError();
		hipEvent_t e;
		if (!d.spare.empty()) { e = d.spare.back(); d.spare.pop_back(); }
		else HIP_ENFORCE(hipEventCreateWithFlags(&e, hipEventDisableTiming));
		HIP_ENFORCE(hipEventRecord(e, st));
		if (!f) { f = new pool_fence_t; f->refs = 0; f->done = false; }
		f->events.push_back(e);
		g_pool_fence_events.fetch_add(1, std::memory_order_relaxed);
	};
	for (hipStream_t st : d.streams) behind(st);
	behind((hipStream_t)0);
	return f;
}
static void pool_drop_kept(pool_device_t& d, kept_list_t::iterator it)
{ // (caller holds the mutex) off both indexes; the caller owns the block now
	std::deque<kept_list_t::iterator>& q = d.by_size[it->size];
	for (size_t i = 0; i < q.size(); i++) if (q[i] == it) { q.erase(q.begin() + (long)i); break; }
	if (q.empty()) d.by_size.erase(it->size);
	d.kept_bytes -= it->size;
	g_pool_kept_bytes.fetch_sub((long)it->size, std::memory_order_relaxed);
	d.kept.erase(it);
}
static void pool_release_all(const int device)
{ // (caller holds the mutex) every kept block of the device back to the driver (hipFree drains the device: the fences need not be waited for)
	pool_device_t& d = g_pool[device];
	for (kept_block_t& k : d.kept) { HIP_ENFORCE(hipFree(k.ptr)); fence_unref(d, k.fence); g_pool_kept_bytes.fetch_sub((long)k.size, std::memory_order_relaxed); }
	d.kept.clear();
	d.by_size.clear();
	d.kept_bytes = 0;
}
static void pool_trim(const int device)
{ // (caller holds the mutex) oldest first, until the kept bytes fit the cap
	pool_device_t& d = g_pool[device];
	if (g_capture_active.load(std::memory_order_acquire) > 0) return; // (hipFree drains the device: not while a stream captures -- the bound is enforced again by the next free after the capture)
	const size_t cap = pool_keep_cap(device);
	while (!d.kept.empty() && d.kept_bytes > cap) {
		const kept_block_t k = d.kept.front();
		pool_drop_kept(d, d.kept.begin());
		HIP_ENFORCE(hipFree(k.ptr));
		fence_unref(d, k.fence);
		g_pool_trimmed.fetch_add(1, std::memory_order_relaxed);
	}
}
namespace {
void pool_stream_created(const int device, hipStream_t st)
{
	if (device < 0 || device >= MAX_DEVICES || !st) return;
	std::lock_guard<std::mutex> lock(g_pool[device].mutex);
	g_pool[device].streams.push_back(st);
}
void pool_stream_destroyed(const int device, hipStream_t st)
{ // (the caller has synchronized the stream: fences that name it are complete as far as it is concerned)
	if (device < 0 || device >= MAX_DEVICES || !st) return;
	std::lock_guard<std::mutex> lock(g_pool[device].mutex);
	std::vector<hipStream_t>& v = g_pool[device].streams;
	for (size_t i = 0; i < v.size(); i++) if (v[i] == st) { v[i] = v.back(); v.pop_back(); break; }
}
}

void* nnc_mi355x_malloc(int device, size_t size)
{
	void* ptr = 0;
	HIP_ENFORCE(hipSetDevice(device));
	if (!pool_on() || device < 0 || device >= MAX_DEVICES) {
		if (hipMalloc(&ptr, size) != hipSuccess || !ptr) {
			(void)hipGetLastError();
			ptr = 0;
			trigger_mem_pressure(); // let the host drop caches (workspaces, xpu_alloc free lists), then retry once
			if (hipMalloc(&ptr, size) != hipSuccess) { (void)hipGetLastError(); ptr = 0; }
		}
		return ptr;
	}
	const size_t rounded = pool_round(size ? size : 1);
	pool_device_t& d = g_pool[device];
	std::unique_lock<std::mutex> lock(d.mutex);
	auto found = d.by_size.find(rounded);
	if (found != d.by_size.end() && !found->second.empty()) {
		std::deque<kept_list_t::iterator>& q = found->second;
		kept_list_t::iterator take = d.kept.end();
		for (size_t i = q.size(); i-- > 0;) if (fence_done(d, q[i]->fence)) { take = q[i]; break; } // the newest block nothing can touch any more
		if (take == d.kept.end()) { take = q.front(); fence_wait(d, take->fence); }                    // none: wait for the oldest one's last users
		ptr = take->ptr;
		fence_unref(d, take->fence);
		pool_drop_kept(d, take);
		g_pool_allocs.fetch_add(1, std::memory_order_relaxed);
	}
	if (!ptr) {
		lock.unlock(); // (the driver call runs without the lock: 33 - 42 ms for a gigabyte)
		if (hipMalloc(&ptr, rounded) != hipSuccess || !ptr) {
			(void)hipGetLastError();
			ptr = 0;
			lock.lock();
			pool_release_all(device);
			lock.unlock();
			trigger_mem_pressure(); // the host drops its caches: they come back through nnc_mi355x_free (and are kept -- so release once more before the retry)
			lock.lock();
			pool_release_all(device);
			lock.unlock();
			g_pool_retries.fetch_add(1, std::memory_order_relaxed);
			if (hipMalloc(&ptr, rounded) != hipSuccess) { (void)hipGetLastError(); ptr = 0; }
		}
		lock.lock();
	}
	if (ptr) { d.live[ptr] = live_block_t{ rounded, g_pool_seq.fetch_add(1, std::memory_order_acq_rel) + 1 }; g_pool_live_bytes.fetch_add((long)rounded, std::memory_order_relaxed); }
	return ptr;
}

// Every kept block of the device (device < 0: of every device) back to the driver: before another allocator of the process needs the memory -- RCCL's
// communicator buffers (cmd_comm.cpp), this library's own direct allocations when they fail -- and for hosts that want the bytes back.
void nnc_mi355x_pool_trim(const int device)
{
	if (!pool_on()) return;
	const int prev = current_device();
	for (int dev = (device < 0 ? 0 : device); dev < (device < 0 ? MAX_DEVICES : device + 1) && dev < MAX_DEVICES; dev++) {
		pool_device_t& d = g_pool[dev];
		std::lock_guard<std::mutex> lock(d.mutex);
		if (d.kept.empty()) continue;
		HIP_ENFORCE(hipSetDevice(dev));
		pool_release_all(dev);
	}
	HIP_ENFORCE(hipSetDevice(prev));
}

} // extern "C"
namespace {
// Memory that came from nnc_mi355x_malloc goes back the way it came; `drained`: the caller has already waited for everything that may use the block
void release_device_block(void* ptr, const bool drained)
{
	if (!ptr) return;
	const int device = current_device();
	if (pool_on() && device >= 0 && device < MAX_DEVICES) {
		pool_device_t& d = g_pool[device];
		std::unique_lock<std::mutex> lock(d.mutex);
		auto at = d.live.find(ptr);
		if (at != d.live.end()) {
			kept_block_t k = { ptr, at->second.size, 0 };
			const unsigned long seq = at->second.seq;
			d.live.erase(at);
			g_pool_live_bytes.fetch_sub((long)k.size, std::memory_order_relaxed);
			if (pool_pinned(seq)) {
				// A capture is running (its kernels have not executed: "drained" says nothing about them, and a capturing stream can be neither queried nor
				// recorded on from here), or a captured graph that may name this block is alive: every replay touches the block again.  It waits on the side
				// until nnc_mi355x_graph_free / the end of a failed capture looks at it again.
				d.parked.push_back(parked_block_t{ ptr, k.size, seq });
				g_pool_parked_bytes.fetch_add((long)k.size, std::memory_order_relaxed);
				return;
			}
			if (!drained && (k.fence = fence_take(d))) k.fence->refs++;
			d.kept.push_back(k);
			d.by_size[k.size].push_back(std::prev(d.kept.end()));
			d.kept_bytes += k.size;
			g_pool_kept_bytes.fetch_add((long)k.size, std::memory_order_relaxed);
			pool_trim(device);
			return;
		}
		// not handed out by the layer.  A block that is on the kept list is being freed TWICE: hipFree would take it from under the list, and the next
		// allocation of its size would hand out unmapped memory -- stop here, as hipFree's own error used to
		for (const kept_block_t& k : d.kept)
			if (k.ptr == ptr) { fprintf(stderr, "[nnc_mi355x] double free of device memory %p (device %d, %zu bytes)\n", ptr, device, k.size); abort(); }
		for (const parked_block_t& k : d.parked)
			if (k.ptr == ptr) { fprintf(stderr, "[nnc_mi355x] double free of device memory %p (device %d, %zu bytes, held for a captured graph)\n", ptr, device, k.size); abort(); }
		lock.unlock(); // allocated before the switch was read, or by the plain path: the driver's free
	}
	HIP_ENFORCE(hipFree(ptr));
}
}
extern "C" {

void nnc_mi355x_free(int device, void* ptr)
{
	nnc::comm_flush_if_pending(); // a recorded collective (or a recorded command and its trail) may still name this memory
	if (!ptr) return;
	HIP_ENFORCE(hipSetDevice(device));
	release_device_block(ptr, false);
}
// Test / measurement hooks: allocations served from kept blocks, allocations that needed the pressure path, bytes kept (free) and bytes handed out
void nnc_mi355x_debug_pool_counts(long* allocs, long* retries, long* reserved_bytes, long* used_bytes)
{
	if (allocs) *allocs = g_pool_allocs.load(std::memory_order_relaxed);
	if (retries) *retries = g_pool_retries.load(std::memory_order_relaxed);
	if (reserved_bytes) *reserved_bytes = g_pool_kept_bytes.load(std::memory_order_relaxed) + g_pool_live_bytes.load(std::memory_order_relaxed);
	if (used_bytes) *used_bytes = g_pool_live_bytes.load(std::memory_order_relaxed);
}

long nnc_mi355x_debug_pool_trimmed(void) { return g_pool_trimmed.load(std::memory_order_relaxed); }
// events recorded by frees (0 for a free that found every stream idle) and allocations that had to wait for a kept block's last users
void nnc_mi355x_debug_pool_fences(long* events, long* waits)
{
	if (events) *events = g_pool_fence_events.load(std::memory_order_relaxed);
	if (waits) *waits = g_pool_fence_waits.load(std::memory_order_relaxed);
}

void nnc_mi355x_set_device(int device)
{
	if (device >= 0) HIP_ENFORCE(hipSetDevice(device));
}

// (Round 5 measured a chunked, pinned, overlapped path for large PAGEABLE copies here -- what ccv_nnc_tensor_write / _read, lib/nnc/ccv_nnc_tensor_io.c:28-133,
// and DATA_TRANSFER of CPU tensors go through -- against the runtime's blocking copy: the runtime moves pageable memory at 52 - 57 GB/s either way on this box
// (PCIe Gen5 x16's practical rate), the hand-made ring reached 34 - 51.  profiles/r05_v1_memcpy_pageable.txt; tools/memcpy_bench.py.  Nothing to build.)
void nnc_mi355x_memcpy(void* dest, const int dest_type, const void* src, const int src_type, size_t n)
{
	nnc::comm_flush_if_pending();
	if (n == 0) return;
	const int sm = CCV_TENSOR_GET_MEMORY(src_type), dm = CCV_TENSOR_GET_MEMORY(dest_type);
	if (sm == CCV_TENSOR_CPU_MEMORY && dm == CCV_TENSOR_GPU_MEMORY) {
		HIP_ENFORCE(hipSetDevice(CCV_TENSOR_GET_DEVICE_ID(dest_type)));
		HIP_ENFORCE(hipMemcpy(dest, src, n, hipMemcpyHostToDevice));
	} else if (sm == CCV_TENSOR_GPU_MEMORY && dm == CCV_TENSOR_CPU_MEMORY) {
		HIP_ENFORCE(hipSetDevice(CCV_TENSOR_GET_DEVICE_ID(src_type)));
		HIP_ENFORCE(hipMemcpy(dest, src, n, hipMemcpyDeviceToHost));
	} else if (sm == CCV_TENSOR_CPU_MEMORY && dm == CCV_TENSOR_CPU_MEMORY) {
		memmove(dest, src, n);
	} else {
		const int da = CCV_TENSOR_GET_DEVICE_ID(src_type), db = CCV_TENSOR_GET_DEVICE_ID(dest_type);
		HIP_ENFORCE(hipSetDevice(db));
		if (da == db) HIP_ENFORCE(hipMemcpy(dest, src, n, hipMemcpyDeviceToDevice));
		else HIP_ENFORCE(hipMemcpyPeer(dest, db, src, da, n)); // xGMI peer copy
	}
	// a blocking copy is a point where the host OBSERVES device results (read-backs, checkpoints): a cluster kernel that gave up must stop the process
	// here, not at some later stream wait
	if (sm == CCV_TENSOR_GPU_MEMORY || dm == CCV_TENSOR_GPU_MEMORY) nnc::cluster_check_timeout();
}

void* nnc_mi355x_host_alloc(size_t size)
{
	void* ptr = 0;
	if (hipHostMalloc(&ptr, size, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return 0; }
	return ptr;
}
void nnc_mi355x_host_free(void* ptr) { HIP_ENFORCE(hipHostFree(ptr)); }
int nnc_mi355x_host_register(void* ptr, size_t size)
{
	if (hipHostRegister(ptr, size, hipHostRegisterDefault) == hipSuccess) return 1;
	(void)hipGetLastError();
	return 0;
}
void nnc_mi355x_host_unregister(void* ptr) { HIP_ENFORCE(hipHostUnregister(ptr)); }

int nnc_mi355x_register_mem_pressure(int device_id, nnc_mi355x_mem_pressure_f func, void* const context)
{
	pthread_mutex_lock(&g_mp_mutex);
	int slot = -1;
	for (size_t i = 0; i < g_mp.size(); i++)
		if (!g_mp[i].func) { slot = (int)i; break; }
	const mem_pressure_t mp = { device_id, func, context };
	if (slot < 0) { g_mp.push_back(mp); slot = (int)g_mp.size() - 1; }
	else g_mp[slot] = mp;
	pthread_mutex_unlock(&g_mp_mutex);
	return slot;
}
void nnc_mi355x_unregister_mem_pressure(const int id)
{
	pthread_mutex_lock(&g_mp_mutex);
	if (id >= 0 && id < (int)g_mp.size()) g_mp[id] = mem_pressure_t{ 0, 0, 0 };
	pthread_mutex_unlock(&g_mp_mutex);
}

// cusetprofiler in the reference (lib/nnc/gpu/ccv_nnc_compat.cu: cudaProfilerStart / Stop): here it switches the per-command
// marker ranges on.  Every command the backend executes is then bracketed by a roctx range named after its registry row
// ("CCV_NNC_CONVOLUTION_FORWARD/CCV_NNC_BACKEND_GPU_CUDNN"), so a `rocprofv3 --marker-trace --kernel-trace` timeline attributes
// kernels to graph nodes (tracing).  The roctx library is looked up at run time (librocprofiler-sdk-roctx,
// then libroctx64); without it the switch does nothing.  NNC_MI355X_MARKERS=1 in the environment switches it on from the start.
void nnc_mi355x_set_profiler(int state) { nnc::markers_enable(state); }

int nnc_mi355x_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
	return n;
}

ccv_nnc_stream_context_t* ccv_nnc_init_stream_context(ccv_nnc_stream_context_t* const stream_context)
{ // the host allocated base + CPU-workspace tail: grow it in place into our subclass (compat.cu:426-436)
	stream_gpu_t* s = (stream_gpu_t*)realloc(stream_context, sizeof(stream_gpu_t));
	memset(&s->cpu, 0, sizeof(stream_gpu_t) - offsetof(stream_gpu_t, cpu));
	if (!is_any(&s->super)) {
		s->one.device = CCV_STREAM_GET_DEVICE_ID(s->super.type);
		const int prev = current_device();
		HIP_ENFORCE(hipSetDevice(s->one.device));
		// A BLOCKING stream, as the backend being replaced creates them (cudaStreamCreate, ccv_nnc_compat.cu:434): commands the host
		// issues with stream_context == NULL go to the legacy NULL stream and must order against the graph's own streams in both
		// directions -- cnnp initialises parameters and copies them between models on the NULL stream, then runs the compiled
		// graph on its streams (a non-blocking stream let the first evaluate read uninitialised weights on the MI355X).
		HIP_ENFORCE(hipStreamCreateWithFlags(&s->one.stream, hipStreamDefault));
		pool_stream_created(s->one.device, s->one.stream);
		HIP_ENFORCE(hipSetDevice(prev));
	}
	return (ccv_nnc_stream_context_t*)s;
}

static void local_release(device_local_t* l)
{
	if (!l->stream && !l->workspace && !l->staging && !l->palette && !l->cluster_sync) return;
	nnc::comm_flush_if_pending();
	const int prev = current_device();
	HIP_ENFORCE(hipSetDevice(l->device));
	if (l->stream) { HIP_ENFORCE(hipStreamSynchronize(l->stream)); nnc::cluster_turn_forget(l->device, l->stream); }
	if (l->workspace) release_device_block(l->workspace, true);
	if (l->staging) release_device_block(l->staging, true);
	if (l->palette) release_device_block(l->palette, true);
	if (l->cluster_sync) release_device_block(l->cluster_sync, true);
	if (l->stream) { pool_stream_destroyed(l->device, l->stream); HIP_ENFORCE(hipStreamDestroy(l->stream)); }
	l->workspace = 0; l->workspace_size = 0; l->staging = 0; l->staging_size = 0; l->palette = 0; l->palette_size = 0; l->stream = 0; l->cluster_sync = 0;
	HIP_ENFORCE(hipSetDevice(prev));
}

void ccv_nnc_deinit_stream_context(ccv_nnc_stream_context_t* const stream_context)
{
	nnc::comm_release_context(stream_context);
	stream_gpu_t* s = (stream_gpu_t*)stream_context;
	if (s->cpu.workspace) free(s->cpu.workspace);
	s->cpu.workspace = 0; s->cpu.workspace_size = 0;
	local_release(&s->one);
	for (int i = 0; i < s->any_size; i++) local_release(s->any + i);
	free(s->any);
	s->any = 0; s->any_size = 0;
}

void ccv_nnc_synchronize_stream_context(const ccv_nnc_stream_context_t* const stream_context)
{
	nnc::comm_flush_if_pending();
	if (!stream_context) { HIP_ENFORCE(hipStreamSynchronize((hipStream_t)0)); nnc::cluster_check_timeout(); return; }
	HIP_ENFORCE(hipStreamSynchronize(nnc::joined(bind(stream_context))->stream));
	nnc::cluster_check_timeout();
}

void* ccv_nnc_stream_compat_get_workspace(const ccv_nnc_stream_context_t* const stream_context, const size_t workspace_size, const int mem)
{ // grow-only scratch, one per stream (and per device for ANY streams); commands on one stream are ordered so they may
  // share it (compat.cu:438-471).  NULL context = this thread's default context.
	if (workspace_size == 0) return 0;
	if (mem == CCV_TENSOR_CPU_MEMORY) {
		size_t* size = &tl_default_cpu.workspace_size;
		void** ws = &tl_default_cpu.workspace;
		if (stream_context) { stream_cpu_t* c = (stream_cpu_t*)stream_context; size = &c->workspace_size; ws = &c->workspace; }
		if (*size >= workspace_size) return *ws;
		free(*ws);
		*ws = 0; *size = 0;
		if (posix_memalign(ws, 64, workspace_size) != 0) { *ws = 0; return 0; }
		*size = workspace_size;
		return *ws;
	}
	if (mem != CCV_TENSOR_GPU_MEMORY) return 0;
	// A command recorded by the look-ahead (peephole.cpp) asks for its scratch when it is LAUNCHED, and may grow (= move) this buffer then.
	// Launch it before anybody is handed a pointer: no caller can hold scratch memory across a recorded command's launch.
	if (nnc::g_deferred_live) nnc::deferred_flush(stream_context);
	device_local_t* l;
	hipStream_t st = 0;
	if (stream_context && CCV_STREAM_GET_CONTEXT(stream_context->type) == CCV_STREAM_CONTEXT_GPU) { l = bind(stream_context); st = effective_stream(l); }
	else {
		const int device = current_device();
		if (device >= MAX_DEVICES) return 0;
		l = &tl_default[device];
		l->device = device;
	}
	if (l->workspace_size >= workspace_size && l->workspace) return l->workspace;
	if (l->workspace) {
		// (inside a capture nothing can be waited for and nothing needs to be: the free below sets the old buffer aside for as long as the graph lives)
		if (!stream_capturing(st)) { HIP_ENFORCE(hipStreamSynchronize(st)); nnc::cluster_check_timeout(); } // queued kernels may still read the old buffer
		release_device_block(l->workspace, true);
	}
	l->workspace = nnc_mi355x_malloc(st ? l->device : current_device(), workspace_size);
	l->workspace_size = l->workspace ? workspace_size : 0;
	return l->workspace;
}

static void local_drain(device_local_t* l, hipStream_t st)
{
	if (!l->workspace && !l->staging && !l->palette) return;
	HIP_ENFORCE(hipStreamSynchronize(st));
	nnc::cluster_check_timeout();
	if (l->workspace) release_device_block(l->workspace, true);
	if (l->staging) release_device_block(l->staging, true);
	if (l->palette) release_device_block(l->palette, true);
	l->workspace = 0; l->workspace_size = 0;
	l->staging = 0; l->staging_size = 0;
	l->palette = 0; l->palette_size = 0;
}

// The staging arena of the stream `stream_context` launches on (NULL = this thread's default context): grow-only like the
// workspace, separate from it.
void* nnc_staging_of(const ccv_nnc_stream_context_t* const stream_context, const size_t size)
{
	if (size == 0) return 0;
	if (nnc::g_deferred_live) nnc::deferred_flush(stream_context); // as for the workspace: a recorded command may stage (and grow the arena) at its launch
	device_local_t* l;
	hipStream_t st = 0;
	if (stream_context && CCV_STREAM_GET_CONTEXT(stream_context->type) == CCV_STREAM_CONTEXT_GPU) { l = bind(stream_context); st = effective_stream(l); }
	else {
		const int device = current_device();
		if (device >= MAX_DEVICES) return 0;
		l = &tl_default[device];
		l->device = device;
	}
	if (l->staging_size >= size && l->staging) return l->staging;
	if (l->staging) {
		if (!stream_capturing(st)) { HIP_ENFORCE(hipStreamSynchronize(st)); nnc::cluster_check_timeout(); } // queued conversions may still read the old arena
		release_device_block(l->staging, true);
	}
	l->staging = nnc_mi355x_malloc(st ? l->device : current_device(), size);
	l->staging_size = l->staging ? size : 0;
	return l->staging;
}

// The palette arena of the stream `stream_context` launches on: the dense images of a command's palettized inputs (palette.cpp).  Grow-only, apart from the
// workspace and the staging arena -- the command that reads the images may grow either of those while it is being enqueued.
void* nnc_palette_of(const ccv_nnc_stream_context_t* const stream_context, const size_t size)
{
	if (size == 0) return 0;
	if (nnc::g_deferred_live) nnc::deferred_flush(stream_context);
	device_local_t* l;
	hipStream_t st = 0;
	if (stream_context && CCV_STREAM_GET_CONTEXT(stream_context->type) == CCV_STREAM_CONTEXT_GPU) { l = bind(stream_context); st = effective_stream(l); }
	else {
		const int device = current_device();
		if (device >= MAX_DEVICES) return 0;
		l = &tl_default[device];
		l->device = device;
	}
	if (l->palette_size >= size && l->palette) return l->palette;
	if (l->palette) {
		if (!stream_capturing(st)) { HIP_ENFORCE(hipStreamSynchronize(st)); nnc::cluster_check_timeout(); } // a queued command may still read the old images
		release_device_block(l->palette, true);
	}
	l->palette = nnc_mi355x_malloc(st ? l->device : current_device(), size);
	l->palette_size = l->palette ? size : 0;
	return l->palette;
}

void ccv_nnc_stream_compat_drain(ccv_nnc_stream_context_t* const stream_context)
{
	nnc::comm_flush_if_pending(); // drop the scratch buffers (the host calls this under memory pressure, ccv_nnc_stream.c:70-86)
	if (!stream_context) {
		free(tl_default_cpu.workspace);
		tl_default_cpu.workspace = 0; tl_default_cpu.workspace_size = 0;
		const int device = current_device();
		if (device < MAX_DEVICES) local_drain(&tl_default[device], (hipStream_t)0);
		return;
	}
	stream_cpu_t* c = (stream_cpu_t*)stream_context;
	free(c->workspace);
	c->workspace = 0; c->workspace_size = 0;
	if (CCV_STREAM_GET_CONTEXT(stream_context->type) != CCV_STREAM_CONTEXT_GPU) return;
	stream_gpu_t* s = (stream_gpu_t*)stream_context;
	const int prev = current_device();
	if (s->one.workspace || s->one.staging || s->one.palette) { HIP_ENFORCE(hipSetDevice(s->one.device)); local_drain(&s->one, s->one.stream); }
	for (int i = 0; i < s->any_size; i++)
		if (s->any[i].workspace || s->any[i].staging || s->any[i].palette) { HIP_ENFORCE(hipSetDevice(s->any[i].device)); local_drain(s->any + i, s->any[i].stream); }
	HIP_ENFORCE(hipSetDevice(prev));
}

static void host_callback_trampoline(void* userdata)
{
	ccv_nnc_async_callback_t* async = (ccv_nnc_async_callback_t*)userdata;
	async->fn(async->callback_context);
	free(async);
}
struct async_trampoline_t { ccv_nnc_async_callback_f async_callback; ccv_nnc_async_callback_t* async; };
static void host_async_trampoline(void* userdata)
{ // HIP callback threads must not call back into HIP: hand over to the host's dispatcher (compat.cu:528-545)
	async_trampoline_t* t = (async_trampoline_t*)userdata;
	t->async_callback(t->async);
	free(t);
}
void ccv_nnc_stream_compat_add_callback(ccv_nnc_stream_context_t* const stream, const ccv_nnc_callback_f callback, const ccv_nnc_async_callback_f async_callback, void* const callback_context)
{
	nnc::comm_flush_if_pending();
	device_local_t* s = nnc::joined(bind(stream));
	ccv_nnc_async_callback_t* async = (ccv_nnc_async_callback_t*)malloc(sizeof(ccv_nnc_async_callback_t));
	async->fn = callback;
	async->callback_context = callback_context;
	if (async_callback) {
		async_trampoline_t* t = (async_trampoline_t*)malloc(sizeof(async_trampoline_t));
		t->async_callback = async_callback;
		t->async = async;
		HIP_ENFORCE(hipLaunchHostFunc(effective_stream(s), host_async_trampoline, t));
	} else
		HIP_ENFORCE(hipLaunchHostFunc(effective_stream(s), host_callback_trampoline, async));
}

ccv_nnc_stream_signal_t* ccv_nnc_init_stream_signal(ccv_nnc_stream_signal_t* const signal)
{
	signal_gpu_t* g = (signal_gpu_t*)realloc(signal, sizeof(signal_gpu_t));
	const int dev = CCV_STREAM_GET_DEVICE_ID(g->super.type);
	if ((g->super.type & CCV_COMPUTE_DEVICE_ANY) != CCV_COMPUTE_DEVICE_ANY) HIP_ENFORCE(hipSetDevice(dev));
	HIP_ENFORCE(hipEventCreateWithFlags(&g->event, hipEventDisableTiming));
	g->capture_id = 0;
	g->capture_dev = -1;
	return (ccv_nnc_stream_signal_t*)g;
}
void ccv_nnc_deinit_stream_signal(ccv_nnc_stream_signal_t* const signal)
{
	nnc::comm_flush_if_pending(); // (a recorded command's trail may still name this signal)
	signal_gpu_t* g = (signal_gpu_t*)signal;
	HIP_ENFORCE(hipEventDestroy(g->event));
}
} // extern "C"
namespace nnc {
void signal_emit_now(const ccv_nnc_stream_context_t* const stream, const ccv_nnc_stream_signal_t* const signal)
{
	device_local_t* const l = joined(bind(stream));
	signal_gpu_t* const g = (signal_gpu_t*)signal;
	hipStream_t const es = effective_stream(l);
	const bool recording = stream_capturing(es);
	g->capture_id = recording ? g_capture_id.load(std::memory_order_acquire) : 0;
	g->capture_dev = (recording && !g_cap_keep_streams && l->device >= 0 && l->device < CAP_MAX_DEVICES && g_cap_rep[l->device] == es) ? l->device : -1;
	HIP_ENFORCE(hipEventRecord(g->event, es));
}
// Folded form, the dependency "device `to`'s recording stream continues behind `event`, which was recorded on device `from`'s recording stream".  ROCm 7.2 files a
// NON-origin stream that waits for a captured event under the stream the event was recorded on and walks those lists recursively when the capture ends -- two
// non-origin streams that wait for each other make that walk endless (capture section below).  So a non-origin stream only ever waits for events recorded ON THE
// ORIGIN: a dependency between two other devices goes origin-waits-for-it, origin-records-a-relay, the target waits for the relay.  (The origin then carries a
// dependency it does not need: its later kernels start behind the signalled work of `from` -- the price of the detour, paid only by steps that span devices.)
static void capture_cross_device(const int from, const int to, hipEvent_t event)
{
	if (to == g_cap_device) { HIP_ENFORCE(hipStreamWaitEvent(g_cap_origin, event, 0)); return; }
	if (from == g_cap_device) { HIP_ENFORCE(hipStreamWaitEvent(g_cap_rep[to], event, 0)); return; }
	HIP_ENFORCE(hipStreamWaitEvent(g_cap_origin, event, 0));
	if (!g_cap_relay[to]) { // (an event is recorded on streams of the device it was made on: the origin's)
		const int prev = current_device();
		HIP_ENFORCE(hipSetDevice(g_cap_device));
		HIP_ENFORCE(hipEventCreateWithFlags(&g_cap_relay[to], hipEventDisableTiming));
		HIP_ENFORCE(hipSetDevice(prev));
	}
	HIP_ENFORCE(hipEventRecord(g_cap_relay[to], g_cap_origin));
	HIP_ENFORCE(hipStreamWaitEvent(g_cap_rep[to], g_cap_relay[to], 0));
}
void signal_wait_now(const ccv_nnc_stream_context_t* const stream, const ccv_nnc_stream_signal_t* const signal)
{
	device_local_t* const l = bind(stream);
	signal_gpu_t* const g = (signal_gpu_t*)signal;
	if (g_capture_active.load(std::memory_order_acquire) > 0 && !g_cap_keep_streams) {
		const unsigned long id = g_capture_id.load(std::memory_order_acquire);
		if (g->capture_id == id && g->capture_dev >= 0 && l->stream && l->device >= 0 && l->device < CAP_MAX_DEVICES) {
			// emitted inside the running capture: the waiting stream is part of the step and works on its device's recording stream from here on (the first
			// stream of a device the step reaches BECOMES that device's recording stream).  Behind a signal of the same device stream order is the dependency.
			const int d = l->device;
			if (!g_cap_rep[d]) g_cap_rep[d] = l->stream;
			l->capture_alias = id;
			if (g->capture_dev != d) capture_cross_device(g->capture_dev, d, g->event);
			return;
		}
		if (g->capture_id != id && l->capture_alias == id) {
			fprintf(stderr, "[nnc_mi355x] a stream of the step being recorded waits for a signal that was emitted outside the capture: the step cannot be replayed\n");
			abort();
		}
	}
	HIP_ENFORCE(hipStreamWaitEvent(l->stream, g->event, 0));
}
}
extern "C" {
// A signal is a point where stream order becomes visible to other streams.  Recorded collectives go first; a recorded command's look-ahead decides whether the
// operation must launch it now or can wait in its trail (peephole.cpp: the emit behind a CONVOLUTION_BACKWARD, the SGD streams' waits for it).
void ccv_nnc_stream_compat_emit_signal(const ccv_nnc_stream_context_t* const stream, const ccv_nnc_stream_signal_t* const signal)
{
	if (nnc::sync_trace_on()) fprintf(stderr, "[nnc_mi355x] > EMIT signal %p on stream %p\n", (const void*)signal, (const void*)stream);
	if (nnc::g_comm_pending) nnc::comm_flush();
	if (nnc::g_deferred_live && nnc::deferred_signal_op(1, stream, signal)) return;
	nnc::signal_emit_now(stream, signal);
}
void ccv_nnc_stream_compat_wait_signal(const ccv_nnc_stream_context_t* const stream, const ccv_nnc_stream_signal_t* const signal)
{
	if (nnc::sync_trace_on()) fprintf(stderr, "[nnc_mi355x] > WAIT signal %p on stream %p\n", (const void*)signal, (const void*)stream);
	if (nnc::g_comm_pending) nnc::comm_flush();
	if (nnc::g_deferred_live && nnc::deferred_signal_op(0, stream, signal)) return;
	nnc::signal_wait_now(stream, signal);
}
int ccv_nnc_stream_context_get_device(const ccv_nnc_stream_context_t* const stream_context)
{
	if (!stream_context || CCV_STREAM_GET_CONTEXT(stream_context->type) != CCV_STREAM_CONTEXT_GPU) return current_device();
	return bind(stream_context)->device;
}
void* nnc_mi355x_stream_context_get_stream(const ccv_nnc_stream_context_t* const stream_context)
{
	return (void*)nnc::stream_of(stream_context);
}

ccv_nnc_stream_context_t* nnc_mi355x_stream_context_new(const int type)
{
	ccv_nnc_stream_context_t* base = (ccv_nnc_stream_context_t*)calloc(1, sizeof(stream_cpu_t)); // base + CPU-workspace tail, like the host
	base->type = type;
	if (CCV_STREAM_GET_CONTEXT(type) == CCV_STREAM_CONTEXT_GPU) return ccv_nnc_init_stream_context(base);
	return base;
}
void nnc_mi355x_stream_context_free(ccv_nnc_stream_context_t* const stream_context)
{
	if (!stream_context) return;
	if (CCV_STREAM_GET_CONTEXT(stream_context->type) == CCV_STREAM_CONTEXT_GPU) ccv_nnc_deinit_stream_context(stream_context);
	else free(((stream_cpu_t*)stream_context)->workspace);
	free(stream_context);
}
void nnc_mi355x_stream_context_wait(const ccv_nnc_stream_context_t* const stream_context)
{
	ccv_nnc_synchronize_stream_context(stream_context);
}
ccv_nnc_stream_signal_t* nnc_mi355x_stream_signal_new(const int type)
{
	ccv_nnc_stream_signal_t* base = (ccv_nnc_stream_signal_t*)calloc(1, sizeof(ccv_nnc_stream_signal_s));
	base->type = type;
	return ccv_nnc_init_stream_signal(base);
}
void nnc_mi355x_stream_signal_free(ccv_nnc_stream_signal_t* const signal)
{
	if (!signal) return;
	ccv_nnc_deinit_stream_signal(signal);
	free(signal);
}

// ---- pinned staging ring (include/nnc_mi355x.h: the host side of the GPU data pipeline) ------------------------------------------------
// A slot goes FREE -> (host fills it) -> SUBMITTED -> ACQUIRED -> FREE.  The loader thread calls host / submit, the consumer thread acquire / release;
// the slot's state is the ONE word they share (an atomic), and a call out of that order is refused (returns 0) instead of waiting on an event
// nobody has recorded (acquire before submit) or overwriting a device buffer a kernel still reads (submit before release).  Every HIP call is made
// with the ring's device current, whatever the calling thread's device was.
namespace {
enum { RING_FREE = 0, RING_SUBMITTED = 1, RING_ACQUIRED = 2 };
struct staging_ring_t {
	int device, slots;
	size_t slot_bytes;
	hipStream_t copy_stream;
	void** host;
	void** dev;
	hipEvent_t* copied;   // recorded on the copy stream behind a slot's host-to-device copy
	hipEvent_t* consumed; // recorded on the consumer's stream behind its last kernel reading the slot's device buffer
	std::atomic<int>* state;        // RING_*
	std::atomic<int>* copy_pending; // a copy out of host[s] has been submitted and not yet waited for (loader side)
	std::atomic<int>* consumed_set; // consumed[s] has been recorded at least once
};
struct DeviceGuard { // the ring's device for the calls below, the caller's afterwards
	int prev;
	explicit DeviceGuard(const int device) : prev(current_device()) { if (prev != device) HIP_ENFORCE(hipSetDevice(device)); else prev = -1; }
	~DeviceGuard() { if (prev >= 0) HIP_ENFORCE(hipSetDevice(prev)); }
};
}
void* nnc_mi355x_staging_ring_new(int device, int slots, size_t slot_bytes)
{
	if (slots < 1 || slots > 64 || slot_bytes == 0) return 0;
	DeviceGuard guard(device);
	staging_ring_t* r = (staging_ring_t*)calloc(1, sizeof(staging_ring_t));
	r->device = device; r->slots = slots; r->slot_bytes = slot_bytes;
	r->host = (void**)calloc(slots, sizeof(void*)); r->dev = (void**)calloc(slots, sizeof(void*));
	r->copied = (hipEvent_t*)calloc(slots, sizeof(hipEvent_t)); r->consumed = (hipEvent_t*)calloc(slots, sizeof(hipEvent_t));
	r->state = new std::atomic<int>[slots]; r->copy_pending = new std::atomic<int>[slots]; r->consumed_set = new std::atomic<int>[slots];
	for (int i = 0; i < slots; i++) { r->state[i] = RING_FREE; r->copy_pending[i] = 0; r->consumed_set[i] = 0; }
	bool ok = hipStreamCreate(&r->copy_stream) == hipSuccess; // a blocking stream like every stream of this library: orders against the legacy NULL stream
	if (ok) pool_stream_created(device, r->copy_stream);
	for (int i = 0; ok && i < slots; i++) {
		ok = hipHostMalloc(&r->host[i], slot_bytes, hipHostMallocDefault) == hipSuccess && (r->dev[i] = nnc_mi355x_malloc(device, slot_bytes)) != 0
			&& hipEventCreateWithFlags(&r->copied[i], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&r->consumed[i], hipEventDisableTiming) == hipSuccess;
	}
	if (!ok) { (void)hipGetLastError(); nnc_mi355x_staging_ring_free(r); return 0; }
	return r;
}
void* nnc_mi355x_staging_ring_host(void* ring, int slot)
{
	staging_ring_t* r = (staging_ring_t*)ring;
	if (!r || slot < 0 || slot >= r->slots) return 0;
	if (r->copy_pending[slot].load(std::memory_order_acquire)) { // the pinned buffer is still being read by the copy
		DeviceGuard guard(r->device);
		HIP_ENFORCE(hipEventSynchronize(r->copied[slot]));
		r->copy_pending[slot].store(0, std::memory_order_release);
	}
	return r->host[slot];
}
void* nnc_mi355x_staging_ring_device(void* ring, int slot)
{
	staging_ring_t* r = (staging_ring_t*)ring;
	return (!r || slot < 0 || slot >= r->slots) ? 0 : r->dev[slot];
}
int nnc_mi355x_staging_ring_submit(void* ring, int slot, size_t bytes)
{
	staging_ring_t* r = (staging_ring_t*)ring;
	if (!r || slot < 0 || slot >= r->slots || bytes > r->slot_bytes) return 0;
	if (r->state[slot].load(std::memory_order_acquire) != RING_FREE) return 0; // the previous round of this slot has not been released: its device buffer may still be read
	nnc::comm_flush_if_pending(); // (a copy is an order-observing point for recorded commands, like nnc_mi355x_memcpy)
	DeviceGuard guard(r->device);
	if (r->consumed_set[slot].load(std::memory_order_acquire)) HIP_ENFORCE(hipStreamWaitEvent(r->copy_stream, r->consumed[slot], 0)); // the device buffer's last reader, on the device
	HIP_ENFORCE(hipMemcpyAsync(r->dev[slot], r->host[slot], bytes, hipMemcpyHostToDevice, r->copy_stream));
	HIP_ENFORCE(hipEventRecord(r->copied[slot], r->copy_stream));
	r->copy_pending[slot].store(1, std::memory_order_release);
	r->state[slot].store(RING_SUBMITTED, std::memory_order_release); // copied[slot] is recorded: an acquire may wait on it now
	return 1;
}
int nnc_mi355x_staging_ring_acquire(void* ring, int slot, ccv_nnc_stream_context_t* const consumer)
{
	staging_ring_t* r = (staging_ring_t*)ring;
	if (!r || slot < 0 || slot >= r->slots) return 0;
	int expect = RING_SUBMITTED;
	if (!r->state[slot].compare_exchange_strong(expect, RING_ACQUIRED, std::memory_order_acq_rel)) return 0; // nothing submitted: there is no copy to wait for
	hipStream_t st = nnc::stream_of(consumer); // (binds the consumer's own device; the wait may name an event of another device)
	HIP_ENFORCE(hipStreamWaitEvent(st, r->copied[slot], 0));
	return 1;
}
int nnc_mi355x_staging_ring_release(void* ring, int slot, ccv_nnc_stream_context_t* const consumer)
{
	staging_ring_t* r = (staging_ring_t*)ring;
	if (!r || slot < 0 || slot >= r->slots) return 0;
	if (r->state[slot].load(std::memory_order_acquire) != RING_ACQUIRED) return 0;
	hipStream_t st = nnc::stream_of(consumer);
	{
		DeviceGuard guard(r->device); // consumed[slot] belongs to the ring's device, and so must the stream recording it
		HIP_ENFORCE(hipEventRecord(r->consumed[slot], st));
	}
	r->consumed_set[slot].store(1, std::memory_order_release);
	r->state[slot].store(RING_FREE, std::memory_order_release);
	return 1;
}
void nnc_mi355x_staging_ring_free(void* ring)
{
	staging_ring_t* r = (staging_ring_t*)ring;
	if (!r) return;
	DeviceGuard guard(r->device);
	if (r->copy_stream) { (void)hipStreamSynchronize(r->copy_stream); pool_stream_destroyed(r->device, r->copy_stream); (void)hipStreamDestroy(r->copy_stream); }
	for (int i = 0; i < r->slots; i++) {
		if (r->host && r->host[i]) (void)hipHostFree(r->host[i]);
		if (r->dev && r->dev[i]) nnc_mi355x_free(r->device, r->dev[i]);
		if (r->copied && r->copied[i]) (void)hipEventDestroy(r->copied[i]);
		if (r->consumed && r->consumed[i]) (void)hipEventDestroy(r->consumed[i]);
	}
	free(r->host); free(r->dev); free(r->copied); free(r->consumed);
	delete[] r->state; delete[] r->copy_pending; delete[] r->consumed_set;
	free(r);
}

// ---- HIP-graph capture of a compiled schedule ("HIP-graph capture of the static schedule to erase per-node launch latency") ----------
// The reference's scheduler walks the compiled graph node by node on ONE host thread (lib/nnc/ccv_nnc_graph_run.c:581-675 _ccv_nnc_graph_exec_run_loop), for every
// device of a data-parallel model: a step of the CIFAR-10 network is ~250 commands and a millisecond of host time for 4.6 ms of GPU time -- one thread cannot feed
// eight devices.  A step is a fixed sequence of launches on fixed addresses once the graph is compiled and autotuned, so the host may record it once and replay it:
//     nnc_mi355x_capture_begin(stream);
//     ccv_cnnp_model_fit(model, ..., stream);          /* or ccv_nnc_graph_run(graph, ..., stream): enqueue-only, as always -- now into the capture */
//     void* step = nnc_mi355x_capture_end(stream);
//     for (...) nnc_mi355x_graph_launch(step, stream); /* one runtime call per step */
//     nnc_mi355x_graph_free(step);
// The schedule's other streams reach the capture through the signals the host emits and waits for (the run forks from and joins back into the caller's stream:
// ccv_nnc_graph_run.c:707-726, :819-839).  TWO FORMS.  Folded (default): a stream of the recording device that waits for a signal emitted inside the capture
// works ON THE RECORDING STREAM from then on (effective_stream) -- the host issues a wait only after the matching emit, so issue order on one stream is a valid
// order of the step; the graph is one chain.  Kept (NNC_MI355X_CAPTURE_STREAMS=1 / nnc_mi355x_capture_keep_streams): the streams join the capture as HIP
// streams do, the graph keeps the schedule's branches.  The kept form is NOT usable with the reference's schedules on ROCm 7.2: the graph's own stream 0 is not
// the recording stream, it and the side streams wait for each other's events, and hipStreamEndCapture's walk over its "parallel capture streams" then recurses
// without end (stack overflow inside the runtime: profiles/r06_v13_capture_streams_fault.txt; the emulator models the bookkeeping and refuses such a capture).
// Streams of OTHER devices always join as HIP streams.  The look-ahead's recorded commands launch into the capture (capture_end flushes them), and what this
// library keeps per launch on the host side is made replayable:
//   * the hand-over areas of the cluster kernels are cleared by a node in front of a stream's first cluster launch (cluster_sync_of), so recorded epochs stay fresh;
//   * DROPOUT / the LSTM's dropout take a word of pinned host memory that the graph's first node increments: the masks differ from replay to replay
//     (capture_tick_of; the host-side generator only runs while the step is recorded);
//   * device memory freed while the capture runs, or later while a graph that may name it is alive, is set aside instead of being reused (release_device_block);
//     a scratch buffer that grows inside the capture is replaced without the wait it needs outside one;
//   * the turn order of spinning launches is re-established in front of every graph launch (ClusterTurn).
// What cannot be recorded stops the process with the runtime's own message (HIP_ENFORCE): waiting for a capturing stream (ccv_nnc_stream_context_wait, a
// while / case-of sub-graph's co_stream_await), a blocking copy out of it.  The tensors a recorded step names -- parameters, activations, the inputs the host
// bound -- must keep their addresses while the graph lives: the host feeds new batches by copying INTO the bound input tensors, not by binding others.
// Capture after a warm-up step (the first step compiles, autotunes and allocates).  One capture at a time per process.
} // extern "C"
namespace {
struct graph_rec_t { hipGraph_t graph; hipGraphExec_t exec; unsigned long end_seq; size_t nodes; int device; hipStream_t last; long launches; };
std::mutex& g_graph_mutex = *new std::mutex;                               // (never destroyed, like the allocator's lists)
std::vector<graph_rec_t*>& g_graphs = *new std::vector<graph_rec_t*>;
struct { hipStream_t origin; int device; } g_cap = { 0, 0 };
unsigned* g_capture_tick = 0;                                              // pinned host memory, readable by every device: bumped once per replay
__global__ void capture_tick_kernel(unsigned* const tick) { if (threadIdx.x == 0 && blockIdx.x == 0) tick[0] = tick[0] + 1; }

// every block set aside that nothing pins any more goes back to the kept lists (the caller has made sure no graph that named them is running)
void pool_unpark_all(void)
{
	const int prev = current_device();
	for (int dev = 0; dev < MAX_DEVICES; dev++) {
		pool_device_t& d = g_pool[dev];
		std::unique_lock<std::mutex> lock(d.mutex);
		if (d.parked.empty()) continue;
		HIP_ENFORCE(hipSetDevice(dev));
		size_t keep = 0;
		for (size_t i = 0; i < d.parked.size(); i++) {
			const parked_block_t b = d.parked[i];
			if (pool_pinned(b.seq)) { d.parked[keep++] = b; continue; }
			kept_block_t k = { b.ptr, b.size, 0 };
			if ((k.fence = fence_take(d))) k.fence->refs++; // (ordinary work queued since then may not touch it -- but a fence costs nothing behind idle streams)
			d.kept.push_back(k);
			d.by_size[k.size].push_back(std::prev(d.kept.end()));
			d.kept_bytes += k.size;
			g_pool_kept_bytes.fetch_add((long)k.size, std::memory_order_relaxed);
			g_pool_parked_bytes.fetch_sub((long)k.size, std::memory_order_relaxed);
		}
		d.parked.resize(keep);
		pool_trim(dev);
	}
	HIP_ENFORCE(hipSetDevice(prev));
}
// spinning launches are one after the other per device (ClusterTurn): `st` is about to receive some -- behind the last turn taken elsewhere, and the last turn itself from now on
void cluster_turn_take(const int device, hipStream_t st)
{
	pthread_once(&nnc::g_cluster_turn_once, nnc::cluster_turn_init);
	if (device < 0 || device >= MAX_DEVICES) return;
	auto& t = nnc::g_cluster_turn[device];
	pthread_mutex_lock(&t.mutex);
	if (t.have && t.last != st) {
		if (!t.event) HIP_ENFORCE(hipEventCreateWithFlags(&t.event, hipEventDisableTiming));
		HIP_ENFORCE(hipEventRecord(t.event, t.last));
		HIP_ENFORCE(hipStreamWaitEvent(st, t.event, 0));
	}
	t.last = st;
	t.have = 1;
	pthread_mutex_unlock(&t.mutex);
}
}
namespace nnc {
const unsigned* capture_tick_of(hipStream_t st) { return stream_capturing(st) ? g_capture_tick : 0; }
}
extern "C" {

int nnc_mi355x_capture_begin(ccv_nnc_stream_context_t* const stream_context)
{
	if (!stream_context || CCV_STREAM_GET_CONTEXT(stream_context->type) != CCV_STREAM_CONTEXT_GPU) { fprintf(stderr, "[nnc_mi355x] capture_begin: a GPU stream context is required (the NULL stream cannot record)\n"); return -1; }
	if (!pool_on()) { fprintf(stderr, "[nnc_mi355x] capture_begin: NNC_MI355X_POOL_ALLOC=0 (plain hipFree drains the device: not possible inside a capture)\n"); return -1; }
	nnc::comm_flush_if_pending(); // recorded collectives and recorded commands of EVERY stream belong in front of the capture
	device_local_t* const l = nnc::joined(bind(stream_context));
	std::lock_guard<std::mutex> lock(g_graph_mutex);
	if (g_capture_active.load(std::memory_order_acquire) > 0) { fprintf(stderr, "[nnc_mi355x] capture_begin: a capture is already running\n"); return -1; }
	if (!g_capture_tick) {
		unsigned* w = 0;
		HIP_ENFORCE(hipHostMalloc((void**)&w, 256, hipHostMallocDefault));
		memset(w, 0, 256);
		g_capture_tick = w;
	}
	cluster_turn_take(l->device, l->stream); // (outside the capture: an event of another stream's queue may still be waited for here)
	if (g_cap_keep_streams < 0) { const char* e = getenv("NNC_MI355X_CAPTURE_STREAMS"); g_cap_keep_streams = (e && *e == '1') ? 1 : 0; }
	g_cap.origin = l->stream;
	g_cap.device = l->device;
	g_cap_origin = l->stream;
	g_cap_device = l->device;
	for (int i = 0; i < CAP_MAX_DEVICES; i++) g_cap_rep[i] = 0;
	if (l->device >= 0 && l->device < CAP_MAX_DEVICES) g_cap_rep[l->device] = l->stream;
	l->capture_alias = g_capture_id.fetch_add(1, std::memory_order_acq_rel) + 1;
	g_capture_active.store(1, std::memory_order_release);
	// relaxed mode: allocations, event queries and the like stay legal on this and every other thread while the stream records (a loader thread keeps working)
	const hipError_t r = hipStreamBeginCapture(l->stream, hipStreamCaptureModeRelaxed);
	if (r != hipSuccess) {
		(void)hipGetLastError();
		g_capture_active.store(0, std::memory_order_release);
		fprintf(stderr, "[nnc_mi355x] capture_begin: hipStreamBeginCapture: %s\n", hipGetErrorString(r));
		return -1;
	}
	hipLaunchKernelGGL(capture_tick_kernel, dim3(1), dim3(64), 0, l->stream, g_capture_tick); // the graph's first node: every other stream forks behind it
	return 0;
}

void* nnc_mi355x_capture_end(ccv_nnc_stream_context_t* const stream_context)
{
	if (!stream_context || g_capture_active.load(std::memory_order_acquire) <= 0) { fprintf(stderr, "[nnc_mi355x] capture_end: no capture is running\n"); return 0; }
	nnc::comm_flush_if_pending(); // what the look-ahead still holds is part of the step: it launches INTO the capture
	device_local_t* const l = bind(stream_context);
	std::unique_lock<std::mutex> lock(g_graph_mutex);
	if (l->stream != g_cap.origin) { fprintf(stderr, "[nnc_mi355x] capture_end: not the stream the capture began on\n"); return 0; }
	hipGraph_t graph = 0;
	const char* const trace = getenv("NNC_MI355X_CAPTURE_TRACE"); // fault hunting: progress lines, and the captured graph as a DOT file (NNC_MI355X_CAPTURE_TRACE=<path>)
	if (trace) fprintf(stderr, "[nnc_mi355x] capture_end: hipStreamEndCapture ...\n");
	const hipError_t r = hipStreamEndCapture(l->stream, &graph);
	if (trace) fprintf(stderr, "[nnc_mi355x] capture_end: ... %s, graph %p\n", hipGetErrorString(r), (void*)graph);
#ifndef NNC_HIP_EMULATOR
	if (trace && r == hipSuccess && graph) { const hipError_t rd = hipGraphDebugDotPrint(graph, trace, 0); fprintf(stderr, "[nnc_mi355x] capture_end: graph written to %s (%s)\n", trace, hipGetErrorString(rd)); }
#endif
	graph_rec_t* rec = 0;
	if (r == hipSuccess && graph) {
		rec = new graph_rec_t{ graph, 0, g_pool_seq.load(std::memory_order_acquire), 0, l->device, 0, 0 };
		const hipError_t ri = hipGraphInstantiate(&rec->exec, graph, 0, 0, 0);
		if (ri != hipSuccess) {
			(void)hipGetLastError();
			fprintf(stderr, "[nnc_mi355x] capture_end: hipGraphInstantiate: %s\n", hipGetErrorString(ri));
			(void)hipGraphDestroy(graph);
			delete rec;
			rec = 0;
		} else {
			(void)hipGraphGetNodes(graph, 0, &rec->nodes);
			g_graphs.push_back(rec);
			if (rec->end_seq > g_graph_max_end_seq.load(std::memory_order_acquire)) g_graph_max_end_seq.store(rec->end_seq, std::memory_order_release);
		}
	} else {
		(void)hipGetLastError();
		fprintf(stderr, "[nnc_mi355x] capture_end: hipStreamEndCapture: %s (a stream of the step was not joined back into the capturing one, or an operation invalidated the capture)\n", hipGetErrorString(r));
	}
	g_capture_active.store(0, std::memory_order_release);
	lock.unlock();
	if (!rec) pool_unpark_all(); // nothing was recorded that could name the blocks set aside meanwhile
	return rec;
}

int nnc_mi355x_graph_launch(void* const graph, ccv_nnc_stream_context_t* const stream_context)
{
	graph_rec_t* const rec = (graph_rec_t*)graph;
	if (!rec || !rec->exec || !stream_context) return -1;
	hipStream_t st = nnc::stream_of(stream_context); // (recorded commands of this stream go first, as in front of any launch)
	cluster_turn_take(rec->device, st);
	const hipError_t r = hipGraphLaunch(rec->exec, st);
	if (r != hipSuccess) { (void)hipGetLastError(); fprintf(stderr, "[nnc_mi355x] graph_launch: %s\n", hipGetErrorString(r)); return -1; }
	rec->last = st;
	rec->launches++;
	return 0;
}

// The form of the NEXT captures: 0 (default) the step's streams of the recording device are folded into the recording stream, 1 they are kept as they are.
void nnc_mi355x_capture_keep_streams(const int on) { g_cap_keep_streams = on ? 1 : 0; }
int nnc_mi355x_graph_node_count(void* const graph) { return graph ? (int)((graph_rec_t*)graph)->nodes : 0; }

void nnc_mi355x_graph_free(void* const graph)
{
	graph_rec_t* const rec = (graph_rec_t*)graph;
	if (!rec) return;
	const int prev = current_device();
	HIP_ENFORCE(hipSetDevice(rec->device));
	HIP_ENFORCE(hipDeviceSynchronize()); // its last replay may still be running (and on other devices' streams only behind this one's: the step joins back into its origin)
	nnc::cluster_check_timeout();
	HIP_ENFORCE(hipGraphExecDestroy(rec->exec));
	HIP_ENFORCE(hipGraphDestroy(rec->graph));
	{
		std::lock_guard<std::mutex> lock(g_graph_mutex);
		unsigned long mx = 0;
		for (size_t i = 0; i < g_graphs.size(); i++) if (g_graphs[i] == rec) { g_graphs[i] = g_graphs.back(); g_graphs.pop_back(); break; }
		for (graph_rec_t* g : g_graphs) if (g->end_seq > mx) mx = g->end_seq;
		g_graph_max_end_seq.store(mx, std::memory_order_release);
	}
	delete rec;
	HIP_ENFORCE(hipSetDevice(prev));
	pool_unpark_all();
}
// bytes set aside for captured graphs (test / measurement hook)
long nnc_mi355x_debug_pool_parked_bytes(void) { return g_pool_parked_bytes.load(std::memory_order_relaxed); }

void* nnc_mi355x_event_new(void)
{
	hipEvent_t e;
	HIP_ENFORCE(hipEventCreate(&e));
	return (void*)e;
}
void nnc_mi355x_event_record(void* event, const ccv_nnc_stream_context_t* const stream_context)
{
	HIP_ENFORCE(hipEventRecord((hipEvent_t)event, nnc::stream_of(stream_context)));
}
float nnc_mi355x_event_elapsed_ms(void* start, void* stop)
{
	float ms = 0;
	HIP_ENFORCE(hipEventSynchronize((hipEvent_t)stop));
	nnc::cluster_check_timeout();
	HIP_ENFORCE(hipEventElapsedTime(&ms, (hipEvent_t)start, (hipEvent_t)stop));
	return ms;
}
void nnc_mi355x_event_free(void* event) { HIP_ENFORCE(hipEventDestroy((hipEvent_t)event)); }
const char* nnc_mi355x_last_kernel_name(void) { return tl_last_kernel; }
long nnc_mi355x_debug_exec_count(void) { return nnc::g_exec_commands.load(std::memory_order_relaxed); }
void nnc_mi355x_debug_force_tile(int wm, int wn)
{
	const bool known = (wm == 2 && wn == 2) || (wm == 2 && wn == 1) || (wm == 1 && wn == 2) || (wm == 1 && wn == 1);
	nnc::g_force_tile = known ? (wm | wn << 8) : 0;
}
void nnc_mi355x_debug_force_splits(int splits) { nnc::g_force_splits = splits > 0 ? splits : 0; } /* measurement aid (tools/conv_half_bench.py): the K-slices of contractions that leave the choice to the launcher */

int nnc_mi355x_tune_set(const char* name, long value)
{
	(void)nnc::tune(0); // environment first, explicit calls override it
	for (int i = 0; i < nnc::TUNE_COUNT; i++)
		if (strcmp(name, nnc::g_tune_names[i]) == 0) { nnc::g_tune_values[i] = value; return 0; }
	return -1;
}
long nnc_mi355x_tune_get(const char* name)
{
	for (int i = 0; i < nnc::TUNE_COUNT; i++)
		if (strcmp(name, nnc::g_tune_names[i]) == 0) return nnc::tune(i);
	return -1;
}

void nnc_mi355x_profile_enable(int on)
{
	pthread_mutex_lock(&nnc::g_prof_mutex);
	if (on) {
		for (size_t i = 0; i < nnc::g_prof.size(); i++) { (void)hipEventDestroy(nnc::g_prof[i]->e0); (void)hipEventDestroy(nnc::g_prof[i]->e1); delete nnc::g_prof[i]; }
		nnc::g_prof.clear();
	}
	nnc::g_prof_on = on;
	pthread_mutex_unlock(&nnc::g_prof_mutex);
}
int nnc_mi355x_profile_count(void) { return (int)nnc::g_prof.size(); }
int nnc_mi355x_profile_get(int i, char* name, int name_len, double* flops, double* bytes, float* ms, int dims[5])
{
	if (i < 0 || i >= (int)nnc::g_prof.size()) return -1;
	nnc::prof_rec_t* r = nnc::g_prof[i];
	snprintf(name, name_len, "%s", r->name);
	*flops = r->flops; *bytes = r->bytes;
	for (int k = 0; k < 5; k++) dims[k] = r->dims[k];
	HIP_ENFORCE(hipEventSynchronize(r->e1));
	HIP_ENFORCE(hipEventElapsedTime(ms, r->e0, r->e1));
	return 0;
}
const char* nnc_mi355x_version(void) { return "nnc-mi355x 0.1 (gfx950)"; }

} // extern "C"
