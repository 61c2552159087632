// Device runtime behind the plugin surface: memory, streams, events, workspace, host callbacks.
// Replaces lib/nnc/gpu/ccv_nnc_compat.cu:101-655 of the reference (CUDA runtime + cuBLAS/cuDNN handle pools);
// here a stream context owns just a HIP stream and a grow-only workspace -- there are no vendor-library handles.
#include "common.h"
#include <pthread.h>
#include <vector>

namespace {

struct stream_gpu_t {
	ccv_nnc_stream_context_s super; // host-allocated base (lib/nnc/_ccv_nnc_stream.h:29-46)
	int device;                     // -1 until bound (CCV_COMPUTE_DEVICE_ANY binds at first use)
	hipStream_t stream;
	void* workspace;
	size_t workspace_size;
};
struct signal_gpu_t {
	ccv_nnc_stream_signal_s super;
	hipEvent_t event;
};
// stream_context == NULL: the device's default stream + a per-thread, per-device workspace
// (lib/nnc/gpu/ccv_nnc_compat.cu:301-340).
struct default_ctx_t { void* workspace; size_t workspace_size; };
constexpr int MAX_DEVICES = 64;
thread_local default_ctx_t tl_default[MAX_DEVICES];
thread_local const char* tl_last_kernel = "";

struct mem_pressure_t { int device_id; nnc_mi355x_mem_pressure_f func; void* ctx; };
pthread_mutex_t g_mp_mutex = PTHREAD_MUTEX_INITIALIZER;
std::vector<mem_pressure_t> g_mp;

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

stream_gpu_t* bind(const ccv_nnc_stream_context_t* ctx)
{
	stream_gpu_t* s = (stream_gpu_t*)ctx;
	if (s->device < 0) { // CCV_COMPUTE_DEVICE_ANY: bind to whatever device is current at first use
		s->device = current_device();
		HIP_ENFORCE(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
	}
	return s;
}

} // namespace

namespace nnc {

hipStream_t stream_of(const ccv_nnc_stream_context_t* ctx)
{
	if (!ctx) return (hipStream_t)0;
	if (CCV_STREAM_GET_CONTEXT(ctx->type) != CCV_STREAM_CONTEXT_GPU) return (hipStream_t)0;
	stream_gpu_t* s = bind(ctx);
	return s->stream;
}

void* workspace_of(const ccv_nnc_stream_context_t* ctx, size_t size)
{
	return ccv_nnc_stream_compat_get_workspace(ctx, size, CCV_TENSOR_GPU_MEMORY);
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
			HIP_ENFORCE(hipMalloc(&ptr, 256));
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

struct prof_rec_t { char name[192]; double flops, bytes; int dims[5]; hipEvent_t e0, e1; };
static pthread_mutex_t g_prof_mutex = PTHREAD_MUTEX_INITIALIZER;
static std::vector<prof_rec_t*> g_prof;
static volatile int g_prof_on = 0;

ProfScope::ProfScope(const char* name, double flops, double bytes, int M, int N, int K, int Z, int S, hipStream_t st) : rec(0), stream(st)
{
	if (!g_prof_on) return;
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

void* nnc_mi355x_malloc(int device, size_t size)
{
	void* ptr = 0;
	HIP_ENFORCE(hipSetDevice(device));
	if (hipMalloc(&ptr, size) != hipSuccess || !ptr) {
		(void)hipGetLastError();
		ptr = 0;
		trigger_mem_pressure(); // let the host drop caches (workspaces, xpu_alloc free lists), then retry once
		if (hipMalloc(&ptr, size) != hipSuccess) { (void)hipGetLastError(); ptr = 0; }
	}
	return ptr;
}

void nnc_mi355x_free(int device, void* ptr)
{
	HIP_ENFORCE(hipSetDevice(device));
	HIP_ENFORCE(hipFree(ptr));
}

void nnc_mi355x_set_device(int device)
{
	if (device >= 0) HIP_ENFORCE(hipSetDevice(device));
}

void nnc_mi355x_memcpy(void* dest, const int dest_type, const void* src, const int src_type, size_t n)
{
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

void nnc_mi355x_set_profiler(int state) { (void)state; /* rocprofv3 attaches externally; nothing to toggle in-process */ }

int nnc_mi355x_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
	return n;
}

ccv_nnc_stream_context_t* ccv_nnc_init_stream_context(ccv_nnc_stream_context_t* const stream_context)
{ // the host allocated only the base struct: grow it in place into our subclass (compat.cu:426-436)
	stream_gpu_t* s = (stream_gpu_t*)realloc(stream_context, sizeof(stream_gpu_t));
	s->workspace = 0;
	s->workspace_size = 0;
	s->stream = 0;
	const int dev = CCV_STREAM_GET_DEVICE_ID(s->super.type);
	if ((s->super.type & CCV_COMPUTE_DEVICE_ANY) == CCV_COMPUTE_DEVICE_ANY) s->device = -1;
	else {
		s->device = dev;
		HIP_ENFORCE(hipSetDevice(dev));
		HIP_ENFORCE(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
	}
	return (ccv_nnc_stream_context_t*)s;
}

void ccv_nnc_deinit_stream_context(ccv_nnc_stream_context_t* const stream_context)
{
	stream_gpu_t* s = (stream_gpu_t*)stream_context;
	if (s->device < 0) return;
	HIP_ENFORCE(hipSetDevice(s->device));
	if (s->workspace) HIP_ENFORCE(hipFree(s->workspace));
	s->workspace = 0;
	s->workspace_size = 0;
	if (s->stream) HIP_ENFORCE(hipStreamDestroy(s->stream));
	s->stream = 0;
}

void ccv_nnc_synchronize_stream_context(const ccv_nnc_stream_context_t* const stream_context)
{
	if (!stream_context) { HIP_ENFORCE(hipStreamSynchronize((hipStream_t)0)); return; }
	stream_gpu_t* s = bind(stream_context);
	HIP_ENFORCE(hipSetDevice(s->device));
	HIP_ENFORCE(hipStreamSynchronize(s->stream));
}

void* ccv_nnc_stream_compat_get_workspace(const ccv_nnc_stream_context_t* const stream_context, const size_t workspace_size, const int mem)
{ // grow-only scratch, one per stream; commands on one stream are ordered so they may share it (compat.cu:438-471)
	if (mem != CCV_TENSOR_GPU_MEMORY) return 0;
	void** ws;
	size_t* ws_size;
	int device;
	if (stream_context) {
		stream_gpu_t* s = bind(stream_context);
		ws = &s->workspace; ws_size = &s->workspace_size; device = s->device;
	} else {
		device = current_device();
		if (device >= MAX_DEVICES) return 0;
		ws = &tl_default[device].workspace; ws_size = &tl_default[device].workspace_size;
	}
	if (*ws_size >= workspace_size && *ws) return *ws;
	if (*ws) {
		// queued kernels may still read the old buffer
		HIP_ENFORCE(hipStreamSynchronize(stream_context ? ((stream_gpu_t*)stream_context)->stream : (hipStream_t)0));
		HIP_ENFORCE(hipFree(*ws));
	}
	*ws = nnc_mi355x_malloc(device, workspace_size);
	*ws_size = *ws ? workspace_size : 0;
	return *ws;
}

void ccv_nnc_stream_compat_drain(ccv_nnc_stream_context_t* const stream_context)
{
	void** ws;
	size_t* ws_size;
	hipStream_t st = 0;
	if (stream_context) {
		stream_gpu_t* s = (stream_gpu_t*)stream_context;
		if (s->device < 0) return;
		ws = &s->workspace; ws_size = &s->workspace_size; st = s->stream;
	} else {
		const int device = current_device();
		if (device >= MAX_DEVICES) return;
		ws = &tl_default[device].workspace; ws_size = &tl_default[device].workspace_size;
	}
	if (*ws) {
		HIP_ENFORCE(hipStreamSynchronize(st));
		HIP_ENFORCE(hipFree(*ws));
		*ws = 0;
		*ws_size = 0;
	}
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
	stream_gpu_t* s = bind(stream);
	ccv_nnc_async_callback_t* async = (ccv_nnc_async_callback_t*)malloc(sizeof(ccv_nnc_async_callback_t));
	async->fn = callback;
	async->callback_context = callback_context;
	HIP_ENFORCE(hipSetDevice(s->device));
	if (async_callback) {
		async_trampoline_t* t = (async_trampoline_t*)malloc(sizeof(async_trampoline_t));
		t->async_callback = async_callback;
		t->async = async;
		HIP_ENFORCE(hipLaunchHostFunc(s->stream, host_async_trampoline, t));
	} else
		HIP_ENFORCE(hipLaunchHostFunc(s->stream, host_callback_trampoline, async));
}

ccv_nnc_stream_signal_t* ccv_nnc_init_stream_signal(ccv_nnc_stream_signal_t* const signal)
{
	signal_gpu_t* g = (signal_gpu_t*)realloc(signal, sizeof(signal_gpu_t));
	const int dev = CCV_STREAM_GET_DEVICE_ID(g->super.type);
	if ((g->super.type & CCV_COMPUTE_DEVICE_ANY) != CCV_COMPUTE_DEVICE_ANY) HIP_ENFORCE(hipSetDevice(dev));
	HIP_ENFORCE(hipEventCreateWithFlags(&g->event, hipEventDisableTiming));
	return (ccv_nnc_stream_signal_t*)g;
}
void ccv_nnc_deinit_stream_signal(ccv_nnc_stream_signal_t* const signal)
{
	signal_gpu_t* g = (signal_gpu_t*)signal;
	HIP_ENFORCE(hipEventDestroy(g->event));
}
void ccv_nnc_stream_compat_emit_signal(const ccv_nnc_stream_context_t* const stream, const ccv_nnc_stream_signal_t* const signal)
{
	stream_gpu_t* s = bind(stream);
	HIP_ENFORCE(hipSetDevice(s->device));
	HIP_ENFORCE(hipEventRecord(((const signal_gpu_t*)signal)->event, s->stream));
}
void ccv_nnc_stream_compat_wait_signal(const ccv_nnc_stream_context_t* const stream, const ccv_nnc_stream_signal_t* const signal)
{
	stream_gpu_t* s = bind(stream);
	HIP_ENFORCE(hipSetDevice(s->device));
	HIP_ENFORCE(hipStreamWaitEvent(s->stream, ((const signal_gpu_t*)signal)->event, 0));
}
int ccv_nnc_stream_context_get_device(const ccv_nnc_stream_context_t* const stream_context)
{
	if (!stream_context) return current_device();
	return bind(stream_context)->device;
}
void* nnc_mi355x_stream_context_get_stream(const ccv_nnc_stream_context_t* const stream_context)
{
	return (void*)nnc::stream_of(stream_context);
}

ccv_nnc_stream_context_t* nnc_mi355x_stream_context_new(const int type)
{
	ccv_nnc_stream_context_t* base = (ccv_nnc_stream_context_t*)calloc(1, sizeof(ccv_nnc_stream_context_s));
	base->type = type;
	if (CCV_STREAM_GET_CONTEXT(type) == CCV_STREAM_CONTEXT_GPU) return ccv_nnc_init_stream_context(base);
	return base;
}
void nnc_mi355x_stream_context_free(ccv_nnc_stream_context_t* const stream_context)
{
	if (!stream_context) return;
	if (CCV_STREAM_GET_CONTEXT(stream_context->type) == CCV_STREAM_CONTEXT_GPU) ccv_nnc_deinit_stream_context(stream_context);
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
	HIP_ENFORCE(hipEventElapsedTime(&ms, (hipEvent_t)start, (hipEvent_t)stop));
	return ms;
}
void nnc_mi355x_event_free(void* event) { HIP_ENFORCE(hipEventDestroy((hipEvent_t)event)); }
const char* nnc_mi355x_last_kernel_name(void) { return tl_last_kernel; }

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
