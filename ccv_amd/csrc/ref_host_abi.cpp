// Glue for linking libnnc_mi355x.so under the UNMODIFIED reference host (lib/nnc/*.c built with its GPU configuration
// macros -- see INTEGRATION.md).  The host spells "a GPU backend exists" with legacy symbol names; none of them refer to
// vendor libraries here: every one forwards to the native entry point of include/nnc_mi355x.h.
//   * device/memory compat calls            lib/nnc/gpu/ccv_nnc_compat.h:23-36
//   * coroutine <-> stream rendezvous        lib/nnc/gpu/ccv_nnc_compat.cu:547-581, lib/nnc/co.h:12-46
//   * the registration rows of the host's generated table we do not implement (empty: exec stays 0)
//   * ccv_nnc_compat_depalettize (palette.cpp) and the classic ccv_convnet accelerator's symbols, which are outside the nnc hot path
#include "common.h"
#include <pthread.h>

extern "C" {

void* cumalloc(int device, size_t size) { return nnc_mi355x_malloc(device, size); }
void cufree(int device, void* ptr) { nnc_mi355x_free(device, ptr); }
void cudevice(int device) { nnc_mi355x_set_device(device); }
void cumemcpy(void* dest, const int dest_type, const void* src, const int src_type, size_t n) { nnc_mi355x_memcpy(dest, dest_type, src, src_type, n); }
void* cuhostalloc(size_t size) { return nnc_mi355x_host_alloc(size); }
void cuhostfree(void* ptr) { nnc_mi355x_host_free(ptr); }
int curegister(void* ptr, size_t size) { return nnc_mi355x_host_register(ptr, size); }
void cuunregister(void* ptr) { nnc_mi355x_host_unregister(ptr); }
int curegmp(int device_id, nnc_mi355x_mem_pressure_f func, void* const context) { return nnc_mi355x_register_mem_pressure(device_id, func, context); }
void cuunregmp(const int id) { nnc_mi355x_unregister_mem_pressure(id); }
void cusetprofiler(int state) { nnc_mi355x_set_profiler(state); }
int ccv_nnc_gpu_device_count(void) { return nnc_mi355x_device_count(); }

// ---- coroutine rendezvous -------------------------------------------------------------------------------------------
// ABI mirror of the head of the host's scheduler / task structs (lib/nnc/co.h:12-46); only these fields are touched.
typedef struct co_routine_s co_routine_t;
typedef struct {
	int active;
	int stream_await_count;
	co_routine_t* head;
	co_routine_t* tail;
	pthread_t thread;
	pthread_cond_t notify;
	pthread_cond_t wait;
	pthread_mutex_t mutex;
} co_scheduler_t;
struct co_routine_s {
	int line;
	int done;
	int root;
	int other_size;
	co_scheduler_t* scheduler;
	/* the host's remaining fields are not accessed here */
};
// Provided by the host (lib/nnc/co.c); weak so that the library also loads standalone (bench.py, our tests).
void _co_prepend_task(co_scheduler_t* const scheduler, co_routine_t* const task) __attribute__((weak));

static void co_resume_on_stream_done(void* userdata)
{
	co_routine_t* const task = (co_routine_t*)userdata;
	co_scheduler_t* const scheduler = task->scheduler;
	pthread_mutex_lock(&scheduler->mutex);
	_co_prepend_task(scheduler, task);
	--scheduler->stream_await_count;
	pthread_cond_signal(&scheduler->wait);
	pthread_mutex_unlock(&scheduler->mutex);
}

// Returns 1 when the stream has already drained (the task continues), otherwise parks the task: a host function queued
// on the HIP stream re-schedules it once everything enqueued so far has completed.
int co_stream_compat_await(co_routine_t* const self, ccv_nnc_stream_context_t* const stream)
{
	nnc::comm_flush_if_pending();
	hipStream_t st = nnc::stream_of(stream);
	const hipError_t q = hipStreamQuery(st);
	if (q == hipSuccess) return 1;
	if (q != hipErrorNotReady) HIP_ENFORCE(q);
	(void)hipGetLastError();
	if (!_co_prepend_task) { fprintf(stderr, "co_stream_compat_await: the reference host's coroutine scheduler is not linked in\n"); abort(); }
	co_scheduler_t* const scheduler = self->scheduler;
	pthread_mutex_lock(&scheduler->mutex);
	++scheduler->stream_await_count;
	HIP_ENFORCE(hipLaunchHostFunc(st, co_resume_on_stream_done, self));
	pthread_mutex_unlock(&scheduler->mutex);
	return 0;
}

// ---- outside the hot path -------------------------------------------------------------------------------------------
static void out_of_scope(const char* what)
{
	fprintf(stderr, "libnnc_mi355x: %s is outside the nnc hot path this backend replaces (see DESIGN.md, out of scope)\n", what);
	abort();
}
void ccv_nnc_compat_depalettize(const void* input, const int datatype, const size_t input_length, const int qbits, const int number_in_blocks, void* output, const size_t output_length, ccv_nnc_stream_context_t* const stream_context)
{ // lib/nnc/gpu/ccv_nnc_palettize.cu:321-469 (the host's ccv_nnc_depalettize of GPU memory, lib/nnc/ccv_nnc_palettize.c:958-966): palette.cpp
	const int ret = nnc_mi355x_depalettize(input, datatype, input_length, qbits, number_in_blocks, output, output_length, stream_context);
	if (ret != CCV_NNC_EXEC_SUCCESS) { fprintf(stderr, "libnnc_mi355x: ccv_nnc_compat_depalettize refused (%d): datatype 0x%x, %d bits, %d per block, %zu -> %zu\n", ret, datatype, qbits, number_in_blocks, input_length, output_length); abort(); } // (the reference asserts)
}
// classic ccv_convnet GPU accelerator (lib/cuda/cwc.h:11-14): only reached when a ccv_convnet_t was created with use_cwc_accel
void cwc_convnet_encode(void* convnet, void** a, void** b, int batch) { out_of_scope("cwc_convnet_encode"); }
void cwc_convnet_classify(void* convnet, void** a, int symmetric, void** ranks, int tops, int batch) { out_of_scope("cwc_convnet_classify"); }
void cwc_convnet_compact(void* convnet) {}

// ---- registration rows of the host table that this library leaves empty ---------------------------------------------
#define NNC_STUB_ROW(cmd, backend) void _register_command_##cmd##_backend_##backend(ccv_nnc_cmd_backend_registry_t* const registry) { (void)registry; }
#include "../../include/nnc_mi355x_registry_stubs.def"
#undef NNC_STUB_ROW

} // extern "C"
