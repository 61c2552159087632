// CCV_NNC_COMM_ALLREDUCE / BROADCAST / REDUCE over RCCL (xGMI).  There is no CPU oracle for these commands; the
// contract is the one of lib/nnc/cmd/comm/gpu/ccv_nnc_comm_gpu_nccl.cu:12-165 and test/int/nnc/nccl.tests.c:14-226:
//   allreduce : outputs[i] = sum_j inputs[j] on every device i (in place allowed)
//   broadcast : outputs[i] = inputs[0]
//   reduce    : outputs[0] = sum_j inputs[j]
// Two deployments share the same commands:
//   (a) single process, N devices (the reference's only mode): tensor i lives on device CCV_TENSOR_GET_DEVICE_ID(type);
//       one communicator clique per device count from ncclCommInitAll, cached for the process
//       (lib/nnc/gpu/ccv_nnc_compat.cu:1404-1445); all per-device calls are issued inside one group.
//   (b) one process per GPU (how bench.py scales, torch.distributed.run): after nnc_mi355x_comm_init_rank() each
//       process passes exactly ONE tensor per command and the collective spans the processes.
// xGMI is point-to-point (7 links x ~153 GB/s per GPU): large gradients are bandwidth-bound on RCCL's ring/tree over
// those links, the many tiny ones (conv biases: 256 B) are latency-bound -- ccv_amd/comm.py therefore packs parameter
// gradients into a few large flat buckets before calling COMM_ALLREDUCE instead of one call per tensor
// (the reference issues one collective per parameter tensor, ccv_nnc_symbolic_graph_parallel.c:545-575).
#include "common.h"
#include <rccl/rccl.h>
#include <pthread.h>
#include <dlfcn.h>
#include <vector>
#include <unordered_map>
#include <algorithm>

using namespace nnc;

#define RCCL_ENFORCE(expr) do { \
	const ncclResult_t _st = (expr); \
	if (_st != ncclSuccess) { fprintf(stderr, "[%s:%d]:RCCL - Error: %d (%s)\n", __FILE__, __LINE__, (int)_st, ncclGetErrorString(_st)); abort(); } \
} while (0)

namespace {

constexpr int MAX_CLIQUE = 64;
pthread_mutex_t g_comm_mutex = PTHREAD_MUTEX_INITIALIZER;
// Single-process cliques: one communicator set per (stream context, device count), as the reference keeps them
// (ccv_nnc_nccl_get_comm, lib/nnc/gpu/ccv_nnc_compat.cu:1415-1445: in the stream's resource container; a static set only for
// the NULL stream).  The host's scheduler may place the all-reduce nodes of a data-parallel graph on different stream
// contexts; through ONE communicator RCCL would serialise them and tie the streams together.  Released with the context.
struct clique_t { const void* ctx; int device_count; ncclComm_t comm[MAX_CLIQUE]; };
// (never destroyed: the reference host frees stream contexts from threads it does not join -- its async-callback thread, ccv_nnc_stream.c -- and such a thread may
// still be in comm_release_context while the main thread runs the exit handlers; a container with a destructor would be gone under it.  ThreadSanitizer found
// exactly that on the reference's partial-schedule cases.  The same for every container of this library that a late thread can reach: device_rt.cpp, peephole.cpp.)
std::vector<clique_t*>& g_cliques = *new std::vector<clique_t*>;
ncclComm_t g_rank_comm = 0; // deployment (b)
int g_rank = 0, g_world = 1;

clique_t* clique_of(const void* ctx, int device_count)
{ // g_comm_mutex held
	for (size_t i = 0; i < g_cliques.size(); i++)
		if (g_cliques[i]->ctx == ctx && g_cliques[i]->device_count == device_count) return g_cliques[i];
	clique_t* c = new clique_t;
	c->ctx = ctx; c->device_count = device_count;
	int devs[MAX_CLIQUE];
	for (int i = 0; i < device_count; i++) devs[i] = i;
	int cur = 0;
	HIP_ENFORCE(hipGetDevice(&cur));
	nnc_mi355x_pool_trim(-1); // the communicators' buffers come from the driver: the blocks this library keeps for reuse go back first
	RCCL_ENFORCE(ncclCommInitAll(c->comm, device_count, devs));
	HIP_ENFORCE(hipSetDevice(cur));
	g_cliques.push_back(c);
	return c;
}

// The stream of `device` that belongs to the same schedule as `ctx`.  With the reference host linked in, that is
// ccv_nnc_stream_context_find_neighbor (lib/nnc/ccv_nnc_stream.c:388); standalone, the context's own stream when it is
// on that device, else the device's default stream.
hipStream_t neighbor_stream(ccv_nnc_stream_context_t* ctx, int device)
{
	if (!ctx) return (hipStream_t)0;
	typedef ccv_nnc_stream_context_t* (*find_f)(ccv_nnc_stream_context_t* const, const int);
	static find_f find = (find_f)dlsym(RTLD_DEFAULT, "ccv_nnc_stream_context_find_neighbor");
	if (find) {
		ccv_nnc_stream_context_t* n = find(ctx, device);
		return stream_of(n);
	}
	return ccv_nnc_stream_context_get_device(ctx) == device ? stream_of(ctx) : (hipStream_t)0;
}

// Coalescing.  The reference host issues ONE collective per parameter tensor (ccv_nnc_symbolic_graph_parallel.c:545-575): for
// VGG-D 32 of them from 256 bytes up, each its own RCCL launch on every device.  xGMI is point-to-point, the small ones are
// pure launch latency.  A COMM command therefore only RECORDS its collectives; they are issued -- all recorded ones inside
// one ncclGroupStart / End, which RCCL aggregates into one launch per device -- the moment anything else could observe
// the order: the next non-COMM launch of this library on any stream (stream_of), a synchronise, a signal, a host callback,
// or MAX_PENDING records.  Back-to-back COMM nodes of a schedule (a layer's weight and bias, the tail of backward) thus
// travel together; results and stream order are exactly those of immediate issue.
struct pending_t { int op; const void* in; void* out; size_t count; ncclDataType_t dt; int root; ncclComm_t comm; hipStream_t stream; int device; };
// (round 4: 4 096 records -- ResNet-50's ~200 gradient tensors x 8 devices are 1 600 records when the host issues them back to back; with 256 the
// queue forced seven mid-stream group launches per step, each under the process-wide mutex)
constexpr int MAX_PENDING = 4096;
pending_t g_pending[MAX_PENDING];
int g_pending_n = 0;
long g_stat_collectives = 0, g_stat_groups = 0;
thread_local int tl_in_comm = 0; // stream_of() calls made while recording / flushing must not recurse into the flush

// ---- Overlap (round 6; deployment (b) only, opt-in).  The reference's own data parallelism puts an all-reduce node behind every gradient in ONE graph and its
// scheduler runs them on other streams while backward continues (lib/nnc/ccv_nnc_symbolic_graph_parallel.c:545-575).  One process per GPU drives the
// unmodified model API: ccv_cnnp_model_backward enqueues the whole backward pass, THEN the host issues one COMM_ALLREDUCE per parameter on the same stream --
// issued immediately they would all run behind the last backward kernel.  With the mode on:
//   * CONVOLUTION_ / GEMM_ / BATCH_NORM_BACKWARD record an event behind the kernels that wrote each weight / bias gradient (comm_gradient_written);
//   * a flush sorts the recorded all-reduces by the order their gradients were WRITTEN (backward produces the last layer's first, the host lists the first
//     layer's first), cuts them into buckets of ~NNC_MI355X_COMM_BUCKET_MB (default 32; xGMI is point-to-point: few large launches), and issues each bucket as
//     one group on the COMMUNICATION stream behind its own gradients' events only -- the GPU is still in the forward pass at that moment, so every bucket
//     starts the moment its last gradient lands and the rest of backward runs beside it;
//   * every other stream joins the communication stream at its next order-observing point (device_rt.cpp joined(): launch, synchronise, signal, callback).
// Every rank runs the same program, so every rank sorts and cuts alike: the collectives meet in the same order.  A gradient with no record (another command
// wrote it: accumulation, a row without the hook) waits for the tail of the stream its all-reduce was issued on -- the order of immediate issue.
struct ready_ev_t { hipEvent_t ev; int refs; }; // one event per backward COMMAND, shared by the gradients it wrote (weight + bias: one record, not two)
struct ready_t { ready_ev_t* e; unsigned long seq; };
std::unordered_map<const void*, ready_t>& g_ready = *new std::unordered_map<const void*, ready_t>;
std::vector<ready_ev_t*>& g_ready_pool = *new std::vector<ready_ev_t*>;
unsigned long g_ready_seq = 0;
int g_overlap_set = -1; // nnc_mi355x_comm_overlap(): -1 = the environment decides
hipStream_t g_overlap_stream = 0;
hipEvent_t g_overlap_done = 0;
long g_stat_buckets = 0, g_stat_overlapped = 0;
bool overlap_wanted()
{
	static const int env = (getenv("NNC_MI355X_COMM_OVERLAP") && *getenv("NNC_MI355X_COMM_OVERLAP") == '1') ? 1 : 0;
	return (g_overlap_set >= 0 ? g_overlap_set : env) != 0;
}
ready_ev_t* ready_event()
{ // g_comm_mutex held; refs = 0
	ready_ev_t* e;
	if (!g_ready_pool.empty()) { e = g_ready_pool.back(); g_ready_pool.pop_back(); }
	else { e = new ready_ev_t; HIP_ENFORCE(hipEventCreateWithFlags(&e->ev, hipEventDisableTiming)); }
	e->refs = 0;
	return e;
}
void ready_unref(ready_ev_t* const e) { if (e && --e->refs <= 0) g_ready_pool.push_back(e); }
size_t bucket_bytes()
{
	static size_t b = 0;
	if (!b) { const char* const e = getenv("NNC_MI355X_COMM_BUCKET_MB"); const double mb = e ? atof(e) : 32.0; b = mb > 0 ? (size_t)(mb * 1048576.0) + 1 : 1; }
	return b;
}
bool overlapped_flush_locked()
{ // g_comm_mutex held, g_pending_n > 0.  false: not this form (anything but in-process-rank all-reduces is pending) -- the caller issues everything in place
	if (!g_rank_comm || !overlap_wanted()) return false;
	for (int i = 0; i < g_pending_n; i++) if (g_pending[i].op != 0 || g_pending[i].comm != g_rank_comm) return false;
	if (!g_overlap_stream) { // blocking, like every stream of this library: the legacy stream orders against it; known to the allocator's fences (device_rt.cpp)
		HIP_ENFORCE(hipStreamCreateWithFlags(&g_overlap_stream, hipStreamDefault));
		int dev = 0;
		HIP_ENFORCE(hipGetDevice(&dev));
		stream_registered(dev, g_overlap_stream);
	}
	struct item_t { int i; unsigned long seq; ready_ev_t* e; };
	std::vector<item_t> items((size_t)g_pending_n);
	for (int i = 0; i < g_pending_n; i++) {
		const pending_t& p = g_pending[i];
		auto at = g_ready.find(p.in);
		if (at != g_ready.end()) { items[i] = item_t{ i, at->second.seq, at->second.e }; g_ready.erase(at); } // (the map's reference becomes the item's)
		else { // unknown writer: the tail of the issuing stream, now
			ready_ev_t* const e = ready_event();
			e->refs = 1;
			HIP_ENFORCE(hipEventRecord(e->ev, p.stream));
			items[i] = item_t{ i, ++g_ready_seq, e };
		}
	}
	std::stable_sort(items.begin(), items.end(), [](const item_t& a, const item_t& b) { return a.seq < b.seq; });
	size_t at = 0;
	while (at < items.size()) {
		size_t end = at, bytes = 0;
		while (end < items.size() && (end == at || bytes < bucket_bytes())) { const pending_t& p = g_pending[items[end].i]; bytes += p.count * (p.dt == ncclHalf ? 2 : 4); end++; }
		for (size_t k = at; k < end; k++) if (k == at || items[k].e != items[k - 1].e) HIP_ENFORCE(hipStreamWaitEvent(g_overlap_stream, items[k].e->ev, 0));
		RCCL_ENFORCE(ncclGroupStart());
		for (size_t k = at; k < end; k++) { const pending_t& p = g_pending[items[k].i]; RCCL_ENFORCE(ncclAllReduce(p.in, p.out, p.count, p.dt, ncclSum, p.comm, g_overlap_stream)); }
		RCCL_ENFORCE(ncclGroupEnd());
		g_stat_groups++; g_stat_buckets++;
		at = end;
	}
	for (const item_t& it : items) ready_unref(it.e); // (the waits have been enqueued: an event nobody else refers to may be re-recorded)
	if (!g_overlap_done) HIP_ENFORCE(hipEventCreateWithFlags(&g_overlap_done, hipEventDisableTiming));
	HIP_ENFORCE(hipEventRecord(g_overlap_done, g_overlap_stream));
	g_stat_collectives += g_pending_n; g_stat_overlapped += g_pending_n;
	g_pending_n = 0;
	nnc::g_comm_pending = 0;
	nnc::g_comm_overlap_epoch.fetch_add(1, std::memory_order_release);
	return true;
}

void flush_locked()
{
	if (!g_pending_n) return;
	tl_in_comm++;
	if (overlapped_flush_locked()) { tl_in_comm--; return; }
	int cur = 0;
	HIP_ENFORCE(hipGetDevice(&cur));
	RCCL_ENFORCE(ncclGroupStart());
	for (int i = 0; i < g_pending_n; i++) {
		const pending_t& p = g_pending[i];
		if (p.op == 0) RCCL_ENFORCE(ncclAllReduce(p.in, p.out, p.count, p.dt, ncclSum, p.comm, p.stream));
		else if (p.op == 1) RCCL_ENFORCE(ncclBroadcast(p.in, p.out, p.count, p.dt, p.root, p.comm, p.stream));
		else RCCL_ENFORCE(ncclReduce(p.in, p.out, p.count, p.dt, ncclSum, p.root, p.comm, p.stream));
	}
	RCCL_ENFORCE(ncclGroupEnd());
	HIP_ENFORCE(hipSetDevice(cur));
	g_stat_collectives += g_pending_n; g_stat_groups++;
	g_pending_n = 0;
	nnc::g_comm_pending = 0;
	tl_in_comm--;
}

int max_pending()
{ // NNC_MI355X_COMM_MAX_PENDING: a smaller queue (tests force the mid-stream group launches a full queue causes)
	static int cap = 0;
	if (!cap) {
		const char* const e = getenv("NNC_MI355X_COMM_MAX_PENDING");
		const int v = e ? atoi(e) : MAX_PENDING;
		cap = v >= 1 && v <= MAX_PENDING ? v : MAX_PENDING;
	}
	return cap;
}

void record(const int op, const void* in, void* out, size_t count, ncclDataType_t dt, int root, ncclComm_t comm, hipStream_t stream, int device)
{ // g_comm_mutex held
	if (g_pending_n >= max_pending()) flush_locked();
	pending_t& p = g_pending[g_pending_n++];
	p.op = op; p.in = in; p.out = out; p.count = count; p.dt = dt; p.root = root; p.comm = comm; p.stream = stream; p.device = device;
	nnc::g_comm_pending = 1;
}

// The element types the reference's rows register (comm_gpu_nccl.cu:65,76,173-206: CCV_32F | CCV_16F, mapped by
// ccv_nnc_nccl_datatype, lib/nnc/gpu/ccv_nnc_compat.cu:1447-1460).  Every tensor of one command has the first one's type: RCCL sums
// halves in half precision exactly as NCCL does for the reference, no widening behind the caller's back.
bool comm_tensor_ok(const ccv_nnc_tensor_t* t, size_t count, int datatype)
{
	return t && tensor_contiguous(t) && tensor_count(t->info) == count && CCV_GET_DATA_TYPE(t->info.datatype) == datatype;
}

enum { OP_ALLREDUCE, OP_BROADCAST, OP_REDUCE };

int comm_exec(const int op, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const ctx)
{
	const int n = op == OP_REDUCE ? input_size : output_size;
	if (n <= 0) return CCV_NNC_EXEC_INVALID;
	const ccv_nnc_tensor_t* first = op == OP_REDUCE ? outputs[0] : inputs[0];
	if (!first) return CCV_NNC_EXEC_INVALID;
	const size_t count = tensor_count(first->info);
	const int datatype = CCV_GET_DATA_TYPE(first->info.datatype);
	if (datatype != CCV_32F && datatype != CCV_16F) return CCV_NNC_EXEC_INVALID;
	const ncclDataType_t dt = datatype == CCV_16F ? ncclHalf : ncclFloat;
	// Lock order: every stream this command needs is resolved BEFORE g_comm_mutex is taken.  stream_of() runs the
	// look-ahead's hooks (peephole.cpp): it launches this stream's recorded command and WAITS for one another thread is still enqueueing
	// (wait_launching).  That other thread -- a loader whose host-to-device copy flushed every slot -- is inside the recorded command's exec
	// function, whose own stream_of() takes g_comm_mutex as soon as a collective is pending (comm_flush).  Waiting for it with the mutex held
	// and one device's collective already recorded deadlocked the two.  tl_in_comm keeps these stream_of() calls from flushing the pending
	// collectives (which would undo the coalescing); nothing below the lock calls back into the stream hooks.
	tl_in_comm++;
	int ret = CCV_NNC_EXEC_SUCCESS;
	int device_count = 0;
	hipStream_t streams[MAX_CLIQUE];
	int devices[MAX_CLIQUE];
	if (g_rank_comm) { // (b): one tensor per process
		if (n != 1 || !comm_tensor_ok(inputs[0], count, datatype) || !comm_tensor_ok(outputs[0], count, datatype)) ret = CCV_NNC_EXEC_INVALID;
		else streams[0] = stream_of(ctx);
	} else {
		if (n > MAX_CLIQUE) ret = CCV_NNC_EXEC_INVALID;
		for (int i = 0; i < n && ret == CCV_NNC_EXEC_SUCCESS; i++) {
			const ccv_nnc_tensor_t* t = op == OP_REDUCE ? inputs[i] : outputs[i];
			if (!comm_tensor_ok(t, count, datatype)) ret = CCV_NNC_EXEC_INVALID;
			else if (op == OP_ALLREDUCE && !comm_tensor_ok(inputs[i], count, datatype)) ret = CCV_NNC_EXEC_INVALID;
			else {
				devices[i] = CCV_TENSOR_GET_DEVICE_ID(t->info.type);
				if (devices[i] + 1 > device_count) device_count = devices[i] + 1;
			}
		}
		if (device_count > MAX_CLIQUE) ret = CCV_NNC_EXEC_INVALID;
		for (int i = 0; i < n && ret == CCV_NNC_EXEC_SUCCESS; i++) streams[i] = neighbor_stream(ctx, devices[i]);
	}
	if (ret != CCV_NNC_EXEC_SUCCESS) { tl_in_comm--; return ret; }
	pthread_mutex_lock(&g_comm_mutex);
	if (g_rank_comm)
		record(op == OP_ALLREDUCE ? 0 : op == OP_BROADCAST ? 1 : 2, inputs[0]->data.u8, outputs[0]->data.u8, count, dt, 0, g_rank_comm, streams[0], -1);
	else {
		const int root = op == OP_BROADCAST ? CCV_TENSOR_GET_DEVICE_ID(inputs[0]->info.type) : op == OP_REDUCE ? CCV_TENSOR_GET_DEVICE_ID(outputs[0]->info.type) : 0;
		clique_t* const cl = clique_of(ctx, device_count);
		for (int i = 0; i < n; i++) {
			const int d = devices[i];
			if (op == OP_ALLREDUCE) record(0, inputs[i]->data.u8, outputs[i]->data.u8, count, dt, 0, cl->comm[d], streams[i], d);
			else if (op == OP_BROADCAST) record(1, inputs[0]->data.u8, outputs[i]->data.u8, count, dt, root, cl->comm[d], streams[i], d);
			else record(2, inputs[i]->data.u8, outputs[0]->data.u8, count, dt, root, cl->comm[d], streams[i], d);
		}
	}
	pthread_mutex_unlock(&g_comm_mutex);
	tl_in_comm--;
	return ret;
}

static int _allreduce(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	const int n = input_size < output_size ? input_size : output_size;
	return comm_exec(OP_ALLREDUCE, inputs, n, outputs, n, stream_context);
}
static int _broadcast_forw(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	if (input_size < 1) return CCV_NNC_EXEC_INVALID;
	return comm_exec(OP_BROADCAST, inputs, 1, outputs, output_size, stream_context);
}
static int _reduce_forw(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	if (output_size < 1) return CCV_NNC_EXEC_INVALID;
	return comm_exec(OP_REDUCE, inputs, input_size, outputs, 1, stream_context);
}
// The gradient of a broadcast is a reduce of the incoming gradients and vice versa (comm_gpu_nccl.cu:151-165).
static int _broadcast_back(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	return comm_exec(OP_REDUCE, inputs, (input_size - 1) / 2, outputs, 1, stream_context);
}
static int _reduce_back(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	return comm_exec(OP_BROADCAST, inputs, 1, outputs, output_size, stream_context);
}

} // namespace

extern "C" {

int nnc_mi355x_comm_unique_id(void* id_out_128_bytes)
{
	ncclUniqueId id;
	static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
	if (ncclGetUniqueId(&id) != ncclSuccess) return -1;
	memcpy(id_out_128_bytes, &id, sizeof(id));
	return 0;
}
int nnc_mi355x_comm_init_rank(const void* id_128_bytes, int rank, int world_size)
{
	ncclUniqueId id;
	memcpy(&id, id_128_bytes, sizeof(id));
	pthread_mutex_lock(&g_comm_mutex);
	int ret = 0;
	if (g_rank_comm) ret = -1;
	else if ((nnc_mi355x_pool_trim(-1), ncclCommInitRank(&g_rank_comm, world_size, id, rank)) != ncclSuccess) { g_rank_comm = 0; ret = -2; }
	else { g_rank = rank; g_world = world_size; nnc::g_comm_overlap_on = overlap_wanted() ? 1 : 0; }
	pthread_mutex_unlock(&g_comm_mutex);
	return ret;
}
int nnc_mi355x_comm_count(void)
{ // ranks of the process communicator (deployment (b)), as RCCL itself counts them; 0 = none
	pthread_mutex_lock(&g_comm_mutex);
	int n = 0;
	if (g_rank_comm && ncclCommCount(g_rank_comm, &n) != ncclSuccess) n = -1;
	pthread_mutex_unlock(&g_comm_mutex);
	return n;
}
void nnc_mi355x_comm_stats(long* collectives, long* groups)
{
	pthread_mutex_lock(&g_comm_mutex);
	*collectives = g_stat_collectives; *groups = g_stat_groups;
	pthread_mutex_unlock(&g_comm_mutex);
}
/* Deployment (b): overlap the gradient all-reduces with the backward pass (see "Overlap" above).  on = 1 / 0; -1 = back to the environment's choice
 * (NNC_MI355X_COMM_OVERLAP).  Call it between steps, with nothing in flight. */
void nnc_mi355x_comm_overlap(const int on)
{
	nnc::comm_flush_if_pending();
	pthread_mutex_lock(&g_comm_mutex);
	g_overlap_set = on < 0 ? -1 : (on ? 1 : 0);
	nnc::g_comm_overlap_on = (g_rank_comm && overlap_wanted()) ? 1 : 0;
	pthread_mutex_unlock(&g_comm_mutex);
}
/* all-reduces that went out overlapped, and the buckets (group launches) they went out in */
void nnc_mi355x_comm_overlap_stats(long* const collectives, long* const buckets)
{
	pthread_mutex_lock(&g_comm_mutex);
	if (collectives) *collectives = g_stat_overlapped;
	if (buckets) *buckets = g_stat_buckets;
	pthread_mutex_unlock(&g_comm_mutex);
}
void nnc_mi355x_comm_destroy(void)
{
	pthread_mutex_lock(&g_comm_mutex);
	flush_locked();
	nnc::g_comm_overlap_on = 0;
	if (g_overlap_stream) HIP_ENFORCE(hipStreamSynchronize(g_overlap_stream)); // the last buckets
	if (g_rank_comm) { (void)ncclCommDestroy(g_rank_comm); g_rank_comm = 0; }
	pthread_mutex_unlock(&g_comm_mutex);
}

}

namespace nnc {
std::atomic<int> g_comm_pending(0);
std::atomic<int> g_comm_overlap_on(0);
std::atomic<unsigned long> g_comm_overlap_epoch(0);
void comm_gradients_written(ccv_nnc_tensor_t* const* const ts, const int n, ccv_nnc_stream_context_t* const ctx)
{ // ONE event behind the command for all the gradients it wrote
	if (!g_comm_overlap_on.load(std::memory_order_relaxed)) return;
	int any = 0;
	for (int i = 0; i < n; i++) if (ts[i]) any = 1;
	if (!any) return;
	const hipStream_t st = stream_peek(ctx);
	pthread_mutex_lock(&g_comm_mutex);
	ready_ev_t* const e = ready_event();
	HIP_ENFORCE(hipEventRecord(e->ev, st));
	const unsigned long seq = ++g_ready_seq;
	for (int i = 0; i < n; i++) {
		if (!ts[i]) continue;
		ready_t& r = g_ready[(const void*)ts[i]->data.u8];
		if (r.e == e) continue; // (two outputs on one buffer)
		ready_unref(r.e);
		r.e = e; r.seq = seq;
		e->refs++;
	}
	if (e->refs == 0) g_ready_pool.push_back(e);
	if (g_ready.size() > 65536) { for (auto& kv : g_ready) ready_unref(kv.second.e); g_ready.clear(); } // (a caller that never all-reduces what it reports)
	pthread_mutex_unlock(&g_comm_mutex);
}
void comm_gradient_touched(const ccv_nnc_tensor_t* const t)
{
	if (!t || !g_comm_overlap_on.load(std::memory_order_relaxed)) return;
	pthread_mutex_lock(&g_comm_mutex);
	auto at = g_ready.find((const void*)t->data.u8);
	if (at != g_ready.end()) { ready_unref(at->second.e); g_ready.erase(at); }
	pthread_mutex_unlock(&g_comm_mutex);
}
void comm_overlap_join(hipStream_t stream, unsigned long* const seen)
{
	pthread_mutex_lock(&g_comm_mutex);
	if (g_overlap_done && stream != g_overlap_stream) HIP_ENFORCE(hipStreamWaitEvent(stream, g_overlap_done, 0));
	*seen = g_comm_overlap_epoch.load(std::memory_order_acquire);
	pthread_mutex_unlock(&g_comm_mutex);
}
void comm_flush(void)
{ // called (through the g_comm_pending check) by stream_of, synchronise, signals, callbacks: see "Coalescing" above
	if (tl_in_comm) return;
	pthread_mutex_lock(&g_comm_mutex);
	flush_locked();
	pthread_mutex_unlock(&g_comm_mutex);
}
void comm_release_context(const void* ctx)
{ // the stream context is going away: its communicator sets with it
	pthread_mutex_lock(&g_comm_mutex);
	flush_locked();
	for (size_t i = 0; i < g_cliques.size();) {
		if (g_cliques[i]->ctx == ctx) {
			for (int d = 0; d < g_cliques[i]->device_count; d++) (void)ncclCommDestroy(g_cliques[i]->comm[d]);
			delete g_cliques[i];
			g_cliques.erase(g_cliques.begin() + i);
		} else i++;
	}
	pthread_mutex_unlock(&g_comm_mutex);
}
}

#define NNC_REG(CMD, EXEC) \
	extern "C" void _register_command_##CMD##_backend_CCV_NNC_BACKEND_GPU_NCCL(ccv_nnc_cmd_backend_registry_t* const registry) \
	{ registry->tensor_formats = CCV_TENSOR_FORMAT_NCHW | CCV_TENSOR_FORMAT_NHWC | CCV_TENSOR_FORMAT_CHWN; registry->tensor_datatypes = CCV_32F | CCV_16F; registry->tensor_memory = CCV_TENSOR_GPU_MEMORY; registry->algorithms = 1; registry->exec = EXEC; }
NNC_REG(CCV_NNC_COMM_ALLREDUCE_FORWARD, _allreduce)
NNC_REG(CCV_NNC_COMM_ALLREDUCE_BACKWARD, _allreduce)
NNC_REG(CCV_NNC_COMM_BROADCAST_FORWARD, _broadcast_forw)
NNC_REG(CCV_NNC_COMM_BROADCAST_BACKWARD, _broadcast_back)
NNC_REG(CCV_NNC_COMM_REDUCE_FORWARD, _reduce_forw)
NNC_REG(CCV_NNC_COMM_REDUCE_BACKWARD, _reduce_back)
