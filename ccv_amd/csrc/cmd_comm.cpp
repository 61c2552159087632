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

using namespace nnc;

#define RCCL_ENFORCE(expr) do { \
	const ncclResult_t _st = (expr); \
	if (_st != ncclSuccess) { fprintf(stderr, "[%s:%d]:RCCL - Error: %d (%s)\n", __FILE__, __LINE__, (int)_st, ncclGetErrorString(_st)); abort(); } \
} while (0)

namespace {

constexpr int MAX_CLIQUE = 64;
pthread_mutex_t g_comm_mutex = PTHREAD_MUTEX_INITIALIZER;
ncclComm_t g_clique[MAX_CLIQUE + 1][MAX_CLIQUE]; // [device_count][device]
bool g_clique_ready[MAX_CLIQUE + 1];
ncclComm_t g_rank_comm = 0; // deployment (b)
int g_rank = 0, g_world = 1;

ncclComm_t clique_comm(int device_count, int device)
{
	pthread_mutex_lock(&g_comm_mutex);
	if (!g_clique_ready[device_count]) {
		int devs[MAX_CLIQUE];
		for (int i = 0; i < device_count; i++) devs[i] = i;
		int cur = 0;
		HIP_ENFORCE(hipGetDevice(&cur));
		RCCL_ENFORCE(ncclCommInitAll(g_clique[device_count], device_count, devs));
		HIP_ENFORCE(hipSetDevice(cur));
		g_clique_ready[device_count] = true;
	}
	ncclComm_t c = g_clique[device_count][device];
	pthread_mutex_unlock(&g_comm_mutex);
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

bool comm_tensor_ok(const ccv_nnc_tensor_t* t, size_t count)
{
	return t && tensor_contiguous(t) && tensor_count(t->info) == count && CCV_GET_DATA_TYPE(t->info.datatype) == CCV_32F;
}

enum { OP_ALLREDUCE, OP_BROADCAST, OP_REDUCE };

int comm_exec(const int op, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const ctx)
{
	const int n = op == OP_REDUCE ? input_size : output_size;
	if (n <= 0) return CCV_NNC_EXEC_INVALID;
	const ccv_nnc_tensor_t* first = op == OP_REDUCE ? outputs[0] : inputs[0];
	if (!first) return CCV_NNC_EXEC_INVALID;
	const size_t count = tensor_count(first->info);
	if (g_rank_comm) { // (b): one tensor per process
		if (n != 1 || !comm_tensor_ok(inputs[0], count) || !comm_tensor_ok(outputs[0], count)) return CCV_NNC_EXEC_INVALID;
		hipStream_t st = stream_of(ctx);
		if (op == OP_ALLREDUCE) RCCL_ENFORCE(ncclAllReduce(inputs[0]->data.f32, outputs[0]->data.f32, count, ncclFloat, ncclSum, g_rank_comm, st));
		else if (op == OP_BROADCAST) RCCL_ENFORCE(ncclBroadcast(inputs[0]->data.f32, outputs[0]->data.f32, count, ncclFloat, 0, g_rank_comm, st));
		else RCCL_ENFORCE(ncclReduce(inputs[0]->data.f32, outputs[0]->data.f32, count, ncclFloat, ncclSum, 0, g_rank_comm, st));
		return CCV_NNC_EXEC_SUCCESS;
	}
	int device_count = 0;
	for (int i = 0; i < n; i++) {
		const ccv_nnc_tensor_t* t = op == OP_REDUCE ? inputs[i] : outputs[i];
		if (!comm_tensor_ok(t, count)) return CCV_NNC_EXEC_INVALID;
		if (op == OP_ALLREDUCE && !comm_tensor_ok(inputs[i], count)) return CCV_NNC_EXEC_INVALID;
		const int d = CCV_TENSOR_GET_DEVICE_ID(t->info.type);
		if (d + 1 > device_count) device_count = d + 1;
	}
	if (device_count > MAX_CLIQUE) return CCV_NNC_EXEC_INVALID;
	int cur = 0;
	HIP_ENFORCE(hipGetDevice(&cur));
	const int root = op == OP_BROADCAST ? CCV_TENSOR_GET_DEVICE_ID(inputs[0]->info.type) : op == OP_REDUCE ? CCV_TENSOR_GET_DEVICE_ID(outputs[0]->info.type) : 0;
	for (int i = 0; i < n; i++) { // create (cache) the clique before the group
		const ccv_nnc_tensor_t* t = op == OP_REDUCE ? inputs[i] : outputs[i];
		clique_comm(device_count, CCV_TENSOR_GET_DEVICE_ID(t->info.type));
	}
	RCCL_ENFORCE(ncclGroupStart());
	for (int i = 0; i < n; i++) {
		const ccv_nnc_tensor_t* t = op == OP_REDUCE ? inputs[i] : outputs[i];
		const int d = CCV_TENSOR_GET_DEVICE_ID(t->info.type);
		ncclComm_t comm = clique_comm(device_count, d);
		hipStream_t st = neighbor_stream(ctx, d);
		if (op == OP_ALLREDUCE) RCCL_ENFORCE(ncclAllReduce(inputs[i]->data.f32, outputs[i]->data.f32, count, ncclFloat, ncclSum, comm, st));
		else if (op == OP_BROADCAST) RCCL_ENFORCE(ncclBroadcast(inputs[0]->data.f32, outputs[i]->data.f32, count, ncclFloat, root, comm, st));
		else RCCL_ENFORCE(ncclReduce(inputs[i]->data.f32, outputs[0]->data.f32, count, ncclFloat, ncclSum, root, comm, st));
	}
	RCCL_ENFORCE(ncclGroupEnd());
	HIP_ENFORCE(hipSetDevice(cur));
	return CCV_NNC_EXEC_SUCCESS;
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
	else if (ncclCommInitRank(&g_rank_comm, world_size, id, rank) != ncclSuccess) { g_rank_comm = 0; ret = -2; }
	else { g_rank = rank; g_world = world_size; }
	pthread_mutex_unlock(&g_comm_mutex);
	return ret;
}
void nnc_mi355x_comm_destroy(void)
{
	pthread_mutex_lock(&g_comm_mutex);
	if (g_rank_comm) { (void)ncclCommDestroy(g_rank_comm); g_rank_comm = 0; }
	pthread_mutex_unlock(&g_comm_mutex);
}

}

#define NNC_REG(CMD, EXEC) \
	extern "C" void _register_command_##CMD##_backend_CCV_NNC_BACKEND_GPU_NCCL(ccv_nnc_cmd_backend_registry_t* const registry) \
	{ registry->tensor_formats = CCV_TENSOR_FORMAT_NCHW | CCV_TENSOR_FORMAT_NHWC | CCV_TENSOR_FORMAT_CHWN; registry->tensor_datatypes = CCV_32F; registry->tensor_memory = CCV_TENSOR_GPU_MEMORY; registry->algorithms = 1; registry->exec = EXEC; }
NNC_REG(CCV_NNC_COMM_ALLREDUCE_FORWARD, _allreduce)
NNC_REG(CCV_NNC_COMM_ALLREDUCE_BACKWARD, _allreduce)
NNC_REG(CCV_NNC_COMM_BROADCAST_FORWARD, _broadcast_forw)
NNC_REG(CCV_NNC_COMM_BROADCAST_BACKWARD, _broadcast_back)
NNC_REG(CCV_NNC_COMM_REDUCE_FORWARD, _reduce_forw)
NNC_REG(CCV_NNC_COMM_REDUCE_BACKWARD, _reduce_back)
