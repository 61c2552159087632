// TEST INFRASTRUCTURE: in-process stand-in for the RCCL API subset cmd_comm.cpp uses, for the CPU HIP emulator.
// Communicators created by ncclCommInitAll share one rendezvous object; a collective executes (on host memory) as soon
// as every rank of the clique has posted its call.  ncclCommInitRank supports nranks == 1 only.
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclFloat32 = 7, ncclFloat = 7, ncclFloat16 = 6, ncclHalf = 6 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
typedef struct { char internal[128]; } ncclUniqueId;
struct emu_nccl_clique;
struct emu_nccl_comm { emu_nccl_clique* clique; int rank; };
typedef emu_nccl_comm* ncclComm_t;
struct emu_nccl_op { int kind; const void* send; void* recv; size_t count; int root; int dt; };
struct emu_nccl_clique { int n; std::vector<emu_nccl_op> pending; std::vector<int> posted; };
static inline void emu_nccl_try_run(emu_nccl_clique* q)
{
	for (int r = 0; r < q->n; r++) if (!q->posted[r]) return;
	const size_t count = q->pending[0].count;
	const int kind = q->pending[0].kind, root = q->pending[0].root;
	std::vector<float> acc(count, 0.f);
	if (q->pending[0].dt == ncclHalf) { // half elements: each partial sum rounded to half, as a ring of half adders leaves it
		if (kind == 1) { const _Float16* s = (const _Float16*)q->pending[root].send; for (size_t i = 0; i < count; i++) acc[i] = (float)s[i]; }
		else for (int r = 0; r < q->n; r++) { const _Float16* s = (const _Float16*)q->pending[r].send; for (size_t i = 0; i < count; i++) acc[i] = (float)(_Float16)(acc[i] + (float)s[i]); }
		for (int r = 0; r < q->n; r++) { if (kind == 2 && r != root) continue; _Float16* d = (_Float16*)q->pending[r].recv; for (size_t i = 0; i < count; i++) d[i] = (_Float16)acc[i]; }
		for (int r = 0; r < q->n; r++) q->posted[r] = 0;
		return;
	}
	if (kind == 1) { const float* s = (const float*)q->pending[root].send; for (size_t i = 0; i < count; i++) acc[i] = s[i]; }
	else for (int r = 0; r < q->n; r++) { const float* s = (const float*)q->pending[r].send; for (size_t i = 0; i < count; i++) acc[i] += s[i]; }
	for (int r = 0; r < q->n; r++) { if (kind == 2 && r != root) continue; float* d = (float*)q->pending[r].recv; for (size_t i = 0; i < count; i++) d[i] = acc[i]; }
	for (int r = 0; r < q->n; r++) q->posted[r] = 0;
}
static inline ncclResult_t emu_nccl_post(ncclComm_t c, int kind, const void* s, void* r, size_t count, int root, int dt)
{
	emu_nccl_clique* q = c->clique;
	q->pending[c->rank] = emu_nccl_op{kind, s, r, count, root, dt};
	q->posted[c->rank] = 1;
	emu_nccl_try_run(q);
	return ncclSuccess;
}
static inline ncclResult_t ncclCommInitAll(ncclComm_t* comms, int n, const int*) { emu_nccl_clique* q = new emu_nccl_clique; q->n = n; q->pending.resize(n); q->posted.assign(n, 0); for (int i = 0; i < n; i++) comms[i] = new emu_nccl_comm{q, i}; return ncclSuccess; }
static inline ncclResult_t ncclGetUniqueId(ncclUniqueId* id) { memset(id, 0x5a, sizeof(*id)); return ncclSuccess; }
static inline ncclResult_t ncclCommInitRank(ncclComm_t* comm, int n, ncclUniqueId, int rank) { if (n != 1 || rank != 0) return ncclInvalidUsage; return ncclCommInitAll(comm, 1, 0); }
static inline ncclResult_t ncclCommCount(const ncclComm_t c, int* n) { *n = c->clique->n; return ncclSuccess; }
static inline ncclResult_t ncclCommDestroy(ncclComm_t c) { delete c; return ncclSuccess; }
static inline ncclResult_t ncclGroupStart() { return ncclSuccess; }
static inline ncclResult_t ncclGroupEnd() { return ncclSuccess; }
static inline ncclResult_t ncclAllReduce(const void* s, void* r, size_t count, ncclDataType_t dt, ncclRedOp_t, ncclComm_t c, hipStream_t) { return emu_nccl_post(c, 0, s, r, count, 0, dt); }
static inline ncclResult_t ncclBroadcast(const void* s, void* r, size_t count, ncclDataType_t dt, int root, ncclComm_t c, hipStream_t) { return emu_nccl_post(c, 1, s, r, count, root, dt); }
static inline ncclResult_t ncclReduce(const void* s, void* r, size_t count, ncclDataType_t dt, ncclRedOp_t, int root, ncclComm_t c, hipStream_t) { return emu_nccl_post(c, 2, s, r, count, root, dt); }
static inline const char* ncclGetErrorString(ncclResult_t) { return "emu-nccl error"; }
