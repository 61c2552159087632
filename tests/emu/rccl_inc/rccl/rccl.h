// TEST INFRASTRUCTURE: in-process stand-in for the RCCL API subset cmd_comm.cpp uses, for the CPU HIP emulator.
// Communicators created by ncclCommInitAll share one rendezvous object; a collective executes (on host memory) as soon
// as every rank of the clique has posted its call.
// ncclCommInitRank with nranks > 1 (round 4: the one-process-per-GPU form on CPU, world_size-2 tests): the ranks are PROCESSES; they meet in a POSIX
// shared-memory segment named after the unique id (rank 0's random bytes).  A collective = every rank copies its send buffer into its slot, a barrier,
// every rank reduces the slots in rank order (all arrive at the same bits), a barrier.  Blocking, like everything in this emulator.
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include <atomic>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclFloat32 = 7, ncclFloat = 7, ncclFloat16 = 6, ncclHalf = 6 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
typedef struct { char internal[128]; } ncclUniqueId;
struct emu_nccl_clique;
struct emu_nccl_shm { std::atomic<unsigned> arrived, generation, attached; unsigned nranks; size_t slot_bytes; }; // then nranks slots
struct emu_nccl_comm { emu_nccl_clique* clique; int rank; emu_nccl_shm* shm; char shm_name[64]; };
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
static const size_t EMU_NCCL_SLOT = (size_t)48 << 20; // bytes a rank may contribute to one collective
static inline bool emu_nccl_barrier(emu_nccl_shm* m)
{
	const unsigned gen = m->generation.load();
	if (m->arrived.fetch_add(1) + 1 == m->nranks) { m->arrived.store(0); m->generation.fetch_add(1); return true; }
	timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
	for (unsigned spins = 0; m->generation.load() == gen; spins++) {
		sched_yield();
		if ((spins & 0xfff) == 0xfff) { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); if (t.tv_sec - t0.tv_sec > 600) { fprintf(stderr, "emu-nccl: a rank never arrived\n"); return false; } }
	}
	return true;
}
static inline ncclResult_t emu_nccl_post_shm(ncclComm_t c, int kind, const void* s, void* r, size_t count, int root, int dt)
{
	emu_nccl_shm* const m = c->shm;
	const size_t esz = dt == ncclHalf ? 2 : 4, bytes = count * esz;
	if (bytes > m->slot_bytes) { fprintf(stderr, "emu-nccl: message of %zu bytes exceeds the slot\n", bytes); return ncclInvalidArgument; }
	char* const slots = (char*)(m + 1);
	const int n = (int)m->nranks;
	if (kind != 1 || c->rank == root) memcpy(slots + (size_t)c->rank * m->slot_bytes, s, bytes);
	if (!emu_nccl_barrier(m)) return ncclUnhandledCudaError;
	if (kind == 1) memcpy(r, slots + (size_t)root * m->slot_bytes, bytes);
	else if (kind == 0 || c->rank == root) {
		if (dt == ncclHalf) {
			_Float16* const d = (_Float16*)r;
			for (size_t i = 0; i < count; i++) { float acc = 0.f; for (int k = 0; k < n; k++) acc = (float)(_Float16)(acc + (float)((const _Float16*)(slots + (size_t)k * m->slot_bytes))[i]); d[i] = (_Float16)acc; }
		} else {
			float* const d = (float*)r;
			for (size_t i = 0; i < count; i++) { float acc = 0.f; for (int k = 0; k < n; k++) acc += ((const float*)(slots + (size_t)k * m->slot_bytes))[i]; d[i] = acc; }
		}
	}
	return emu_nccl_barrier(m) ? ncclSuccess : ncclUnhandledCudaError; // (the slots may be overwritten by the next collective from here on)
}
static inline ncclResult_t emu_nccl_post(ncclComm_t c, int kind, const void* s, void* r, size_t count, int root, int dt)
{
	if (c->shm) return emu_nccl_post_shm(c, kind, s, r, count, root, dt);
	emu_nccl_clique* q = c->clique;
	q->pending[c->rank] = emu_nccl_op{kind, s, r, count, root, dt};
	q->posted[c->rank] = 1;
	emu_nccl_try_run(q);
	return ncclSuccess;
}
static inline ncclResult_t ncclCommInitAll(ncclComm_t* comms, int n, const int*) { emu_nccl_clique* q = new emu_nccl_clique; q->n = n; q->pending.resize(n); q->posted.assign(n, 0); for (int i = 0; i < n; i++) comms[i] = new emu_nccl_comm{q, i, 0, {0}}; return ncclSuccess; }
static inline ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{ // 16 random bytes (they name the shared-memory segment of a cross-process communicator), the rest a fixed pattern
	memset(id, 0x5a, sizeof(*id));
	const int fd = open("/dev/urandom", O_RDONLY);
	if (fd >= 0) { if (read(fd, id->internal, 16) != 16) memset(id->internal, 0x5a, 16); close(fd); }
	return ncclSuccess;
}
static inline ncclResult_t ncclCommInitRank(ncclComm_t* comm, int n, ncclUniqueId id, int rank)
{
	if (n < 1 || rank < 0 || rank >= n) return ncclInvalidUsage;
	if (n == 1) return ncclCommInitAll(comm, 1, 0);
	emu_nccl_comm* const c = new emu_nccl_comm{0, rank, 0, {0}};
	static const char hex[] = "0123456789abcdef";
	char* w = c->shm_name;
	w += sprintf(w, "/nnc_emu_");
	for (int i = 0; i < 16; i++) { *w++ = hex[(unsigned char)id.internal[i] >> 4]; *w++ = hex[(unsigned char)id.internal[i] & 15]; }
	*w = 0;
	const size_t total = sizeof(emu_nccl_shm) + (size_t)n * EMU_NCCL_SLOT;
	const int fd = shm_open(c->shm_name, O_CREAT | O_RDWR, 0600);
	if (fd < 0 || ftruncate(fd, (off_t)total) != 0) { perror("emu-nccl shm"); delete c; return ncclUnhandledCudaError; }
	void* const p = mmap(0, total, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_NORESERVE, fd, 0); // (fresh pages of a shared-memory object read as zeros: the counters start at 0)
	close(fd);
	if (p == MAP_FAILED) { perror("emu-nccl mmap"); delete c; return ncclUnhandledCudaError; }
	c->shm = (emu_nccl_shm*)p;
	if (rank == 0) { c->shm->slot_bytes = EMU_NCCL_SLOT; c->shm->nranks = (unsigned)n; }
	c->shm->attached.fetch_add(1);
	for (unsigned spins = 0; c->shm->attached.load() < (unsigned)n || c->shm->nranks != (unsigned)n; spins++) { sched_yield(); if (spins > 200000000u) { fprintf(stderr, "emu-nccl: rendezvous timed out\n"); return ncclUnhandledCudaError; } }
	*comm = c;
	return ncclSuccess;
}
static inline ncclResult_t ncclCommCount(const ncclComm_t c, int* n) { *n = c->shm ? (int)c->shm->nranks : c->clique->n; return ncclSuccess; }
static inline ncclResult_t ncclCommDestroy(ncclComm_t c)
{
	if (c->shm) { if (c->shm->attached.fetch_sub(1) == 1) shm_unlink(c->shm_name); munmap(c->shm, sizeof(emu_nccl_shm) + (size_t)c->shm->nranks * EMU_NCCL_SLOT); }
	delete c;
	return ncclSuccess;
}
static inline ncclResult_t ncclGroupStart() { return ncclSuccess; }
static inline ncclResult_t ncclGroupEnd() { return ncclSuccess; }
// (a collective on a CAPTURING stream is recorded like any other node: the ranks' posts replay in the order they were issued, the last one runs the collective)
static inline ncclResult_t emu_nccl_call(ncclComm_t c, int kind, const void* s, void* r, size_t count, int root, int dt, hipStream_t st)
{
	if (st && emu::capturing(st)) { emu::capture_push(st, [=]() { (void)emu_nccl_post(c, kind, s, r, count, root, dt); }); return ncclSuccess; }
	return emu_nccl_post(c, kind, s, r, count, root, dt);
}
static inline ncclResult_t ncclAllReduce(const void* s, void* r, size_t count, ncclDataType_t dt, ncclRedOp_t, ncclComm_t c, hipStream_t st) { return emu_nccl_call(c, 0, s, r, count, 0, dt, st); }
static inline ncclResult_t ncclBroadcast(const void* s, void* r, size_t count, ncclDataType_t dt, int root, ncclComm_t c, hipStream_t st) { return emu_nccl_call(c, 1, s, r, count, root, dt, st); }
static inline ncclResult_t ncclReduce(const void* s, void* r, size_t count, ncclDataType_t dt, ncclRedOp_t, int root, ncclComm_t c, hipStream_t st) { return emu_nccl_call(c, 2, s, r, count, root, dt, st); }
static inline const char* ncclGetErrorString(ncclResult_t) { return "emu-nccl error"; }
