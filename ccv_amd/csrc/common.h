// Host-side helpers shared by the command implementations (C-style C++, compiled by hipcc).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <assert.h>
#include "../../include/nnc_mi355x.h"

// Device faults are fatal, exactly like the reference's *_ENFORCE macros (lib/nnc/gpu/ccv_nnc_compat.h:107-135):
// command exec functions only ever return CCV_NNC_EXEC_* codes.
#define HIP_ENFORCE(expr) do { \
	const hipError_t _st = (expr); \
	if (_st != hipSuccess) { \
		fprintf(stderr, "[%s:%d]:HIP - Error: %d (%s)\n", __FILE__, __LINE__, (int)_st, hipGetErrorString(_st)); \
		abort(); \
	} \
} while (0)

extern "C" void* nnc_staging_of(const ccv_nnc_stream_context_t* stream_context, size_t size); // device_rt.cpp
extern "C" void* nnc_palette_of(const ccv_nnc_stream_context_t* stream_context, size_t size); // device_rt.cpp: dense images of palettized inputs

namespace nnc {

static inline int tensor_nd(const int dim[CCV_NNC_MAX_DIM_ALLOC])
{ // lib/nnc/ccv_nnc.h:558
	int i;
	for (i = 0; i < CCV_NNC_MAX_DIM_ALLOC; i++)
		if (dim[i] == 0) return i;
	return CCV_NNC_MAX_DIM_ALLOC;
}
static inline size_t tensor_count(const ccv_nnc_tensor_param_t& p)
{ // lib/nnc/ccv_nnc_easy.h ccv_nnc_tensor_count
	if (p.dim[0] == 0) return 0;
	size_t c = 1;
	for (int i = 0; i < CCV_NNC_MAX_DIM_ALLOC && p.dim[i] > 0; i++) c *= (size_t)p.dim[i];
	return c;
}
static inline size_t datatype_size(int datatype)
{
	switch (CCV_GET_DATA_TYPE(datatype)) {
		case CCV_8U: return 1;
		case CCV_16F: return 2;
		case CCV_32S: case CCV_32F: return 4;
		case CCV_64S: case CCV_64F: return 8;
	}
	return 0;
}
// Element strides of a tensor or tensor view (lib/nnc/ccv_nnc_easy.h:315 ccv_nnc_tensor_view_get_stride).
static inline void tensor_strides(const ccv_nnc_tensor_t* t, int stride[CCV_NNC_MAX_DIM_ALLOC])
{
	const int nd = tensor_nd(t->info.dim);
	if (CCV_IS_TENSOR_VIEW(t)) {
		const ccv_nnc_tensor_view_t* tv = (const ccv_nnc_tensor_view_t*)t;
		for (int i = 0; i < CCV_NNC_MAX_DIM_ALLOC; i++) stride[i] = i < nd ? tv->stride[i] : 0;
		return;
	}
	int s = 1;
	for (int i = nd - 1; i >= 0; i--) { stride[i] = s; s *= t->info.dim[i]; }
	for (int i = nd; i < CCV_NNC_MAX_DIM_ALLOC; i++) stride[i] = 0;
}
static inline bool tensor_contiguous(const ccv_nnc_tensor_t* t)
{
	if (!CCV_IS_TENSOR_VIEW(t)) return true;
	int st[CCV_NNC_MAX_DIM_ALLOC];
	tensor_strides(t, st);
	const int nd = tensor_nd(t->info.dim);
	int s = 1;
	for (int i = nd - 1; i >= 0; i--) { if (t->info.dim[i] != 1 && st[i] != s) return false; s *= t->info.dim[i]; }
	return true;
}
static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// A 4-d (N, H, W, C) logical view of an image-like tensor in either layout, with element strides.
struct Image4 {
	float* p;
	int n, h, w, c;
	long sn, sh, sw, sc;
};
// 3-d tensors have no batch dimension (lib/nnc/ccv_nnc_internal.h:44-55 ccv_nnc_tensor_hw).
static inline bool image4(const ccv_nnc_tensor_t* t, Image4* o)
{
	const int nd = tensor_nd(t->info.dim);
	if (nd != 3 && nd != 4) return false;
	int st[CCV_NNC_MAX_DIM_ALLOC];
	tensor_strides(t, st);
	const int* d = t->info.dim;
	const int b = (nd == 4);
	o->p = t->data.f32;
	o->n = b ? d[0] : 1;
	o->sn = b ? st[0] : 0;
	if (t->info.format == CCV_TENSOR_FORMAT_NHWC) {
		o->h = d[b]; o->w = d[b + 1]; o->c = d[b + 2];
		o->sh = st[b]; o->sw = st[b + 1]; o->sc = st[b + 2];
	} else if (t->info.format == CCV_TENSOR_FORMAT_NCHW) {
		o->c = d[b]; o->h = d[b + 1]; o->w = d[b + 2];
		o->sc = st[b]; o->sh = st[b + 1]; o->sw = st[b + 2];
	} else
		return false;
	return true;
}

// Recorded-but-not-yet-issued collectives (cmd_comm.cpp "Coalescing"): anything that could observe stream order flushes them first.
extern std::atomic<int> g_comm_pending; // (atomics, not volatile ints: these are read outside their mutexes by every order-observing hook on any thread -- ThreadSanitizer run, round 4)
void comm_flush(void);
void comm_release_context(const void* ctx);
// Deployment (b) with NNC_MI355X_COMM_OVERLAP=1 / nnc_mi355x_comm_overlap(1): the gradient all-reduces go out on a communication stream in buckets, each behind
// the commands that WROTE its gradients (not behind the whole issuing stream), and every other stream joins them at its next order-observing point.
extern std::atomic<int> g_comm_overlap_on;            // 1 while the mode is on and a rank communicator exists: the backward commands then report their gradient outputs
extern std::atomic<unsigned long> g_comm_overlap_epoch; // bumped by every overlapped flush
void comm_gradients_written(ccv_nnc_tensor_t* const* ts, int n, ccv_nnc_stream_context_t* ctx); // a backward command has just enqueued the kernels that write these weight / bias gradients (null entries skipped): one event for all of them
void comm_gradient_touched(const ccv_nnc_tensor_t* t); // some other command writes it too (accumulation): forget the record, the all-reduce takes stream order
void comm_overlap_join(hipStream_t stream, unsigned long* seen);
hipStream_t stream_peek(const ccv_nnc_stream_context_t* ctx); // device_rt.cpp: the context's stream, no hooks
void stream_registered(int device, hipStream_t st); // device_rt.cpp: a stream this library made outside it -- the allocator's free fences must name it
// Recorded-but-not-yet-launched commands waiting for the ReLU that may follow them (peephole.cpp): the same points flush them.
typedef int (*exec_fn_t)(const ccv_nnc_cmd_t, const ccv_nnc_hint_t, const int, ccv_nnc_tensor_t* const*, const int, ccv_nnc_tensor_t* const*, const int, ccv_nnc_stream_context_t*);
enum { DEFER_CONV_FORWARD = 1, DEFER_CONV_BACKWARD = 2, DEFER_POOL_BACKWARD = 3, DEFER_BNORM_FORWARD = 4, DEFER_EWSUM_FORWARD = 5, DEFER_SGD_BATCH = 6 };
extern std::atomic<int> g_deferred_live;
bool deferred_try(exec_fn_t fn, int kind, const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, int flags, ccv_nnc_tensor_t* const* inputs, int input_size, ccv_nnc_tensor_t* const* outputs, int output_size, ccv_nnc_stream_context_t* ctx, uint64_t* sig); // true: recorded, report success
void deferred_mark_good(uint64_t sig); // this signature ran successfully on the spot: the next one like it may be recorded
int deferred_fuse_relu_forw(const ccv_nnc_tensor_t* a, ccv_nnc_tensor_t* b, ccv_nnc_stream_context_t* ctx); // -1: no recorded command this ReLU completes
int deferred_fuse_relu_back(const ccv_nnc_tensor_t* g, const ccv_nnc_tensor_t* b, ccv_nnc_tensor_t* h, ccv_nnc_stream_context_t* ctx);
bool deferred_signal_op(int emit, const ccv_nnc_stream_context_t* ctx, const ccv_nnc_stream_signal_t* signal); // true: kept in a recorded command's trail (peephole.cpp), not to be performed now
bool deferred_trail_cmd(exec_fn_t fn, const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, int flags, ccv_nnc_tensor_t* const* inputs, int input_size, ccv_nnc_tensor_t* const* outputs, int output_size, ccv_nnc_stream_context_t* ctx); // true: kept behind the trail operations of its stream
bool deferred_sgd_head(exec_fn_t fn, const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, int flags, ccv_nnc_tensor_t* const* inputs, int input_size, ccv_nnc_tensor_t* const* outputs, int output_size, ccv_nnc_stream_context_t* ctx); // true: kept as the first of a batch of updates on its stream
extern std::atomic<unsigned long> g_launch_seq; // device_rt.cpp: stream_of() calls so far
void deferred_sgd_launched(const ccv_nnc_stream_context_t* ctx); // an update has just been launched on the spot on this stream: the NEXT one, if it arrives with nothing launched in between, starts a batch
bool sgd_is_exec(exec_fn_t fn); // cmd_ew.cpp: is this the SGD_FORWARD exec function (whose consecutive trail entries sgd_forw_multi can launch together)?
int sgd_forw_multi(const ccv_nnc_cmd_t* const* cmds, ccv_nnc_tensor_t* const* const* ins, ccv_nnc_tensor_t* const* const* outs, int n, ccv_nnc_stream_context_t* ctx); // 0: launched; > 0: not batchable, run them one by one
void signal_emit_now(const ccv_nnc_stream_context_t* ctx, const ccv_nnc_stream_signal_t* signal); // device_rt.cpp: the event record / stream wait themselves
void signal_wait_now(const ccv_nnc_stream_context_t* ctx, const ccv_nnc_stream_signal_t* signal);
void deferred_flush(const ccv_nnc_stream_context_t* ctx); // 0: every stream's
int deferred_take_error(const ccv_nnc_stream_context_t* ctx); // a recorded command of this stream failed when a flush launched it: returned (once) by the stream's next recordable command
void deferred_suppress(int delta); // +1 / -1 around commands that must run on the spot (half_stage.cpp)
static inline void comm_flush_if_pending(void) { if (g_comm_pending) comm_flush(); if (g_deferred_live) deferred_flush(0); }

// The HIP stream a command must enqueue on, and that stream's scratch memory.
hipStream_t stream_of(const ccv_nnc_stream_context_t* ctx);
// device_rt.cpp "HIP-graph capture": null, or -- `st` is recording a step -- the word (pinned host memory) the graph's first node increments at every replay:
// kernels whose host-drawn seed would otherwise repeat mix it in
const unsigned* capture_tick_of(hipStream_t st);
void* workspace_of(const ccv_nnc_stream_context_t* ctx, size_t size);
const float* zero_page_of(const ccv_nnc_stream_context_t* ctx); // 256 zero bytes in the HBM of the device `ctx` launches on
int device_cu_count(void);
// kernels whose workgroups wait for each other inside one launch: the stream's hand-over area (device_rt.cpp), this launch's epoch, the timeout word
struct cluster_sync_t { unsigned ticket, done; unsigned pad[62]; }; // 256 bytes, the granules follow
constexpr size_t CLUSTER_SYNC_BYTES = 2u << 20;
constexpr unsigned CLUSTER_SPIN_LIMIT = 1u << 21; // polls (each ~ a microsecond) before a waiting workgroup gives up: seconds, not a hung GPU
void* cluster_sync_of(const ccv_nnc_stream_context_t* ctx, size_t granule_bytes, unsigned* epoch, unsigned** timeout_word);
// Kernels whose workgroups wait for each other need their waiting sets RESIDENT; two such launches from two streams of one device can each hold part of
// the CUs and starve one another until CLUSTER_SPIN_LIMIT (ordinary kernels cannot: they finish and free their CUs).  A ClusterTurn brackets the
// launches of one command: the stream first waits for the event behind the previous turn taken on ANOTHER stream of the device, and leaves an event
// behind its own launches -- spinning launches of one process are one after the other per device, whatever streams the host's scheduler picked.
struct ClusterTurn {
	hipStream_t stream;
	int device;
	explicit ClusterTurn(const ccv_nnc_stream_context_t* ctx);
	~ClusterTurn();
	ClusterTurn(const ClusterTurn&) = delete;
	ClusterTurn& operator=(const ClusterTurn&) = delete;
};
long cluster_turns_chained(void); // (tests) turns that had to wait for a turn on another stream
void note_kernel(const char* name);

long tune(int key);
static inline int grid_for(size_t n, int threads)
{ // memory-bound grid-stride kernels.  The cap (TUNE_GRID_WG_PER_CU workgroups per CU) defaults to none: see device_rt.cpp.
	size_t b = (n + threads - 1) / threads;
	const long per_cu = tune(3 /* TUNE_GRID_WG_PER_CU, checked below the enum */);
	const size_t cap = per_cu > 0 ? (size_t)device_cu_count() * (size_t)per_cu : (size_t)0x7fffffff;
	if (b > cap) b = cap;
	if (b < 1) b = 1;
	return (int)b;
}

// Carve a caller-owned prefix out of the stream's single grow-only workspace for the lifetime of the object: the whole
// (prefix + inner) size is requested up front so the base cannot move, and workspace_of() calls made underneath (the
// contraction launcher's split-K slabs) are handed the region behind the prefix.
struct WorkspaceScope {
	char* base;
	size_t prev, prev_limit;
	WorkspaceScope(const ccv_nnc_stream_context_t* ctx, size_t prefix_bytes, size_t inner_bytes);
	~WorkspaceScope();
	void* prefix() const { return base; }
};
size_t gemm_workspace_bound(long M, long N, long K); // most bytes gemm_run() can ask of workspace_of() for one contraction

// Layout helpers (cmd_util.cpp).
int format_transform(const ccv_nnc_tensor_t* a, ccv_nnc_tensor_t* b, ccv_nnc_stream_context_t* ctx);
int transpose_half_to_float(const void* in, float* out, int batch, int R, int C, ccv_nnc_stream_context_t* ctx); // in[batch][R][C] halves -> out[batch][C][R] floats
int transpose_float_to_half(const float* in, void* out, int batch, int R, int C, ccv_nnc_stream_context_t* ctx);
int transpose_half(const void* in, void* out, int batch, int R, int C, ccv_nnc_stream_context_t* ctx); // in[batch][R][C] halves -> out[batch][C][R] halves
// The same pass that also SUMS every input row (fp32) over its tile's 64 columns: row_partial[(b * ceil(C / 64) + tile) * R + r].  A convolution's backward pass re-lays the NCHW output
// gradient anyway; with the sums taken on the way, the bias gradient is a fold of these partials (colsum_partials_f16) instead of another pass over the gradient.
// CCV_NNC_EXEC_NO_KERNEL: the tensor is not in whole 16-byte chunks -- nothing was launched, the caller takes the two separate passes.
int transpose_half_rowsum(const void* in, void* out, int batch, int R, int C, float* row_partial, ccv_nnc_stream_context_t* ctx);
inline long transpose_half_rowsum_slices(const int batch, const int C) { return (long)batch * ((C + 63) / 64); }
int relu_inplace(ccv_nnc_tensor_t* t, ccv_nnc_stream_context_t* ctx); // t = max(t, 0), dense CCV_32F / CCV_16F (cmd_ew.cpp)
int relu_back_inplace(ccv_nnc_tensor_t* h, const ccv_nnc_tensor_t* b, ccv_nnc_stream_context_t* ctx); // h = b > 0 ? h : 0, dense, same type and count
int weights_nchw_to_nhwc(const float* w, float* out, int K, int C, int khw, ccv_nnc_stream_context_t* ctx);
int weights_nhwc_to_nchw(const float* w, float* out, int K, int C, int khw, ccv_nnc_stream_context_t* ctx);

// Shared device helpers (cmd_ew.cpp).
int colsum_f32(const float* x, long rows, int cols, long ld, float* out, int accumulate, ccv_nnc_stream_context_t* ctx); // out[c] (+)= sum_r x[r*ld + c]
int fill_f32(float* p, size_t n, float v, ccv_nnc_stream_context_t* ctx);
int chan_sum_planes(const float* x, long outer, int C, long inner, float* out, int accumulate, ccv_nnc_stream_context_t* ctx); // cmd_norm.cpp: out[c] (+)= sum_{o,i} x[(o * C + c) * inner + i]

// Optional in-library kernel timing (bench.py roofline leg): when enabled, a ProfScope brackets ONE kernel launch with
// HIP events on the stream the kernel is launched on and files (name, algorithmic flops/bytes, problem dims).
struct ProfScope {
	void* rec;
	hipStream_t stream;
	ProfScope(const char* name, double flops, double bytes, int M, int N, int K, int Z, int S, hipStream_t stream);
	~ProfScope();
};

void prof_next_bytes(double bytes); // the algorithmic bytes of this thread's next contraction record (a ProfScope given bytes < 0 takes them; else it uses -bytes)

// Marker range around one command (device_rt.cpp; on after nnc_mi355x_set_profiler(1) / cusetprofiler(1) or NNC_MI355X_MARKERS=1).
struct MarkerScope {
	int active;
	explicit MarkerScope(uint32_t cmd);
	~MarkerScope();
};
void markers_enable(int on);
const char* command_row_name(uint32_t cmd); // registry.cpp: "CMD/BACKEND" of the row that implements cmd

// nnc_mi355x_debug_force_tile (device_rt.cpp): wm | wn << 8, 0 = built-in choice.
extern int g_force_tile;
extern int g_force_splits; // nnc_mi355x_debug_force_splits: 0 = built-in choice

// Tunables (nnc_mi355x_tune_set / environment NNC_MI355X_<NAME>, device_rt.cpp): performance policy only, never semantics.
enum {
	TUNE_WINO_SLICE_KB = 0, // Winograd via HBM: run the three stages per slice of images whose V + M scratch is at most this many KB (0 = whole batch)
	TUNE_WINO_FUSED_MAX_C,  // algorithm -1 picks the fused Winograd kernel when the reduction channels are <= this (0 = never)
	TUNE_WINO_FUSED_GRID,   // persistent workgroups of the fused Winograd kernel (0 = one per CU)
	TUNE_GRID_WG_PER_CU,    // grid-stride kernels (grid_for): cap in workgroups per CU, 0 = no cap (one trip per thread)
	TUNE_WINO_WGRAD_FUSED_MAX, // algorithm -1 takes the fused Winograd filter gradient when both channel counts are <= this (0 = never)
	TUNE_GEMM_BUFFER_LOADS, // plain-matrix contractions fetch their operands with buffer loads (no address VALU in the K loop); 0 = the pointer path;
	                        // half precision: 2 = never the 256 x 256 tile, 3 = that tile wherever it fits, 4 = K-steps of 32 only (gemm_launch.h)
	TUNE_CONV_NCHW_HALF_F16, // half NCHW convolutions larger than 1 x 1: forward / data gradient on the f16 implicit-GEMM core between half transposes when the
	                        // reduction has at least this many channels (0 = never: the fp32 Winograd kernels between converting transposes).  Default 32 since round 4
	                        // (was 64): with 16-byte chunks and the planar epilogue the 32-channel stem layers of ResNet-50 are 5.5 % of the f16 step faster there
	TUNE_BN_SMALL_PLANES,   // batch norm on [N][C][planes]: four planes per wave when a plane is at most 1 KB (1), or a wave per plane always (0)
	TUNE_SDPA_MFMA,         // scaled-dot-product attention forward on the matrix cores where the shapes allow (1), or the VALU kernel always (0)
	TUNE_BN_CLUSTER,        // batch norm (training) on [N][C][planes]: a cluster of workgroups per channel holds the channel in registers between the statistics and the apply pass -- x read ONCE (1), or the plane kernels (0); > 1: chunks per workgroup (tests force several workgroups per channel on small tensors)
	TUNE_GEMM_VEC_EPILOGUE, // contraction epilogues stage the block tile through LDS and store 16-byte (8-byte for halves) row segments where the output allows, half NCHW convolutions write their planar result themselves (1); 2 = the row segments only; 0 = one element per lane always
	TUNE_POOL_ROWS,         // pooling on NCHW tensors: a lane per four consecutive x of a row where the maps allow (1), or a lane per element always (0)
	TUNE_GEMM_HALF_CHUNK8,  // the half-precision contraction kernel stages its operands in 16-byte chunks of eight halves where strides and channel counts allow (1), or always in 8-byte chunks of four (0)
	TUNE_LSTM_PERSISTENT,   // LSTM: one launch walks a pseudo-layer's whole sequence with its slice of R in registers, the state handed between workgroups through tagged words (1), or one launch per step (0)
	TUNE_LSTM_ROWS,         // LSTM without projection, hidden size <= 128: a workgroup owns ONE batch row (two when the hidden size is no multiple of four; 2 = two for every size) and ALL hidden units for the whole sequence, R entirely in its registers, no word passes between workgroups (1), or the forms above (0)
	TUNE_GEMM_BF16X3,       // fp32 plain-matrix contractions on the bf16 matrix pipe, every operand split exactly into three bf16 values and all nine partial products accumulated in fp32 (mfma_gemm_bf16x3.h): 0 = never (the fp32 matrix instructions), 1 = where the launcher's rules say it pays, 2 = wherever the kernel applies, 3 / 4 = as 2 with the 128 x 128 / 256 x 256 tile forced (measurements)
	TUNE_BN_CLUSTER_SLOTS,  // the cluster batch-norm kernels on tensors too small to fill the chip with full workgroup shares: shares are cut down until the launch has about this many workgroups (never below two chunks per thread); 0 = always the largest share a workgroup's registers hold (rounds 4 - 5)
	TUNE_GEMM_BATCH_XCD,    // batched contractions without split-K (a 1 x 1 convolution on NCHW tensors: one matrix product per image) launch ONE grid dimension over (entry, tile) and give every batch entry to one XCD: the tiles of an entry meet in ONE L2, so its B operand -- the image's planes, which every row block of the output reads -- leaves HBM once, not once per XCD (1); 0 = entries on grid z, tiles dealt round-robin over the XCDs (rounds 1 - 5)
	TUNE_COUNT
};
static_assert(TUNE_GRID_WG_PER_CU == 3, "grid_for() above names this key by value");
long tune(int key);

// Folding per-slice partials [slices][C] into per-channel sums in a fixed order: a workgroup of 256 is 16 channels x 16 phases, thread
// (phase = t >> 4, channel = t & 15) adds the slices phase, phase + 16, ... with four independent running sums (four loads in
// flight), the 16 phases meet in LDS.  (One thread per channel walking every slice -- one dependent add chain -- was 42 - 68 us per
// call on the VGG-D / ResNet-50 steps: as long as the reductions these folds finish.)
constexpr int FOLD_CH = 16, FOLD_PH = 16;
__device__ __forceinline__ float fold_slices(const float* __restrict__ p, const long slices, const int C, const int c, const int phase)
{
	float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
	long i = phase;
	for (; i + 3 * FOLD_PH < slices; i += 4 * FOLD_PH) {
		s0 += p[i * C + c]; s1 += p[(i + FOLD_PH) * C + c]; s2 += p[(i + 2 * FOLD_PH) * C + c]; s3 += p[(i + 3 * FOLD_PH) * C + c];
	}
	for (; i < slices; i += FOLD_PH) s0 += p[i * C + c];
	return (s0 + s1) + (s2 + s3);
}
__device__ __forceinline__ float fold_phases(float (*red)[FOLD_CH], const int ch)
{ // after __syncthreads(): the 16 phase sums of channel ch, pairwise in a fixed order
	float a[FOLD_PH];
#pragma unroll
	for (int k = 0; k < FOLD_PH; k++) a[k] = red[k][ch];
#pragma unroll
	for (int w = FOLD_PH / 2; w >= 1; w >>= 1)
#pragma unroll
		for (int k = 0; k < w; k++) a[k] = a[k] + a[k + w];
	return a[0];
}

// Half precision (half_stage.cpp): rows whose kernels compute in fp32 run CCV_16F tensors through fp32 images in the stream's
// staging arena.  NNC_HALF_STAGED(registry, EXEC) -- used by every row's registration -- adds CCV_16F next to CCV_32F and routes
// the row through the wrapper (which is a plain call of EXEC when no tensor is half precision).
typedef int (*nnc_exec_f)(const ccv_nnc_cmd_t, const ccv_nnc_hint_t, const int, ccv_nnc_tensor_t* const* const, const int, ccv_nnc_tensor_t* const* const, const int, ccv_nnc_stream_context_t* const);
int half_staged_exec(nnc_exec_f inner, const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const ctx);
template <nnc_exec_f F>
static int half_staged(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const ctx)
{
	return half_staged_exec(F, cmd, hint, flags, inputs, input_size, outputs, output_size, ctx);
}
#define NNC_HALF_STAGED(registry, EXEC) do { if (((registry)->tensor_datatypes & CCV_32F) && !((registry)->tensor_datatypes & CCV_16F)) { /* rows that list CCV_16F themselves handle it natively */ (registry)->tensor_datatypes |= CCV_16F; (registry)->exec = nnc::half_staged<EXEC>; } } while (0)
bool any_half_tensor(ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size);
int half_to_float(const void* in, float* out, size_t n, ccv_nnc_stream_context_t* ctx);
int float_to_half(const float* in, void* out, size_t n, ccv_nnc_stream_context_t* ctx);
int chan_sum_planes_f16(const void* x, long outer, int C, long inner, void* out, int accumulate, ccv_nnc_stream_context_t* ctx);
int colsum_f16(const void* x, long rows, int cols, long ld, void* out, int accumulate, ccv_nnc_stream_context_t* ctx); // halves: out[c] (+)= sum_r x[r * ld + c], fp32 sums
int colsum_partials_f16(const float* partial, long slices, int cols, void* out, int accumulate, ccv_nnc_stream_context_t* ctx); // out[c] (+)= sum_s partial[s * cols + c], slices in a fixed order, rounded to half once; `partial` has room for (slices + 256) * cols floats (a second, grouped level)

// Palettized inputs (palette.cpp): rows whose reference counterparts list CCV_QX (GEMM, convolution, transposed convolution, attention's head projection) run
// on dense images of their CCV_QX inputs.  NNC_DEPALETTIZED(registry, EXEC) adds CCV_QX to the row and routes it through the wrapper (a plain call of EXEC
// when no input is palettized); EXEC is whatever the row's exec would have been (the half-staged form included).
int depalettize(const void* input, int datatype, size_t input_length, int qbits, int number_in_blocks, void* output, size_t output_length, ccv_nnc_stream_context_t* ctx);
size_t palettized_bytes(int palette_datatype, size_t count, int qbits, int number_in_blocks);
bool any_palettized(ccv_nnc_tensor_t* const* const inputs, const int input_size);
int depalettized_exec(nnc_exec_f inner, const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const ctx);
template <nnc_exec_f F>
static int depalettized(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const ctx)
{
	return depalettized_exec(F, cmd, hint, flags, inputs, input_size, outputs, output_size, ctx);
}
#define NNC_DEPALETTIZED(registry, EXEC) do { (registry)->tensor_datatypes |= CCV_QX; (registry)->exec = nnc::depalettized<EXEC>; } while (0)

// Registration table (registry.cpp).
typedef void (*register_fn_t)(ccv_nnc_cmd_backend_registry_t* const);

} // namespace nnc
