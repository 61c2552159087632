// ReLU folded into its neighbour WITHOUT the caller's help: a one-command look-ahead per stream.
//
// The reference's graphs issue CONVOLUTION_FORWARD (or, in the conv - bn - relu blocks of its ResNet / CIFAR-10 models, BATCH_NORM_FORWARD; or, at the
// end of a residual block, EWSUM_FORWARD) followed by an in-place RELU_FORWARD on its output
// (test/int/nnc/graph.vgg.d.tests.c:14-90, bin/nnc/cifar-10.c:76-127 through ccv_cnnp's convolution + relu blocks), and on the way
// back MAX_POOL_BACKWARD / CONVOLUTION_BACKWARD followed by an in-place RELU_BACKWARD on the gradient they wrote, masked by the map
// they read.  The host has no fusion for these pairs (ccv_nnc_ops_fusions[] in lib/nnc/ccv_nnc_symbolic_graph_simplify.c:595- holds
// softmax + crossentropy only), so each pair costs one more pass over the largest tensors of the step: 12.8 of VGG-D's 98.9 ms.
//
// What this does: such a command is not launched when it arrives but recorded (one slot per stream context).  If the NEXT command on
// that stream is the matching ReLU, the recorded command runs with the opt-in bit (NNC_MI355X_CONV_ALGO_FUSE_RELU /
// NNC_MI355X_POOL_ALGO_FUSE_RELU_BACKWARD, include/nnc_mi355x.h) and the ReLU is done; anything else -- another command on the
// stream, a signal, a wait, a copy, a free, a callback: every point that could observe the stream's order, the same hooks the
// recorded collectives use (cmd_comm.cpp) -- launches it as it was first.  Results are the same either way: max(0, .) and the
// a > 0 mask are exact.
//
// Only commands whose exact signature (parameters, hint, flags, every tensor's type / format / shape / strides, and which of its tensors
// share memory) has already run successfully on the spot are recorded, so a recorded command cannot fail on a parameter check later.  If it
// fails at launch all the same (out of memory): completed by its ReLU, the ReLU command returns the failure; launched by a flush, the failure
// is printed and kept, and the next recordable command of the process returns it (deferred_take_error) -- never a silent success.
// NNC_MI355X_PEEPHOLE=0 in the environment (or nnc_mi355x_set_peephole(0)) turns the look-ahead off.
//
// The TRAIL (round 5).  The reference's static schedule puts work BETWEEN a CONVOLUTION_BACKWARD and the RELU_BACKWARD that would complete it: behind the
// convolution it emits the signal the layer's two SGD_FORWARD commands (one stream each) wait for, issues those waits, the SGD commands and their own
// signals, and only then the ReLU (lib/nnc/ccv_nnc_graph_run.c:581-675 walks the schedule in topological order).  A flush at the first of those hooks lost
// every such pair: 95 of VGG-D's 218 recorded commands went out as they were, 4.4 ms of the step through the host.  So a recorded command now carries a
// trail: operations that arrived behind it and cannot be observed by anyone until something else is -- an EMIT on the recorded command's own stream (or on
// a stream already in the trail), a WAIT for a signal whose emit is in the trail (the waiting stream joins the trail), an SGD_FORWARD on a stream in the
// trail -- are kept, in arrival order, and replayed in that order right behind the recorded command when it finally launches, folded or not.  Everything
// the trail does not hold is decided as before: an operation on a stream outside the trail that names no signal of the trail cannot depend on it and runs
// at once; any other operation that orders against a stream of the trail launches the recorded command and replays the trail first.  Stream order and
// signal order are what the host's schedule is made of, and both are kept; the kernels, their streams and their events are the same, only the ReLU's pass
// over the gradient is gone.
//
// SGD BATCHES (round 6).  An SGD_FORWARD that arrives right behind another update of its stream (nothing launched in between: deferred_sgd_head) with nothing
// recorded in front of it becomes a recorded command of its own kind (DEFER_SGD_BATCH: the update itself is the first entry of the slot's trail), the updates
// that follow it on that stream join the trail, and when the slot is launched -- at the
// stream's next order-observing point, like every slot -- consecutive updates of one stream go out as ONE multi-tensor launch (cmd_ew.cpp sgd_forw_multi:
// bit-identical arithmetic).  The reference's models end a step with one SGD_FORWARD per parameter tensor (~200 for ResNet-50, 5 - 18 us each, most of it
// launch latency and one host enqueue each).  The same merge applies to the updates a CONVOLUTION_BACKWARD's trail holds.  NNC_MI355X_SGD_BATCH=0 turns it off.
//
// Threads.  Every order-observing hook flushes (stream_of, copies, frees, signals, callbacks), and some of them flush EVERY stream's slot from
// whatever thread called (a loader thread's host-to-device copy).  A slot being launched stays visible in state LAUNCHING -- still counted in
// g_deferred_live -- until its kernels have been enqueued; anything that would order itself against that stream (the matching ReLU, another
// command's stream_of, a flush of that context) waits for it on a condition variable instead of finding no slot and running ahead of it.
#include "common.h"
#include <mutex>
#include <condition_variable>
#include <unordered_set>
#include <cstdlib>
#include <cstring>
#include <unistd.h>

namespace nnc {

std::atomic<int> g_deferred_live(0);

namespace {

constexpr int MAX_IO = 6;
enum { FREE = 0, RECORDED = 1, LAUNCHING = 2 };
enum { OP_EMIT = 1, OP_WAIT = 2, OP_CMD = 3 };
constexpr int TRAIL_IO = 4;   // tensors per side of a trail command (SGD_FORWARD: 3 in, 2 out)
constexpr int TRAIL_MAX = 24; // operations behind one recorded command (VGG-D through the reference host: 7)
struct TrailOp {
	int op;
	ccv_nnc_stream_context_t* ctx;
	int device;
	const ccv_nnc_stream_signal_t* signal; // OP_EMIT / OP_WAIT
	exec_fn_t fn;                           // OP_CMD ...
	ccv_nnc_cmd_t cmd;
	ccv_nnc_hint_t hint;
	int flags, nin, nout;
	ccv_nnc_tensor_view_t in[TRAIL_IO], out[TRAIL_IO];
	int has_in[TRAIL_IO], has_out[TRAIL_IO];
};
struct Slot {
	int live; // FREE / RECORDED / LAUNCHING
	exec_fn_t fn;
	int kind;
	ccv_nnc_cmd_t cmd;
	ccv_nnc_hint_t hint;
	int flags;
	ccv_nnc_tensor_view_t in[MAX_IO], out[MAX_IO]; // (the host hands a backward command the forward's inputs and outputs too: up to 5 + 3)
	int has_in[MAX_IO], has_out[MAX_IO];
	int nin, nout;
	ccv_nnc_stream_context_t* ctx;
	int device;
	int ntrail;
	TrailOp trail[TRAIL_MAX];
};
constexpr int SLOTS = 16;
Slot g_slots[SLOTS];
// (never destroyed: a host thread the process does not join may still enqueue or flush while the exit handlers run; cmd_comm.cpp g_cliques)
std::recursive_mutex& g_mu = *new std::recursive_mutex;
std::condition_variable_any& g_launched = *new std::condition_variable_any; // a LAUNCHING slot became FREE
// a flushed command failed at launch: the next recordable command ON THAT STREAM (context, device) reports it -- not whichever command of the process
// comes next (a valid command on another stream or device used to be refused with somebody else's out-of-memory)
struct StickyError { const ccv_nnc_stream_context_t* ctx; int device; int err; };
constexpr int MAX_STICKY = 16;
StickyError g_sticky[MAX_STICKY];
std::atomic<int> g_sticky_live(0);
std::unordered_set<uint64_t>& g_good = *new std::unordered_set<uint64_t>;
int g_enabled = -1;
long g_recorded = 0, g_folded = 0, g_plain = 0; // nnc_mi355x_debug_peephole_counts
long g_trailed = 0;                             // operations that waited in a trail (nnc_mi355x_debug_peephole_trailed)
long g_sgd_batches = 0, g_sgd_batched = 0;      // multi-tensor launches, and the updates they carried (nnc_mi355x_debug_sgd_batches)
int g_debug_fail_trailed = 0;     // nnc_mi355x_debug_peephole_fail_trailed: the next command replayed out of a trail reports this instead of running (tests)
int g_debug_launch_delay_us = 0; // nnc_mi355x_debug_peephole_launch_delay_us: tests widen the window between a slot's release and its launch
thread_local int tl_running = 0; // inside a recorded command's launch: its own stream_of / nested commands must not touch the slots

struct StatsAtExit { // NNC_MI355X_PEEPHOLE_STATS=1: one line at unload -- how many pairs of a run actually folded
	~StatsAtExit()
	{
		const char* v = getenv("NNC_MI355X_PEEPHOLE_STATS");
		if (v && *v == '1') fprintf(stderr, "[nnc_mi355x] look-ahead: %ld commands recorded, %ld completed by their ReLU, %ld launched as they were\n", g_recorded, g_folded, g_plain);
	}
} g_stats_at_exit;

bool enabled()
{
	if (g_enabled < 0) {
		const char* v = getenv("NNC_MI355X_PEEPHOLE");
		g_enabled = (v && *v == '0') ? 0 : 1;
	}
	return g_enabled == 1;
}

void mix(uint64_t& h, const void* p, size_t n)
{
	const unsigned char* b = (const unsigned char*)p;
	for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 0x100000001b3ULL; }
}
void mix_tensor(uint64_t& h, const ccv_nnc_tensor_t* t)
{
	const int none = -1;
	if (!t) { mix(h, &none, sizeof(none)); return; }
	mix(h, &t->type, sizeof(t->type));
	mix(h, &t->info, sizeof(t->info));
	if (CCV_IS_TENSOR_VIEW(t)) {
		const ccv_nnc_tensor_view_t* v = (const ccv_nnc_tensor_view_t*)t;
		mix(h, &v->contiguous, sizeof(v->contiguous));
		mix(h, v->stride, sizeof(v->stride));
	}
	const uintptr_t align = (uintptr_t)t->data.u8 & 15; // kernels choose 16-byte paths by alignment
	mix(h, &align, sizeof(align));
}
uint64_t signature(const int kind, const ccv_nnc_cmd_t& cmd, const ccv_nnc_hint_t& hint, const int flags, ccv_nnc_tensor_t* const* inputs, const int nin, ccv_nnc_tensor_t* const* outputs, const int nout)
{
	uint64_t h = 0xcbf29ce484222325ULL;
	mix(h, &kind, sizeof(kind));
	mix(h, &cmd.cmd, sizeof(cmd.cmd));
	mix(h, &cmd.backend, sizeof(cmd.backend));
	mix(h, &cmd.algorithm, sizeof(cmd.algorithm));
	mix(h, &cmd.info, sizeof(cmd.info));
	mix(h, &hint, sizeof(hint));
	mix(h, &flags, sizeof(flags));
	mix(h, &nin, sizeof(nin));
	mix(h, &nout, sizeof(nout));
	for (int i = 0; i < nin; i++) mix_tensor(h, inputs[i]);
	for (int i = 0; i < nout; i++) mix_tensor(h, outputs[i]);
	// which tensors share memory: the exec functions validate aliasing (batch norm's running statistics in = out, cmd_norm.cpp; in-place
	// outputs), so a call with the same shapes but another aliasing pattern is a different call
	const int n = nin + nout;
	for (int i = 0; i < n; i++)
		for (int j = i + 1; j < n; j++) {
			const ccv_nnc_tensor_t* const a = i < nin ? inputs[i] : outputs[i - nin];
			const ccv_nnc_tensor_t* const b = j < nin ? inputs[j] : outputs[j - nin];
			const unsigned char same = a && b && a->data.u8 == b->data.u8;
			mix(h, &same, 1);
		}
	return h;
}

void keep(ccv_nnc_tensor_view_t* dst, int* has, const ccv_nnc_tensor_t* t)
{
	*has = t ? 1 : 0;
	if (!t) return;
	memset(dst, 0, sizeof(*dst));
	memcpy(dst, t, CCV_IS_TENSOR_VIEW(t) ? sizeof(ccv_nnc_tensor_view_t) : sizeof(ccv_nnc_tensor_t));
}

// launch a recorded command (relu_bit: with the ReLU folded in).  The slot is LAUNCHING until the launch has been enqueued: the launch's own
// hooks skip the slots (tl_running), every other thread that needs this stream's order waits (wait_launching).
typedef std::unique_lock<std::recursive_mutex> Lock;
// (the launch itself runs WITHOUT the slots' mutex: it takes the collectives' mutex through stream_of, and a thread recording a
// collective takes this one through the same hook -- never both at once in opposite orders)
struct Head { exec_fn_t fn; int kind; ccv_nnc_cmd_t cmd; ccv_nnc_hint_t hint; int flags; ccv_nnc_tensor_view_t in[MAX_IO], out[MAX_IO]; int has_in[MAX_IO], has_out[MAX_IO]; int nin, nout; ccv_nnc_stream_context_t* ctx; int device; };
void sticky(const ccv_nnc_stream_context_t* const ctx, const int device, const int r)
{ // no caller to hand a failed launch to: the next recordable command of that stream returns it (deferred_take_error)
	int at = -1;
	for (int i = 0; i < MAX_STICKY && at < 0; i++) if (g_sticky[i].err && g_sticky[i].ctx == ctx && g_sticky[i].device == device) at = i;
	for (int i = 0; i < MAX_STICKY && at < 0; i++) if (!g_sticky[i].err) at = i;
	if (at < 0) at = 0; // table full of unreported failures: the oldest gives way
	if (!g_sticky[at].err) ++g_sticky_live;
	g_sticky[at].ctx = ctx; g_sticky[at].device = device; g_sticky[at].err = r;
}
int run(Slot& s, const int relu_bit, Lock& lk, const bool report)
{
	Head c;
	c.fn = s.fn; c.kind = s.kind; c.cmd = s.cmd; c.hint = s.hint; c.flags = s.flags; c.nin = s.nin; c.nout = s.nout; c.ctx = s.ctx; c.device = s.device;
	memcpy(c.in, s.in, sizeof(c.in)); memcpy(c.out, s.out, sizeof(c.out)); memcpy(c.has_in, s.has_in, sizeof(c.has_in)); memcpy(c.has_out, s.has_out, sizeof(c.has_out));
	s.live = LAUNCHING; // (the trail stays in the slot: nobody appends to or reads a LAUNCHING slot but this thread)
	ccv_nnc_tensor_t* in[MAX_IO];
	ccv_nnc_tensor_t* out[MAX_IO];
	for (int i = 0; i < c.nin; i++) in[i] = c.has_in[i] ? (ccv_nnc_tensor_t*)&c.in[i] : 0;
	for (int i = 0; i < c.nout; i++) out[i] = c.has_out[i] ? (ccv_nnc_tensor_t*)&c.out[i] : 0;
	if (c.kind != DEFER_SGD_BATCH) { if (relu_bit) ++g_folded; else ++g_plain; } // (a batch of updates is not one of the ReLU pairs these count)
	if (relu_bit) c.cmd.algorithm = relu_bit | (c.cmd.algorithm < 0 ? 0xff : (c.cmd.algorithm & 0xff));
	int prev = 0;
	HIP_ENFORCE(hipGetDevice(&prev));
	if (prev != c.device) HIP_ENFORCE(hipSetDevice(c.device));
	++tl_running;
	lk.unlock();
	if (g_debug_launch_delay_us > 0) usleep(g_debug_launch_delay_us);
	const int r = c.fn ? c.fn(c.cmd, c.hint, c.flags, in, c.nin, out, c.nout, c.ctx) : CCV_NNC_EXEC_SUCCESS; // (an SGD batch has no head: its first update is trail[0])
	// the trail, in arrival order, right behind the command: its emits, the waits for them, the commands behind those waits
	for (int k = 0; k < s.ntrail; k++) {
		TrailOp& t = s.trail[k];
		int dev = 0;
		HIP_ENFORCE(hipGetDevice(&dev));
		if (dev != t.device) HIP_ENFORCE(hipSetDevice(t.device));
		if (t.op == OP_CMD && !g_debug_fail_trailed && sgd_is_exec(t.fn)) { // consecutive updates of one stream: one multi-tensor launch
			int e = k + 1;
			while (e < s.ntrail && s.trail[e].op == OP_CMD && s.trail[e].fn == t.fn && s.trail[e].ctx == t.ctx && s.trail[e].device == t.device) e++;
			if (e - k >= 2) {
				const ccv_nnc_cmd_t* cmds[TRAIL_MAX];
				ccv_nnc_tensor_t* tin[TRAIL_MAX][TRAIL_IO];
				ccv_nnc_tensor_t* tout[TRAIL_MAX][TRAIL_IO];
				ccv_nnc_tensor_t* const* pin[TRAIL_MAX];
				ccv_nnc_tensor_t* const* pout[TRAIL_MAX];
				for (int j = k; j < e; j++) {
					TrailOp& u = s.trail[j];
					cmds[j - k] = &u.cmd;
					for (int i = 0; i < u.nin; i++) tin[j - k][i] = u.has_in[i] ? (ccv_nnc_tensor_t*)&u.in[i] : 0;
					for (int i = 0; i < u.nout; i++) tout[j - k][i] = u.has_out[i] ? (ccv_nnc_tensor_t*)&u.out[i] : 0;
					pin[j - k] = tin[j - k]; pout[j - k] = tout[j - k];
				}
				if (sgd_forw_multi(cmds, pin, pout, e - k, t.ctx) == CCV_NNC_EXEC_SUCCESS) {
					lk.lock();
					++g_sgd_batches; g_sgd_batched += e - k;
					lk.unlock();
					k = e - 1;
					continue;
				}
			}
		}
		if (t.op == OP_EMIT) signal_emit_now(t.ctx, t.signal);
		else if (t.op == OP_WAIT) signal_wait_now(t.ctx, t.signal);
		else {
			ccv_nnc_tensor_t* tin[TRAIL_IO];
			ccv_nnc_tensor_t* tout[TRAIL_IO];
			for (int i = 0; i < t.nin; i++) tin[i] = t.has_in[i] ? (ccv_nnc_tensor_t*)&t.in[i] : 0;
			for (int i = 0; i < t.nout; i++) tout[i] = t.has_out[i] ? (ccv_nnc_tensor_t*)&t.out[i] : 0;
			int tr = g_debug_fail_trailed;
			if (tr) g_debug_fail_trailed = 0;
			else tr = t.fn(t.cmd, t.hint, t.flags, tin, t.nin, tout, t.nout, t.ctx);
			if (tr != CCV_NNC_EXEC_SUCCESS) {
				fprintf(stderr, "[nnc_mi355x] a command (0x%x) kept behind a recorded one failed at launch with %d after its caller was told it had been enqueued\n", t.cmd.cmd, tr);
				lk.lock();
				sticky(t.ctx, t.device, tr);
				lk.unlock();
			}
		}
	}
	lk.lock();
	--tl_running;
	s.ntrail = 0;
	s.live = FREE;
	--g_deferred_live;
	g_launched.notify_all();
	int now = prev;
	HIP_ENFORCE(hipGetDevice(&now));
	if (now != prev) HIP_ENFORCE(hipSetDevice(prev)); // (binding a fixed-device stream sets the device: the caller's stays what it was)
	if (r != CCV_NNC_EXEC_SUCCESS) {
		fprintf(stderr, "[nnc_mi355x] a recorded command (0x%x) failed at launch with %d after its caller was told it had been enqueued\n", c.cmd.cmd, r);
		if (!report) sticky(c.ctx, c.device, r);
	}
	return r;
}

bool in_trail(const Slot& s, const ccv_nnc_stream_context_t* const ctx)
{
	for (int k = 0; k < s.ntrail; k++) if (s.trail[k].ctx == ctx) return true;
	return false;
}
bool orders_against(const Slot& s, const ccv_nnc_stream_context_t* const ctx)
{ // the default stream orders against every other: no context = all; a stream with operations in the slot's trail orders against the slot
	return !ctx || !s.ctx || s.ctx == ctx || in_trail(s, ctx);
}

// another thread is enqueueing a recorded command whose stream `ctx` orders against: wait until it is in the stream
void wait_launching(const ccv_nnc_stream_context_t* const ctx, Lock& lk)
{
	for (;;) {
		bool busy = false;
		for (int i = 0; i < SLOTS && !busy; i++)
			busy = g_slots[i].live == LAUNCHING && orders_against(g_slots[i], ctx);
		if (!busy) return;
		g_launched.wait(lk);
	}
}

// The device a command on this stream runs on: the stream's own (a fixed-device context binds it -- the host does not set the current
// device before a command that carries a stream, device_rt.cpp bind()), or the current one (no stream / any-device contexts).
int device_for(const ccv_nnc_stream_context_t* ctx)
{
	if (ctx && CCV_STREAM_GET_CONTEXT(ctx->type) == CCV_STREAM_CONTEXT_GPU) return ccv_nnc_stream_context_get_device(ctx);
	int device = 0;
	HIP_ENFORCE(hipGetDevice(&device));
	return device;
}

Slot* slot_of(const ccv_nnc_stream_context_t* ctx, const int device)
{
	for (int i = 0; i < SLOTS; i++)
		if (g_slots[i].live == RECORDED && g_slots[i].ctx == ctx && g_slots[i].device == device) return &g_slots[i];
	return 0;
}

bool same_buffer(const ccv_nnc_tensor_view_t& kept, const ccv_nnc_tensor_t* t)
{
	// the ReLU runs over dense tensors (cmd_ew.cpp) in place: same memory, same element count and type as what the recorded command wrote / read
	return t && t->data.u8 == kept.data.u8 && t->info.datatype == kept.info.datatype && tensor_count(t->info) == tensor_count(kept.info) && tensor_contiguous(t) && tensor_contiguous((const ccv_nnc_tensor_t*)&kept);
}

} // namespace

bool deferred_try(exec_fn_t fn, const int kind, const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const ctx, uint64_t* const sig)
{
	*sig = 0;
	if (tl_running || !enabled() || input_size > MAX_IO || output_size > MAX_IO || output_size < 1 || !outputs[0] || (flags & CCV_NNC_ACCUMULATE_OUTPUT)) return false;
	if (cmd.algorithm > 0 && (cmd.algorithm & ~0xff)) return false; // the caller set the bit itself
	static const int kinds = getenv("NNC_MI355X_PEEPHOLE_KINDS") ? atoi(getenv("NNC_MI355X_PEEPHOLE_KINDS")) : ~0; // debugging aid: bit k = kind k may be recorded
	if (!(kinds & (1 << kind))) return false;
	if (CCV_TENSOR_GET_MEMORY(outputs[0]->info.type) != CCV_TENSOR_GPU_MEMORY) return false;
	const uint64_t h = signature(kind, cmd, hint, flags, inputs, input_size, outputs, output_size);
	*sig = h;
	Lock lock(g_mu);
	if (!g_good.count(h)) return false; // first time: run on the spot, deferred_mark_good() files it when it succeeds
	const int device = device_for(ctx);
	wait_launching(ctx, lock);
	if (Slot* const old = slot_of(ctx, device)) run(*old, 0, lock, false); // two recordable commands in a row: the first goes as it is
	for (int i = 0; i < SLOTS; i++) // this stream has operations in another recorded command's trail: they go first
		if (g_slots[i].live == RECORDED && in_trail(g_slots[i], ctx)) run(g_slots[i], 0, lock, false);
	Slot* s = 0;
	for (int i = 0; i < SLOTS && !s; i++)
		if (g_slots[i].live == FREE) s = &g_slots[i];
	if (!s) return false;
	s->fn = fn; s->kind = kind; s->cmd = cmd; s->hint = hint; s->flags = flags; s->ctx = ctx; s->device = device;
	s->nin = input_size; s->nout = output_size; s->ntrail = 0;
	for (int i = 0; i < input_size; i++) keep(&s->in[i], &s->has_in[i], inputs[i]);
	for (int i = 0; i < output_size; i++) keep(&s->out[i], &s->has_out[i], outputs[i]);
	s->live = RECORDED;
	++g_deferred_live;
	++g_recorded;
	return true;
}

void deferred_mark_good(const uint64_t sig)
{
	if (!sig) return;
	std::lock_guard<std::recursive_mutex> lock(g_mu);
	if (g_good.size() < 65536) g_good.insert(sig);
}

int deferred_fuse_relu_forw(const ccv_nnc_tensor_t* const a, ccv_nnc_tensor_t* const b, ccv_nnc_stream_context_t* const ctx)
{
	if (!g_deferred_live || tl_running) return -1;
	Lock lock(g_mu);
	const int device = device_for(ctx);
	wait_launching(ctx, lock); // (a foreign thread's flush is enqueueing this stream's recorded command: the ReLU goes behind it, unfolded)
	Slot* const s = slot_of(ctx, device);
	if (!s || (s->kind != DEFER_CONV_FORWARD && s->kind != DEFER_BNORM_FORWARD && s->kind != DEFER_EWSUM_FORWARD) || !s->has_out[0] || a->data.u8 != b->data.u8 || !same_buffer(s->out[0], b) || !same_buffer(s->out[0], a)) return -1;
	return run(*s, s->kind == DEFER_CONV_FORWARD ? NNC_MI355X_CONV_ALGO_FUSE_RELU : (s->kind == DEFER_BNORM_FORWARD ? NNC_MI355X_BNORM_ALGO_FUSE_RELU : NNC_MI355X_EWSUM_ALGO_FUSE_RELU), lock, true);
}

int deferred_fuse_relu_back(const ccv_nnc_tensor_t* const g, const ccv_nnc_tensor_t* const b, ccv_nnc_tensor_t* const h, ccv_nnc_stream_context_t* const ctx)
{
	if (!g_deferred_live || tl_running || !g) return -1;
	Lock lock(g_mu);
	const int device = device_for(ctx);
	wait_launching(ctx, lock);
	Slot* const s = slot_of(ctx, device);
	if (s && s->kind == DEFER_EWSUM_FORWARD) {
		// the gradient sum at the head of a residual block, then RELU_BACKWARD (g, -, b) -> h in place on that sum: the sum masks as it stores.  The mask map b
		// joins the recorded command as its last input (NNC_MI355X_EWSUM_ALGO_FUSE_RELU_BACKWARD, cmd_ew.cpp); two summands, none of them the mask or the output's alias of it
		if (s->nin != 2 || !s->has_in[0] || !s->has_in[1] || !s->has_out[0] || g->data.u8 != h->data.u8 || !same_buffer(s->out[0], h)) return -1;
		if (!b || !tensor_contiguous(b) || b->info.datatype != h->info.datatype || tensor_count(b->info) != tensor_count(h->info) || b->data.u8 == h->data.u8 || CCV_TENSOR_GET_MEMORY(b->info.type) != CCV_TENSOR_GPU_MEMORY) return -1;
		keep(&s->in[2], &s->has_in[2], b);
		s->nin = 3;
		return run(*s, NNC_MI355X_EWSUM_ALGO_FUSE_RELU_BACKWARD, lock, true);
	}
	if (!s || (s->kind != DEFER_CONV_BACKWARD && s->kind != DEFER_POOL_BACKWARD) || !s->has_out[0] || s->nin < 2 || !s->has_in[1]) return -1;
	if (s->out[0].info.datatype != s->in[1].info.datatype) return -1; // the folded epilogue masks h by a in one element type (conv_back_entry refuses a mix)
	// RELU_BACKWARD (g, -, b) -> h in place on the gradient the recorded command writes, b the map the recorded command read as its input a
	if (g->data.u8 != h->data.u8 || !same_buffer(s->out[0], h) || !same_buffer(s->in[1], b) || s->in[1].info.format != s->out[0].info.format) return -1;
	return run(*s, s->kind == DEFER_CONV_BACKWARD ? NNC_MI355X_CONV_ALGO_FUSE_RELU : NNC_MI355X_POOL_ALGO_FUSE_RELU_BACKWARD, lock, true);
}

// ---- the trail (see the head of this file)
namespace {
Slot* slot_with_stream(const ccv_nnc_stream_context_t* const ctx, const int device)
{ // the RECORDED slot whose own stream this is, or whose trail holds operations of it
	if (!ctx) return 0;
	for (int i = 0; i < SLOTS; i++) {
		Slot& s = g_slots[i];
		if (s.live != RECORDED) continue;
		if (s.ctx == ctx && s.device == device) return &s;
		for (int k = 0; k < s.ntrail; k++) if (s.trail[k].ctx == ctx && s.trail[k].device == device) return &s;
	}
	return 0;
}
Slot* slot_with_signal(const ccv_nnc_stream_signal_t* const signal)
{
	for (int i = 0; i < SLOTS; i++) {
		Slot& s = g_slots[i];
		if (s.live != RECORDED) continue;
		for (int k = 0; k < s.ntrail; k++) if (s.trail[k].op != OP_CMD && s.trail[k].signal == signal) return &s;
	}
	return 0;
}
// another thread is replaying a trail that names `signal` (without the slots' mutex): until it is through, the signal's emit may not have been recorded in
// its stream yet -- a wait issued now, on a stream the trail does not hold, would find an event with nothing behind it and let that stream run ahead of the
// recorded command (wait_launching() above only knows streams)
void wait_launching_signal(const ccv_nnc_stream_signal_t* const signal, Lock& lk)
{
	for (;;) {
		bool busy = false;
		for (int i = 0; i < SLOTS && !busy; i++) {
			const Slot& s = g_slots[i];
			if (s.live != LAUNCHING) continue;
			for (int k = 0; k < s.ntrail && !busy; k++) busy = s.trail[k].op != OP_CMD && s.trail[k].signal == signal;
		}
		if (!busy) return;
		g_launched.wait(lk);
	}
}
bool trailing_on() { static const int on = !(getenv("NNC_MI355X_PEEPHOLE_TRAIL") && *getenv("NNC_MI355X_PEEPHOLE_TRAIL") == '0'); return on; }
}

// ccv_nnc_stream_compat_emit_signal / _wait_signal (device_rt.cpp) ask here first.  true: the operation waits in a trail (it will run behind the recorded
// command it follows); false: the caller performs it now -- whatever had to go first has been launched.
bool deferred_signal_op(const int emit, const ccv_nnc_stream_context_t* const ctx, const ccv_nnc_stream_signal_t* const signal)
{
	if (!g_deferred_live || tl_running) return false;
	Lock lock(g_mu);
	if (!ctx || !trailing_on()) { wait_launching_signal(signal, lock); lock.unlock(); deferred_flush(0); return false; }
	const int device = device_for(ctx);
	wait_launching(ctx, lock);
	wait_launching_signal(signal, lock);
	Slot* const by_stream = slot_with_stream(ctx, device);
	Slot* const by_signal = slot_with_signal(signal);
	if (!by_stream && !by_signal) return false; // neither the stream nor the signal has anything to do with a recorded command
	Slot* const s = by_stream ? by_stream : by_signal;
	// an emit on a stream outside every trail that re-records a signal of a trail, an operation that would tie two recorded commands together, a full
	// trail: decided the old way -- launch what is involved, then do it now
	// (a WAIT on the recorded command's OWN stream is decided the old way too: the ReLU that would complete the command comes behind that wait and may need it)
	const bool own_wait = !emit && by_stream && by_stream->ctx == ctx && by_stream->device == device;
	if ((emit && !by_stream) || own_wait || (by_stream && by_signal && by_stream != by_signal) || s->ntrail >= TRAIL_MAX) {
		if (by_stream) run(*by_stream, 0, lock, false);
		if (by_signal && by_signal != by_stream && by_signal->live == RECORDED) run(*by_signal, 0, lock, false);
		return false;
	}
	TrailOp& t = s->trail[s->ntrail++];
	t.op = emit ? OP_EMIT : OP_WAIT;
	t.ctx = (ccv_nnc_stream_context_t*)ctx; t.device = device; t.signal = signal;
	++g_trailed;
	return true;
}

// A command that may wait in a trail (SGD_FORWARD: its parameters have been checked by the caller, it allocates nothing) on a stream that already has
// operations in one: kept behind them.  false: the caller launches it now (its stream_of flushes whatever orders against the stream).
bool deferred_trail_cmd(exec_fn_t fn, const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const ctx)
{
	if (!g_deferred_live || tl_running || !ctx || input_size > TRAIL_IO || output_size > TRAIL_IO || !trailing_on()) return false;
	Lock lock(g_mu);
	const int device = device_for(ctx);
	wait_launching(ctx, lock);
	Slot* const s = slot_with_stream(ctx, device);
	if (!s || (s->ctx == ctx && s->device == device && s->kind != DEFER_SGD_BATCH) || s->ntrail >= TRAIL_MAX) return false; // (on the recorded command's own stream only its ReLU may follow -- or, behind a recorded update, more updates)
	for (int i = 0; i < SLOTS; i++) // ... and in no other trail
		if (&g_slots[i] != s && g_slots[i].live == RECORDED && in_trail(g_slots[i], ctx)) return false;
	TrailOp& t = s->trail[s->ntrail++];
	t.op = OP_CMD; t.ctx = ctx; t.device = device; t.signal = 0;
	t.fn = fn; t.cmd = cmd; t.hint = hint; t.flags = flags; t.nin = input_size; t.nout = output_size;
	for (int i = 0; i < input_size; i++) keep(&t.in[i], &t.has_in[i], inputs[i]);
	for (int i = 0; i < output_size; i++) keep(&t.out[i], &t.has_out[i], outputs[i]);
	++g_trailed;
	return true;
}

// Which updates start a batch: only one that arrives RIGHT BEHIND another update of the same stream launched by the same thread, with nothing launched through
// the library in between -- the signature of a caller that issues its updates back to back (ccv_amd/vgg.py, a hand-written trainer).  The reference host's
// scheduler spreads a model's updates over many streams, each between a wait and an emit of its own: no two ever meet in a batch there, and recording every one
// of them anyway cost what the first form of this did on ResNet-50 -- all 16 slots held by lone updates, 830 of 2 435 foldable commands per run no longer
// recorded (the EWSUM + RELU_BACKWARD folds gone: +1.1 ms per step), 1.3 ms more host time per step (profiles/r06_v10_resnet50-nchw-bs256-f16_kernel_stats.md).
static thread_local const ccv_nnc_stream_context_t* tl_last_sgd_ctx = 0;
static thread_local unsigned long tl_last_sgd_seq = 0;
void deferred_sgd_launched(const ccv_nnc_stream_context_t* const ctx) { tl_last_sgd_ctx = ctx; tl_last_sgd_seq = g_launch_seq.load(std::memory_order_relaxed); }
// An SGD_FORWARD (parameters checked by the caller) on a stream with nothing recorded: it starts a batch (see the head of this file).  false: the caller launches it now.
bool deferred_sgd_head(exec_fn_t fn, const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const ctx)
{
	static const int on = !(getenv("NNC_MI355X_SGD_BATCH") && *getenv("NNC_MI355X_SGD_BATCH") == '0');
	if (!on || tl_running || !enabled() || !ctx || CCV_STREAM_GET_CONTEXT(ctx->type) != CCV_STREAM_CONTEXT_GPU || input_size > TRAIL_IO || output_size > TRAIL_IO || !trailing_on()) return false;
	if (tl_last_sgd_ctx != ctx || tl_last_sgd_seq != g_launch_seq.load(std::memory_order_relaxed)) return false; // not back to back (see above)
	Lock lock(g_mu);
	const int device = device_for(ctx);
	wait_launching(ctx, lock);
	if (slot_with_stream(ctx, device)) return false; // (deferred_trail_cmd has already declined to put it behind what is recorded there: the caller's stream_of launches that first)
	Slot* s = 0;
	for (int i = 0; i < SLOTS && !s; i++)
		if (g_slots[i].live == FREE) s = &g_slots[i];
	if (!s) return false;
	s->fn = 0; s->kind = DEFER_SGD_BATCH; s->cmd = cmd; s->hint = hint; s->flags = flags; s->ctx = ctx; s->device = device;
	s->nin = 0; s->nout = 0; s->ntrail = 1;
	TrailOp& t = s->trail[0];
	t.op = OP_CMD; t.ctx = ctx; t.device = device; t.signal = 0;
	t.fn = fn; t.cmd = cmd; t.hint = hint; t.flags = flags; t.nin = input_size; t.nout = output_size;
	for (int i = 0; i < input_size; i++) keep(&t.in[i], &t.has_in[i], inputs[i]);
	for (int i = 0; i < output_size; i++) keep(&t.out[i], &t.has_out[i], outputs[i]);
	s->live = RECORDED;
	++g_deferred_live;
	++g_trailed;
	return true;
}

void deferred_flush(const ccv_nnc_stream_context_t* const ctx)
{
	if (tl_running) return;
	Lock lock(g_mu);
	for (int i = 0; i < SLOTS; i++)
		if (g_slots[i].live == RECORDED && orders_against(g_slots[i], ctx)) run(g_slots[i], 0, lock, false);
	wait_launching(ctx, lock); // what another thread is enqueueing right now is part of the order this caller is about to observe
}

int deferred_take_error(const ccv_nnc_stream_context_t* const ctx)
{
	if (!g_sticky_live || tl_running) return 0;
	Lock lock(g_mu);
	const int device = device_for(ctx);
	for (int i = 0; i < MAX_STICKY; i++)
		if (g_sticky[i].err && g_sticky[i].ctx == ctx && g_sticky[i].device == device) {
			const int e = g_sticky[i].err;
			g_sticky[i].err = 0;
			--g_sticky_live;
			return e;
		}
	return 0;
}

// Commands run underneath while this is raised are never recorded (and their stream hooks leave the slots alone): half_stage.cpp runs a command on fp32
// images that live in a staging arena and converts them back right behind it -- a recorded command there would have to be launched at once anyway, and a
// failure of that launch had no caller to go to.
void deferred_suppress(const int delta) { tl_running += delta; }

} // namespace nnc

extern "C" void nnc_mi355x_set_peephole(const int on)
{
	nnc::deferred_flush(0);
	nnc::g_enabled = on ? 1 : 0;
}

extern "C" long nnc_mi355x_debug_peephole_trailed(void)
{
	std::lock_guard<std::recursive_mutex> lock(nnc::g_mu);
	return nnc::g_trailed;
}

extern "C" void nnc_mi355x_debug_sgd_batches(long* const launches, long* const updates)
{
	std::lock_guard<std::recursive_mutex> lock(nnc::g_mu);
	if (launches) *launches = nnc::g_sgd_batches;
	if (updates) *updates = nnc::g_sgd_batched;
}

extern "C" void nnc_mi355x_debug_peephole_launch_delay_us(const int us) { nnc::g_debug_launch_delay_us = us; }
extern "C" void nnc_mi355x_debug_peephole_fail_trailed(const int err) { nnc::g_debug_fail_trailed = err; }

extern "C" void nnc_mi355x_debug_peephole_counts(long* const recorded, long* const folded, long* const plain)
{
	std::lock_guard<std::recursive_mutex> lock(nnc::g_mu);
	if (recorded) *recorded = nnc::g_recorded;
	if (folded) *folded = nnc::g_folded;
	if (plain) *plain = nnc::g_plain;
}
