// CCV_NNC_LSTM_FORWARD / BACKWARD on gfx950 (the recurrent row of the NLP trainers, test/int/nnc/lstm.tests.c, imdb.tests.c:1278).
// Replaces lib/nnc/cmd/rnn/gpu/ccv_nnc_lstm_gpu_cudnn.cu:50-249 -- cudnnRNNForward / cudnnRNNBackwardData_v8 / cudnnRNNBackwardWeights_v8 with
// CUDNN_LSTM, CUDNN_LINEAR_INPUT, CUDNN_RNN_DOUBLE_BIAS / NO_BIAS, CUDNN_RNN_PADDED_IO_ENABLED, uni- or bidirectional, optional recurrent projection.
// The reference has NO CPU implementation of this command (lib/nnc/cmd/rnn/ccv_nnc_lstm_cpu_ref.c is empty) and its tests assert no values, so the
// semantics are cuDNN's published ones (oracle/lstm_numpy.py restates them; "parity unpinned" there):
//   i = sigmoid(W_i x + R_i h' + bW_i + bR_i)   f = sigmoid(W_f ..)   g = tanh(W_g ..)   o = sigmoid(W_o ..)      (h', c': the state before the step)
//   c = f c' + i g      h = o tanh(c)           with a projection (proj_size != hidden_size): h = W_p (o tanh(c))
// weight space (the host sizes it, lstm.tests.c:14-21): every pseudo-layer's matrices first -- layer-major, direction inside; W_i W_f W_g W_o (each H x in),
// R_i R_f R_g R_o (each H x P), then W_p (P x H) -- and then every pseudo-layer's biases, bW_i .. bW_o, bR_i .. bR_o (each H).
// Sequences shorter than the longest (input 1, an int32 tensor in host memory, one length per batch item): a step past the end hands the state on
// unchanged and writes zeros to y; the backward direction starts at each item's own last step.  Dropout (training only) scales what one layer hands to
// the next; the draw is a counter hash like DROPOUT_FORWARD's (cmd_ew.cpp), the scales are kept in the reserved space.
//
// MI355X shape of the work: per pseudo-layer ONE contraction on the matrix cores for the input half of every step (X W^T, all T x B rows), then per
// step one launch that does the recurrent half (h' R^T against a k-major copy of R: lanes along the hidden units, loads coalesced, a tile of 16 batch
// rows per workgroup so that R is read B / 16 times per step) WITH the gate arithmetic fused behind it; backward: per step an element-wise gate
// gradient and the same small product against R, then three contractions per pseudo-layer (dX = dG W, dW = dG^T X, dR = dG^T H') and a column sum.
// Reserved space (floats, H wide; fits the reference tests' sizing, lstm.tests.c:23-34): per pseudo-layer and step S = 5 (+1 projection, +1 dropout)
// planes of B x H -- i, f, g, o, tanh(c), [projected h], [dropout scale] --, then the cell state after each of the first T - 1 steps.
#include <optional>
#include "gemm_launch.h"
#include "isa.h"

using namespace nnc;

namespace {

#define EXEC_ARGS_L const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context

struct lstm_geom_t {
	int T, B, I, H, P, L, D, bias, batch_first, is_test, proj, S;
	float dropout;
	size_t BH() const { return (size_t)B * H; }
	int in_of(int l) const { return l == 0 ? I : D * P; }
	size_t mats_of(int l) const { return (size_t)4 * H * in_of(l) + (size_t)4 * H * P + (proj ? (size_t)P * H : 0); }
	size_t w_off(int p) const { size_t o = 0; for (int q = 0; q < p; q++) o += mats_of(q / D); return o; } // the matrices of pseudo-layer p
	size_t b_off(int p) const { return w_off(L * D) + (size_t)p * 8 * H; }
	size_t w_total() const { return w_off(L * D) + (bias ? (size_t)L * D * 8 * H : 0); }
	size_t slot(int p, int s, int k) const { return (((size_t)p * T + s) * S + k) * BH(); }
	size_t cslot(int p, int s) const { return (size_t)L * D * T * S * BH() + ((size_t)p * (T - 1) + s) * BH(); }
	size_t reserve_total() const { return (size_t)L * D * BH() * ((size_t)T * S + (T - 1)); }
};

static bool lstm_geometry(const ccv_nnc_cmd_t cmd, const ccv_nnc_tensor_t* const x, lstm_geom_t* const g)
{
	const int nd = tensor_nd(x->info.dim);
	if (nd != 2 && nd != 3) return false;
	g->batch_first = nd == 3 && cmd.info.rnn.batch_first;
	g->B = nd == 3 ? (g->batch_first ? x->info.dim[0] : x->info.dim[1]) : 1;
	g->T = nd == 3 ? (g->batch_first ? x->info.dim[1] : x->info.dim[0]) : x->info.dim[0];
	g->I = x->info.dim[nd - 1];
	g->H = cmd.info.rnn.hidden_size;
	g->P = cmd.info.rnn.proj_size == 0 ? g->H : cmd.info.rnn.proj_size;
	g->L = cmd.info.rnn.num_layers;
	g->D = cmd.info.rnn.bidirectional ? 2 : 1;
	g->bias = !!cmd.info.rnn.bias;
	g->is_test = !!cmd.info.rnn.is_test;
	g->dropout = g->is_test || g->L < 2 ? 0.f : cmd.info.rnn.dropout;
	g->proj = g->P != g->H;
	g->S = 5 + g->proj + (g->dropout > 0.f ? 1 : 0);
	return g->T > 0 && g->B > 0 && g->I > 0 && g->H > 0 && g->P > 0 && g->P <= g->H && g->L > 0 && g->dropout >= 0.f && g->dropout < 1.f;
}

static bool dense_f32(const ccv_nnc_tensor_t* const t, const size_t least)
{
	return !t || (CCV_GET_DATA_TYPE(t->info.datatype) == CCV_32F && tensor_contiguous(t) && tensor_count(t->info) >= least);
}

// the reserved space: fp32 planes inside a dense fp32 tensor, or inside the CCV_16F tensor the host sized for a half-precision command (twice the elements:
// half_stage.cpp hands it over untouched)
static bool dense_reserve(const ccv_nnc_tensor_t* const t, const size_t least)
{
	if (!t || !tensor_contiguous(t)) return false;
	const int dt = CCV_GET_DATA_TYPE(t->info.datatype);
	return (dt == CCV_32F && tensor_count(t->info) >= least) || (dt == CCV_16F && tensor_count(t->info) >= 2 * least && ((uintptr_t)t->data.u8 & 15) == 0);
}

constexpr int LS_ROWS = 16; // batch rows per workgroup (4 groups of 4: a lane keeps 4 rows of its column in registers)
constexpr int LS_KT = 64;   // the reduction index is staged through LDS in tiles of this many

__device__ __forceinline__ float lstm_sigmoid(const float x) { return 1.f / (1.f + expf(-x)); }
// The rows kernels' step is a chain of dependent latencies (LDS round trips, barriers, the gate arithmetic): there the activations are the hardware's 2^x and
// 1 / x (1 ulp each; sigmoid within ~2e-7, tanh = 2 sigmoid(2x) - 1 within ~2.5e-7 absolute; saturating to 0 / 1 / -1 through inf and 0 without a branch).
__device__ __forceinline__ float lstm_fast_sigmoid(const float x) { return nnc_fast_rcp(1.f + nnc_fast_exp2(-1.4426950408889634f * x)); }
__device__ __forceinline__ float lstm_fast_tanh(const float x) { return 2.f * nnc_fast_rcp(1.f + nnc_fast_exp2(-2.8853900817779268f * x)) - 1.f; }

// acc[g][r] = sum_k a[(row0 + 4 * grp + r) * lda + k] * mt[k * ldm + g * gstride + col] for NG column groups: the product both step kernels share.
template <int NG>
__device__ __forceinline__ void lstm_rows_times(const float* const a, const int lda, const float* const mt, const int ldm, const int gstride, const int K, const int B, const int row0, const int col, const bool col_ok, float (&tile)[LS_ROWS][LS_KT], float (&acc)[NG][4])
{
	const int tid = threadIdx.x, grp = tid >> 6;
#pragma unroll
	for (int g = 0; g < NG; g++)
#pragma unroll
		for (int r = 0; r < 4; r++) acc[g][r] = 0.f;
	for (int k0 = 0; k0 < K; k0 += LS_KT) {
		__syncthreads();
		for (int e = tid; e < LS_ROWS * LS_KT; e += 256) {
			const int r = e / LS_KT, kk = e % LS_KT;
			tile[r][kk] = (row0 + r < B && k0 + kk < K) ? a[(size_t)(row0 + r) * lda + k0 + kk] : 0.f;
		}
		__syncthreads();
		if (!col_ok) continue;
		const int kn = K - k0 < LS_KT ? K - k0 : LS_KT;
		for (int kk = 0; kk < kn; kk++) {
			const float* const m = mt + (size_t)(k0 + kk) * ldm + col;
			float v[NG];
#pragma unroll
			for (int g = 0; g < NG; g++) v[g] = m[(size_t)g * gstride];
#pragma unroll
			for (int r = 0; r < 4; r++) {
				const float h = tile[grp * 4 + r][kk];
#pragma unroll
				for (int g = 0; g < NG; g++) acc[g][r] += v[g] * h;
			}
		}
	}
}

// ---- the whole sequence of one pseudo-layer in ONE launch (TUNE_LSTM_PERSISTENT; no projection, hidden size <= 512, the grid within the CU count) ----
// A workgroup owns 16 hidden units (their 64 gate columns) of a 16-row batch tile for all T steps.  Its slice of R -- 64 columns x P -- never leaves the
// register file, as the B fragments of v_mfma_f32_16x16x4_f32: wave q holds gate q's 16 columns, P / 4 registers per lane.  Per step: the tile's state h' [16][P] arrives in LDS, every
// wave forms its gate's 16 x 16 block with P / 4 matrix instructions (the A fragment one LDS word per lane and instruction), the four gates meet in LDS, thread (row, unit) adds the input half, does the gate arithmetic, keeps c in a register and
// PUBLISHES h: one 8-byte {step tag, value} word per element written with one agent-scope store -- the data is the flag (isa.h) --, which the tile's other
// workgroups poll while they fill their LDS image for the next step.  Two tag-parity planes suffice: a workgroup can only publish step s + 2 after it has read
// every word of step s + 1, whose writers had all finished reading step s.  The words are zeroed before the launch (tags start at 1).  A poll that gives up
// (CLUSTER_SPIN_LIMIT: the workgroups were not resident together) raises the stream's timeout word like cmd_norm.cpp's cluster kernels; the next synchronise stops the process.
struct lstm_seq_t {
	const float* gx; const float* r; const float* bw; const float* hx; const float* cx;
	float* y; float* hy; float* cy; float* rsv; unsigned long long* xch; const int* lens; unsigned* timeout_word;
	int T, B, H, dir, ldy, S;
	size_t slot0, cslot0; // of this pseudo-layer in the reserved space (floats)
};
#define LSTM_SEQ_LDS(KPT) (sizeof(float) * (16 * (4 * (KPT) + 4) + 4 * 16 * 65 + 4))
template <int KPT>
__global__ void __launch_bounds__(256) lstm_seq_forw_kernel(const lstm_seq_t a)
{
	constexpr int KT = 4 * KPT, PITCH = KT + 4;
	HIP_DYNAMIC_SHARED(float, lds) // (dynamic: the emulator keeps several of these workgroups resident, and only this form is per workgroup there)
	float* const htile = lds;                                        // [16][PITCH]
	float (*const part)[16][65] = (float (*)[16][65])(lds + 16 * PITCH); // [4][16][65]
	int& dead = *(int*)(lds + 16 * PITCH + 4 * 16 * 65);
	const int tid = threadIdx.x, c = tid & 63, kq = tid >> 6;
	const int H = a.H, B = a.B, j0 = blockIdx.x * 16, row0 = blockIdx.y * 16;
	const size_t BH = (size_t)B * H;
	float rreg[KPT]; // wave kq = gate kq: the B fragments of v_mfma_f32_16x16x4_f32 -- lane (unit = l & 15, k = 4 kk + (l >> 4)) -- of R_gate[j0 .. j0 + 15][0 .. P)
	{
		const int n = kq * H + j0 + (c & 15);
		const bool on = j0 + (c & 15) < H;
#pragma unroll
		for (int i = 0; i < KPT; i++) { const int k = 4 * i + (c >> 4); rreg[i] = on && k < H ? a.r[(size_t)n * H + k] : 0.f; }
	}
	const int rr = tid >> 4, u = tid & 15, b = row0 + rr, j = j0 + u;
	const bool mine = b < B && j < H;
	const size_t e = (size_t)b * H + j;
	float cst = mine && a.cx ? a.cx[e] : 0.f, hst = mine && a.hx ? a.hx[e] : 0.f;
	const int len = mine && a.lens ? a.lens[b] : a.T;
	float bias[4] = { 0.f, 0.f, 0.f, 0.f };
	if (mine && a.bw)
#pragma unroll
		for (int g = 0; g < 4; g++) bias[g] = a.bw[g * H + j] + a.bw[4 * H + g * H + j];
	if (tid == 0) dead = 0;
	__syncthreads();
	for (int s = 0; s < a.T; s++) {
		const int t = a.dir ? a.T - 1 - s : s;
		float gin[4] = { 0.f, 0.f, 0.f, 0.f };
		if (mine) { // (issued before the wait: the input half does not depend on the other workgroups)
			const float* const gr = a.gx + ((size_t)t * B + b) * 4 * H + j;
#pragma unroll
			for (int g = 0; g < 4; g++) gin[g] = gr[g * H];
		}
		// the tile's state before this step: hx at the first step, the words the tile's workgroups published at the step before otherwise
		for (int q0 = tid; q0 < 16 * H; q0 += 256 * 8) { // eight words in flight per thread, then the ones that had not arrived yet again
			unsigned long long gr[8];
			const unsigned long long* w[8];
#pragma unroll
			for (int i = 0; i < 8; i++) {
				const int q = q0 + 256 * i, r2 = q / H, k = q - r2 * H, b2 = row0 + r2;
				const bool on = q < 16 * H && b2 < B;
				w[i] = a.xch + ((size_t)((s + 1) & 1) * B + (on ? b2 : 0)) * H + (on ? k : 0);
				gr[i] = (unsigned long long)(unsigned)s << 32; // (rows past the batch: "arrived", zero)
				if (on) gr[i] = s == 0 ? (unsigned long long)__float_as_uint(a.hx ? a.hx[(size_t)b2 * H + k] : 0.f) : nnc_load_granule(w[i]);
			}
#pragma unroll
			for (int i = 0; i < 8; i++) {
				unsigned spins = 0;
				while ((unsigned)(gr[i] >> 32) != (unsigned)s) {
					if (*(volatile int*)&dead) break;
					if (++spins > CLUSTER_SPIN_LIMIT) { dead = 1; nnc_store_agent(a.timeout_word, 0xc0000000u | (unsigned)s); break; }
					NNC_SPIN_SLEEP();
					gr[i] = nnc_load_granule(w[i]);
				}
				const int q = q0 + 256 * i, r2 = q / H, k = q - r2 * H;
				if (q < 16 * H) htile[r2 * PITCH + k] = __uint_as_float((unsigned)gr[i]);
			}
		}
		if (KT > H) for (int q = tid; q < 16 * (KT - H); q += 256) { const int r2 = q / (KT - H), k = H + q - r2 * (KT - H); htile[r2 * PITCH + k] = 0.f; }
		__syncthreads();
		// this wave's gate for the tile's 16 rows x the workgroup's 16 units on the matrix cores: the A fragment -- lane (row = l & 15, k = 4 kk + (l >> 4)) -- out of LDS
		floatx4 acc = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
		for (int i = 0; i < KPT; i++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(htile[(c & 15) * PITCH + 4 * i + (c >> 4)], rreg[i], acc, 0, 0, 0);
#pragma unroll
		for (int i = 0; i < 4; i++) part[kq][4 * (c >> 4) + i][c & 15] = acc[i]; // (D: rows 4 (l >> 4) + i, column l & 15)
		__syncthreads();
		if (mine) {
			float* const gates = a.rsv ? a.rsv + a.slot0 + (size_t)s * a.S * BH : 0;
			float hnew = hst;
			if (t < len) {
				float pre[4];
#pragma unroll
				for (int g = 0; g < 4; g++) pre[g] = part[g][rr][u] + gin[g] + bias[g];
				const float i = lstm_sigmoid(pre[0]), f = lstm_sigmoid(pre[1]), g = tanhf(pre[2]), o = lstm_sigmoid(pre[3]);
				cst = f * cst + i * g;
				const float tc = tanhf(cst);
				hnew = o * tc;
				a.y[((size_t)t * B + b) * a.ldy + j] = hnew;
				if (gates) { gates[e] = i; gates[BH + e] = f; gates[2 * BH + e] = g; gates[3 * BH + e] = o; gates[4 * BH + e] = tc; }
			} else {
				a.y[((size_t)t * B + b) * a.ldy + j] = 0.f;
				if (gates) for (int k = 0; k < 5; k++) gates[k * BH + e] = 0.f;
			}
			hst = hnew;
			if (s < a.T - 1) {
				nnc_store_granule(a.xch + ((size_t)(s & 1) * B + b) * H + j, (unsigned)(s + 1), hnew);
				if (a.rsv) a.rsv[a.cslot0 + (size_t)s * BH + e] = cst;
			}
		}
		__syncthreads();
	}
	if (mine) { if (a.hy) a.hy[e] = hst; if (a.cy) a.cy[e] = cst; }
}

// The backward pass of one pseudo-layer's whole sequence in ONE launch (hidden size <= 512: NCH = ceil(H / 128) chunks of dG through LDS per step): the same ownership as lstm_seq_forw_kernel.  Per step, thread (row, unit)
// turns its state gradients (dh, dc: registers for the whole sequence) into the four gate gradients, writes them to dG (the contractions after the loop read them)
// and publishes them as tagged words; the tile's workgroups gather the tile's dG [16][4H] into LDS, and every workgroup forms dh' = dG R for ITS 16 units on the matrix cores --
// wave w reduces columns w * 128 .. + 127 of every 512-column chunk of dG that passes through LDS, its R coefficients (the B fragments of v_mfma_f32_16x16x4_f32: H / 4 registers per lane) resident, the four partial blocks meet in LDS.
struct lstm_seq_back_t {
	const float* r; const float* rsv; const float* cx; const float* dy; const float* dhy; const float* dcy;
	float* dg; float* dhx; float* dcx; unsigned long long* xch; const int* lens; unsigned* timeout_word;
	int T, B, H, dir, ldy, S;
	size_t slot0, cslot0;
};
constexpr int LSTM_SEQ_BACK_NC = 512; // gate-gradient columns staged in LDS at a time: 4H in NCH = ceil(H / 128) chunks
#define LSTM_SEQ_BACK_LDS (sizeof(float) * (16 * (LSTM_SEQ_BACK_NC + 4) + 4 * 16 * 17 + 4))
template <int NCH>
__global__ void __launch_bounds__(256) lstm_seq_back_kernel(const lstm_seq_back_t a)
{
	constexpr int NC = LSTM_SEQ_BACK_NC, PITCH = NC + 4;
	HIP_DYNAMIC_SHARED(float, lds)
	float* const dgtile = lds;                                            // [16][PITCH]
	float (*const part)[16][17] = (float (*)[16][17])(lds + 16 * PITCH);  // [4 waves][16 rows][16 units]
	int& dead = *(int*)(lds + 16 * PITCH + 4 * 16 * 17);
	const int tid = threadIdx.x, u = tid & 15, rr = tid >> 4; // (row, unit): the element of dh / dc a thread keeps for the whole sequence
	const int H = a.H, B = a.B, N4 = 4 * H, j0 = blockIdx.x * 16, row0 = blockIdx.y * 16;
	const size_t BH = (size_t)B * H;
	const int wq = tid >> 6, l = tid & 63; // wave wq reduces columns wq * 128 .. + 127 of every 512-column chunk on the matrix cores
	float rreg[NCH][32]; // the B fragments: lane (unit = l & 15, n = 4 kk + (l >> 4)) of R[chunk's columns of this wave][j0 .. j0 + 15]
#pragma unroll
	for (int ch = 0; ch < NCH; ch++) {
		const int ju = j0 + (l & 15) < H ? j0 + (l & 15) : H - 1;
#pragma unroll
		for (int i = 0; i < 32; i++) {
			const int n = ch * NC + wq * 128 + 4 * i + (l >> 4);
			const float v = a.r[(unsigned)((n < N4 ? n : N4 - 1) * H + ju)];
			rreg[ch][i] = n < N4 && j0 + (l & 15) < H ? v : 0.f;
		}
	}
	const int b = row0 + rr, j = j0 + u;
	const bool mine = b < B && j < H;
	const size_t e = (size_t)b * H + j;
	float dh = mine && a.dhy ? a.dhy[e] : 0.f, dc = mine && a.dcy ? a.dcy[e] : 0.f;
	const int len = mine && a.lens ? a.lens[b] : a.T;
	if (tid == 0) dead = 0;
	__syncthreads();
	for (int it = 0; it < a.T; it++) {
		const int s = a.T - 1 - it, t = a.dir ? a.T - 1 - s : s;
		if (mine) {
			float d4[4] = { 0.f, 0.f, 0.f, 0.f };
			if (t < len) {
				const float* const gates = a.rsv + a.slot0 + (size_t)s * a.S * BH;
				const float i = gates[e], f = gates[BH + e], g = gates[2 * BH + e], o = gates[3 * BH + e], tc = gates[4 * BH + e];
				const float cprev = s == 0 ? (a.cx ? a.cx[e] : 0.f) : a.rsv[a.cslot0 + (size_t)(s - 1) * BH + e];
				const float dht = dh + a.dy[((size_t)t * B + b) * a.ldy + j];
				const float dct = dc + dht * o * (1.f - tc * tc);
				d4[0] = dct * g * i * (1.f - i);
				d4[1] = dct * cprev * f * (1.f - f);
				d4[2] = dct * i * (1.f - g * g);
				d4[3] = dht * tc * o * (1.f - o);
				dc = dct * f;
			}
			float* const dgt = a.dg + ((size_t)t * B + b) * N4 + j;
#pragma unroll
			for (int g = 0; g < 4; g++) {
				dgt[g * H] = d4[g];
				nnc_store_granule(a.xch + ((size_t)(it & 1) * B + b) * N4 + g * H + j, (unsigned)(it + 1), d4[g]);
			}
		}
		floatx4 acc = { 0.f, 0.f, 0.f, 0.f };
		auto chunk = [&](const int ch, const float (&rc)[32]) __attribute__((always_inline)) {
			const int n0 = ch * NC, nn = N4 - n0 < NC ? N4 - n0 : NC; // this chunk's columns
			if (ch > 0) __syncthreads();
			__builtin_amdgcn_sched_barrier(0); // (one chunk's words at a time: hoisting the next chunk's loads over this one's arithmetic spills)
#pragma unroll 1
			for (int q0 = tid; q0 < 16 * NC; q0 += 256 * 8) { // (row = q / 512: shifts; eight words in flight per thread, not all 32 of the chunk: registers)
				unsigned long long gr[8];
				const unsigned long long* w[8];
#pragma unroll
				for (int i = 0; i < 8; i++) {
					const int q = q0 + 256 * i, r2 = q / NC, n = q - r2 * NC, b2 = row0 + r2;
					const bool on = n < nn && b2 < B;
					w[i] = a.xch + ((size_t)(it & 1) * B + (on ? b2 : 0)) * N4 + (on ? n0 + n : 0);
					gr[i] = (unsigned long long)(unsigned)(it + 1) << 32; // (columns / rows past the end: "arrived", zero)
					if (on) gr[i] = nnc_load_granule(w[i]);
				}
#pragma unroll
				for (int i = 0; i < 8; i++) {
					unsigned spins = 0;
					while ((unsigned)(gr[i] >> 32) != (unsigned)(it + 1)) {
						if (*(volatile int*)&dead) break;
						if (++spins > CLUSTER_SPIN_LIMIT) { dead = 1; nnc_store_agent(a.timeout_word, 0xd0000000u | (unsigned)it); break; }
						NNC_SPIN_SLEEP();
						gr[i] = nnc_load_granule(w[i]);
					}
					const int q = q0 + 256 * i, r2 = q / NC, n = q - r2 * NC;
					dgtile[r2 * PITCH + n] = __uint_as_float((unsigned)gr[i]);
				}
			}
			__syncthreads();
#pragma unroll
			for (int i = 0; i < 32; i++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dgtile[(l & 15) * PITCH + wq * 128 + 4 * i + (l >> 4)], rc[i], acc, 0, 0, 0);
		};
		chunk(0, rreg[0]);
		if constexpr (NCH > 1) chunk(1, rreg[1]);
		if constexpr (NCH > 2) chunk(2, rreg[2]);
		if constexpr (NCH > 3) chunk(3, rreg[3]);
#pragma unroll
		for (int i = 0; i < 4; i++) part[wq][4 * (l >> 4) + i][l & 15] = acc[i]; // (D: rows 4 (l >> 4) + i, column l & 15; one partial block per wave)
		__syncthreads();
		if (mine) {
			const float v = (part[0][rr][u] + part[1][rr][u]) + (part[2][rr][u] + part[3][rr][u]);
			dh = t < len ? v : dh + v; // (past the end the gate gradients were zero: v == 0, the state gradient goes on unchanged)
		}
		__syncthreads();
	}
	if (mine) { if (a.dhx) a.dhx[e] = dh; if (a.dcx) a.dcx[e] = dc; }
}

// ---- the whole sequence of one pseudo-layer in ONE launch WITHOUT any traffic between workgroups (TUNE_LSTM_ROWS; round 5; no projection, hidden size <= 128) ----
// The one-launch kernels above split the HIDDEN units over workgroups, so every step ends with the tile's workgroups handing each other the new state through
// tagged words in the L2: 5.8 us per step forward, 8.3 backward at the IMDB classifier's shape (64 x 512 x 128), of which the arithmetic is a tenth.  For H <= 128
// all of R -- 4H x H floats, 256 KB at H = 128 -- fits the REGISTER FILE of one workgroup: 4H threads (eight waves at H = 128), thread n keeps row n of R (H
// registers).  A workgroup then owns LS_RB = 2 batch rows and ALL hidden units for the whole sequence; the state never leaves its LDS, there is nothing to wait
// for and nothing that can time out (no ClusterTurn, no co-residency condition: any number of these launches may share the device), and B / 2 workgroups run
// side by side.  Per step: thread n forms pre[r][n] = sum_k h'[r][k] R[n][k] for its gate column -- the state read as one broadcast 8-byte LDS word per k, the
// two rows in one packed multiply-add --, the columns meet in LDS, thread (r, u) does the gate arithmetic of its element, writes y and the tape and puts the new
// state back into LDS.  Backward: thread (r, u) turns its state gradients into the four gate gradients (tape loads issued a step ahead), thread (q, k) holds
// column k of gate q's block of R and reduces that block's 128 terms of dh' = dG R, the four partial sums meet in LDS.
// Cost per step ~ the LDS broadcast of the state to eight waves (H x 512 bytes per wave) -- the register-resident multiply-adds run beside it.
// LS_RB = 1 (round 5, second form; hidden sizes in multiples of four): ONE batch row per workgroup.  The step's cost is the LDS broadcast of the state to the
// eight waves (every lane of a wave reads the same 16 bytes: the LDS charges the full 64 lanes' width for it) and the multiply-adds behind it, both per ROW of the
// workgroup -- so a row per workgroup halves the step's latency and doubles the workgroups (the IMDB shape's 64 rows: 64 of the 256 CUs instead of 32); the packed
// multiply-add then pairs two consecutive k of the one row (R's row is already laid out in pairs).
typedef float lstm_f2 __attribute__((ext_vector_type(2))); // an aligned register pair: v_pk_fma_f32
template <int HT, int LS_RB>
__global__ void __launch_bounds__(4 * HT) lstm_rows_forw_kernel(const lstm_seq_t a)
{
	__shared__ __attribute__((aligned(16))) float htile[HT][LS_RB];      // the state before the step, k-major: one 8-byte broadcast read per k (LS_RB = 2) / 16 bytes per four k (LS_RB = 1)
	__shared__ float pre[LS_RB][4 * HT + 4];
	const int tid = threadIdx.x, H = a.H, B = a.B, N4 = 4 * H, row0 = blockIdx.x * LS_RB;
	const size_t BH = (size_t)B * H;
	lstm_f2 rreg[HT / 2]; // row tid of R (zero past 4H / H), two consecutive k per register pair
#pragma unroll
	for (int k = 0; k < HT; k++) rreg[k >> 1][k & 1] = tid < N4 && k < H ? a.r[(size_t)tid * H + k] : 0.f;
	const int rr = tid / H, u = tid - rr * H, b = row0 + rr; // (row, unit): the element whose cell and state this thread keeps for the whole sequence
	const bool mine = rr < LS_RB && b < B;
	const size_t e = (size_t)b * H + u;
	float cst = mine && a.cx ? a.cx[e] : 0.f, hst = mine && a.hx ? a.hx[e] : 0.f;
	const int len = mine && a.lens ? a.lens[b] : a.T;
	float bias[4] = { 0.f, 0.f, 0.f, 0.f };
	if (mine && a.bw)
#pragma unroll
		for (int g = 0; g < 4; g++) bias[g] = a.bw[g * H + u] + a.bw[4 * H + g * H + u];
	for (int q = tid; q < HT * LS_RB; q += 4 * HT) (&htile[0][0])[q] = 0.f;
	__syncthreads();
	if (mine) htile[u][rr] = hst;
	__syncthreads();
	float gnx[4] = { 0.f, 0.f, 0.f, 0.f };
	auto load_gin = [&](const int s) {
		const int t = a.dir ? a.T - 1 - s : s;
		if (mine && s < a.T) {
			const float* const gr = a.gx + ((size_t)t * B + b) * N4 + u;
#pragma unroll
			for (int g = 0; g < 4; g++) gnx[g] = gr[g * H];
		}
	};
	load_gin(0);
	for (int s = 0; s < a.T; s++) {
		const int t = a.dir ? a.T - 1 - s : s;
		float gin[4];
#pragma unroll
		for (int g = 0; g < 4; g++) gin[g] = gnx[g];
		load_gin(s + 1); // the input half of the NEXT step (a ring of four steps ahead, unrolled, measured slower: profiles/r05_v10_lstm_forms.txt)
		lstm_f2 acc4[4] = { { 0.f, 0.f }, { 0.f, 0.f }, { 0.f, 0.f }, { 0.f, 0.f } }; // four chains: a packed multiply-add waits for its predecessor otherwise
		if (LS_RB == 2) {
#pragma unroll
			for (int k2 = 0; k2 < HT / 2; k2++) {
				const float4 hh = ((const float4*)&htile[0][0])[k2]; // rows 0, 1 of k = 2 k2 and of k = 2 k2 + 1
				const lstm_f2 h0 = { hh.x, hh.y }, h1 = { hh.z, hh.w };
				NNC_PK_FMA_LO(acc4[(2 * k2) & 3], h0, rreg[k2]);
				NNC_PK_FMA_HI(acc4[(2 * k2 + 1) & 3], h1, rreg[k2]);
			}
			const lstm_f2 acc = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
			if (tid < N4) { pre[0][tid] = acc[0]; pre[LS_RB - 1][tid] = acc[1]; }
		} else {
#pragma unroll
			for (int k4 = 0; k4 < HT / 4; k4++) { // the one row's state at k = 4 k4 .. 4 k4 + 3; a register pair of R = two consecutive k
				const float4 hh = ((const float4*)&htile[0][0])[k4];
				acc4[(2 * k4) & 3] += lstm_f2{ hh.x, hh.y } * rreg[2 * k4];
				acc4[(2 * k4 + 1) & 3] += lstm_f2{ hh.z, hh.w } * rreg[2 * k4 + 1];
			}
			const lstm_f2 acc = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
			if (tid < N4) pre[0][tid] = acc[0] + acc[1];
		}
		NNC_LDS_BARRIER();
		if (mine) {
			float* const gates = a.rsv ? a.rsv + a.slot0 + (size_t)s * a.S * BH : 0;
			float hnew = hst;
			if (t < len) {
				float p4[4];
#pragma unroll
				for (int g = 0; g < 4; g++) p4[g] = pre[rr][g * H + u] + gin[g] + bias[g];
				const float i = lstm_fast_sigmoid(p4[0]), f = lstm_fast_sigmoid(p4[1]), g = lstm_fast_tanh(p4[2]), o = lstm_fast_sigmoid(p4[3]);
				cst = f * cst + i * g;
				const float tc = lstm_fast_tanh(cst);
				hnew = o * tc;
				a.y[((size_t)t * B + b) * a.ldy + u] = hnew;
				if (gates) { gates[e] = i; gates[BH + e] = f; gates[2 * BH + e] = g; gates[3 * BH + e] = o; gates[4 * BH + e] = tc; }
			} else {
				a.y[((size_t)t * B + b) * a.ldy + u] = 0.f;
				if (gates) for (int k = 0; k < 5; k++) gates[k * BH + e] = 0.f;
			}
			hst = hnew;
			htile[u][rr] = hnew;
			if (s < a.T - 1 && a.rsv) a.rsv[a.cslot0 + (size_t)s * BH + e] = cst;
		}
		NNC_LDS_BARRIER();
	}
	if (mine) { if (a.hy) a.hy[e] = hst; if (a.cy) a.cy[e] = cst; }
}

template <int HT, int LS_RB>
__global__ void __launch_bounds__(4 * HT) lstm_rows_back_kernel(const lstm_seq_back_t a)
{
	__shared__ __attribute__((aligned(16))) float dgt[4 * HT][LS_RB];    // the step's gate gradients, column-major: one 8-byte broadcast read per column
	__shared__ float part[4][LS_RB][HT + 1];
	const int tid = threadIdx.x, H = a.H, B = a.B, N4 = 4 * H, row0 = blockIdx.x * LS_RB;
	const size_t BH = (size_t)B * H;
	const int q = tid / H, kc = tid - q * H; // (gate block, unit): column kc of R's rows q H .. q H + H - 1
	lstm_f2 rreg[HT / 2];
#pragma unroll
	for (int i = 0; i < HT; i++) rreg[i >> 1][i & 1] = tid < N4 && i < H ? a.r[(size_t)(q * H + i) * H + kc] : 0.f;
	const int rr = q, u = kc, b = row0 + rr; // the same split of the thread index names the element (row, unit) of dh / dc this thread keeps
	const bool mine = rr < LS_RB && b < B;
	const size_t e = (size_t)b * H + u;
	float dh = mine && a.dhy ? a.dhy[e] : 0.f, dc = mine && a.dcy ? a.dcy[e] : 0.f;
	const int len = mine && a.lens ? a.lens[b] : a.T;
	for (int z = tid; z < 4 * HT * LS_RB; z += 4 * HT) (&dgt[0][0])[z] = 0.f; // (rows past the batch, columns past 4H: zero for good)
	__syncthreads();
	// the tape of the step about to be processed, loaded a step ahead
	float ti = 0.f, tf = 0.f, tg = 0.f, to = 0.f, ttc = 0.f, tcp = 0.f, tdy = 0.f;
	auto load_tape = [&](const int it) {
		const int s = a.T - 1 - it, t = a.dir ? a.T - 1 - s : s;
		if (mine && it < a.T && t < len) {
			const float* const gates = a.rsv + a.slot0 + (size_t)s * a.S * BH;
			ti = gates[e]; tf = gates[BH + e]; tg = gates[2 * BH + e]; to = gates[3 * BH + e]; ttc = gates[4 * BH + e];
			tcp = s == 0 ? (a.cx ? a.cx[e] : 0.f) : a.rsv[a.cslot0 + (size_t)(s - 1) * BH + e];
			tdy = a.dy[((size_t)t * B + b) * a.ldy + u];
		}
	};
	load_tape(0);
	for (int it = 0; it < a.T; it++) {
		const int s = a.T - 1 - it, t = a.dir ? a.T - 1 - s : s;
		if (mine) {
			float d4[4] = { 0.f, 0.f, 0.f, 0.f };
			if (t < len) {
				const float i = ti, f = tf, g = tg, o = to, tc = ttc, cprev = tcp;
				const float dht = dh + tdy;
				const float dct = dc + dht * o * (1.f - tc * tc);
				d4[0] = dct * g * i * (1.f - i);
				d4[1] = dct * cprev * f * (1.f - f);
				d4[2] = dct * i * (1.f - g * g);
				d4[3] = dht * tc * o * (1.f - o);
				dc = dct * f;
			}
			float* const dgo = a.dg + ((size_t)t * B + b) * N4 + u;
#pragma unroll
			for (int g = 0; g < 4; g++) { dgo[g * H] = d4[g]; dgt[g * H + u][rr] = d4[g]; }
		}
		load_tape(it + 1);
		NNC_LDS_BARRIER();
		lstm_f2 acc4[4] = { { 0.f, 0.f }, { 0.f, 0.f }, { 0.f, 0.f }, { 0.f, 0.f } };
		const int nb = tid < N4 ? q * H : 0;
		if (LS_RB == 2) {
#pragma unroll
			for (int i2 = 0; i2 < HT / 2; i2++) { // (nb + i < 3 H + HT <= 4 HT; past the block's H terms the coefficient is zero)
				const lstm_f2 d0 = *(const lstm_f2*)&dgt[nb + 2 * i2][0], d1 = *(const lstm_f2*)&dgt[nb + 2 * i2 + 1][0];
				NNC_PK_FMA_LO(acc4[(2 * i2) & 3], d0, rreg[i2]);
				NNC_PK_FMA_HI(acc4[(2 * i2 + 1) & 3], d1, rreg[i2]);
			}
			const lstm_f2 acc = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
			if (tid < N4) { part[q][0][kc] = acc[0]; part[q][LS_RB - 1][kc] = acc[1]; }
		} else {
			const float4* const d4 = (const float4*)&dgt[nb][0]; // (H in multiples of four: 16-byte aligned)
#pragma unroll
			for (int i4 = 0; i4 < HT / 4; i4++) { // the one row's gate gradients of columns nb + 4 i4 .. + 3; a register pair of R = two consecutive rows of the block
				const float4 dd = d4[i4];
				acc4[(2 * i4) & 3] += lstm_f2{ dd.x, dd.y } * rreg[2 * i4];
				acc4[(2 * i4 + 1) & 3] += lstm_f2{ dd.z, dd.w } * rreg[2 * i4 + 1];
			}
			const lstm_f2 acc = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
			if (tid < N4) part[q][0][kc] = acc[0] + acc[1];
		}
		NNC_LDS_BARRIER();
		if (mine) {
			const float v = (part[0][rr][u] + part[1][rr][u]) + (part[2][rr][u] + part[3][rr][u]);
			dh = t < len ? v : dh + v; // (past the end the gate gradients were zero: v == 0, the state gradient goes on unchanged)
		}
	}
	if (mine) { if (a.dhx) a.dhx[e] = dh; if (a.dcx) a.dcx[e] = dc; }
}

// Measured and not in the tree (profiles/r05_v10_lstm_forms.txt, the IMDB shape, forward / backward per launch of 512 steps): the one-row form above 0.556 / 0.746 ms;
// the reduction split eight ways over lanes with the partial sums meeting in LDS and one gate column per thread for the activation (three barriers per
// step) 0.628 / 0.893; split over the four lanes of a unit with two lane exchanges, the state double-buffered, one barrier per step 0.721 / 1.340 -- fewer LDS
// cycles each time, a longer chain of dependent round trips each time: the step is a latency chain, not an LDS-throughput problem; the input half loaded
// four steps ahead through an unrolled register ring 1.245 forward.  What did pay: a row per workgroup instead of two (1.026 / 1.670 -> 0.680 / 0.752: half
// the per-step work on the chain, twice the workgroups) and the hardware's 2^x and 1 / x for the activations (0.659 -> 0.556 forward).

// One step of one pseudo-layer: the four gates' recurrent products + the gate arithmetic.  direct = no projection (hout is the next state, P == H).
__global__ void __launch_bounds__(256) lstm_step_forw_kernel(const float* const gx, const float* const rt, const float* const bw, const float* const br, const float* const hprev, const float* const cprev,
	float* const hout, float* const cnext, float* const gates, float* const cstore, float* const y, const int ldy, const int* const lens, const int t, const int B, const int H, const int P, const int direct)
{
	__shared__ float tile[LS_ROWS][LS_KT];
	const int j = blockIdx.x * 64 + (threadIdx.x & 63), row0 = blockIdx.y * LS_ROWS, grp = threadIdx.x >> 6;
	float acc[4][4];
	lstm_rows_times<4>(hprev, P, rt, 4 * H, H, P, B, row0, j, j < H, tile, acc);
	if (j >= H) return;
	const size_t BH = (size_t)B * H;
	for (int r = 0; r < 4; r++) {
		const int b = row0 + grp * 4 + r;
		if (b >= B) break;
		const size_t e = (size_t)b * H + j;
		const float cp = cprev[e];
		if (lens && t >= lens[b]) { // past this item's end: the state goes on unchanged
			cnext[e] = cp;
			hout[e] = direct ? hprev[e] : 0.f;
			if (y) y[(size_t)b * ldy + j] = 0.f;
			if (cstore) cstore[e] = cp;
			if (gates) for (int k = 0; k < 5; k++) gates[k * BH + e] = 0.f;
			continue;
		}
		const float* const gr = gx + (size_t)b * 4 * H + j;
		float a[4];
#pragma unroll
		for (int k = 0; k < 4; k++) a[k] = acc[k][r] + gr[k * H] + (bw ? bw[k * H + j] + br[k * H + j] : 0.f);
		const float i = lstm_sigmoid(a[0]), f = lstm_sigmoid(a[1]), g = tanhf(a[2]), o = lstm_sigmoid(a[3]);
		const float c = f * cp + i * g, tc = tanhf(c), h = o * tc;
		cnext[e] = c;
		hout[e] = h;
		if (y) y[(size_t)b * ldy + j] = h;
		if (cstore) cstore[e] = c;
		if (gates) { gates[e] = i; gates[BH + e] = f; gates[2 * BH + e] = g; gates[3 * BH + e] = o; gates[4 * BH + e] = tc; }
	}
}

// out[b][n] = sum_k a[b][k] mt[k][n] with the step's bookkeeping behind it.
// MODE 0 (projection, forward): out = the next state (the sum, or pass[b][n] past the end), y = the sum or 0, store (row stride ldo2) = the same
// MODE 1: out = the sum      MODE 2 (state gradient): out = the sum + (past the end ? pass[b][n] : 0)
template <int MODE>
__global__ void __launch_bounds__(256) lstm_rowmat_kernel(const float* const a, const int lda, const float* const mt, const int ldm, const int K, const int N, const int B, float* const out, const int ldo,
	const float* const pass, float* const y, const int ldy, float* const store, const int ldo2, const int* const lens, const int t)
{
	__shared__ float tile[LS_ROWS][LS_KT];
	const int n = blockIdx.x * 64 + (threadIdx.x & 63), row0 = blockIdx.y * LS_ROWS, grp = threadIdx.x >> 6;
	float acc[1][4];
	lstm_rows_times<1>(a, lda, mt, ldm, 0, K, B, row0, n, n < N, tile, acc);
	if (n >= N) return;
	for (int r = 0; r < 4; r++) {
		const int b = row0 + grp * 4 + r;
		if (b >= B) break;
		const bool past = lens && t >= lens[b];
		const float v = acc[0][r];
		if (MODE == 0) {
			out[(size_t)b * ldo + n] = past ? pass[(size_t)b * ldo + n] : v;
			if (y) y[(size_t)b * ldy + n] = past ? 0.f : v;
			if (store) store[(size_t)b * ldo2 + n] = past ? 0.f : v;
		} else if (MODE == 1) out[(size_t)b * ldo + n] = v;
		else out[(size_t)b * ldo + n] = v + (past ? pass[(size_t)b * ldo + n] : 0.f);
	}
}

// The gate gradients of one step: dg[b][4H] (pre-activation), dc in place.  dh: the state gradient [B][H] (direct: + dy's columns of this direction).
__global__ void __launch_bounds__(256) lstm_step_back_kernel(const float* const gates, const float* const cprev, const float* const dh, const float* const dy, const int ldy, float* const dc, float* const dg, const int* const lens, const int t, const int B, const int H)
{
	const size_t BH = (size_t)B * H;
	const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (e >= BH) return;
	const int b = (int)(e / H), j = (int)(e % H);
	float* const d = dg + (size_t)b * 4 * H + j;
	if (lens && t >= lens[b]) { d[0] = d[H] = d[2 * H] = d[3 * H] = 0.f; return; }
	const float i = gates[e], f = gates[BH + e], g = gates[2 * BH + e], o = gates[3 * BH + e], tc = gates[4 * BH + e];
	const float dht = dh[e] + (dy ? dy[(size_t)b * ldy + j] : 0.f);
	const float dct = dc[e] + dht * o * (1.f - tc * tc);
	d[0] = dct * g * i * (1.f - i);
	d[H] = dct * cprev[e] * f * (1.f - f);
	d[2 * H] = dct * i * (1.f - g * g);
	d[3 * H] = dht * tc * o * (1.f - o);
	dc[e] = dct * f;
}

// projection, backward: what reaches the projected state of step t -- the recurrent gradient + dy's columns, zero past the end
__global__ void __launch_bounds__(256) lstm_dhp_kernel(const float* const dh, const float* const dy, const int ldy, float* const dhp, const int* const lens, const int t, const int B, const int P)
{
	const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (e >= (size_t)B * P) return;
	const int b = (int)(e / P), k = (int)(e % P);
	dhp[e] = (lens && t >= lens[b]) ? 0.f : dh[e] + dy[(size_t)b * ldy + k];
}

struct lstm_view_t { // what the element-wise passes need to find a pseudo-layer's planes in the reserved space
	const float* r;
	int T, B, H, P, D, S, proj, drop;
};
__device__ __forceinline__ size_t lstm_slot(const lstm_view_t& v, const int p, const int s, const int k) { return (((size_t)p * v.T + s) * v.S + k) * ((size_t)v.B * v.H); }
// the output of pseudo-layer p at processing step s for (b, j), before dropout
__device__ __forceinline__ float lstm_h_of(const lstm_view_t& v, const int p, const int s, const int b, const int j)
{
	const size_t e = (size_t)b * v.H + j;
	return v.proj ? v.r[lstm_slot(v, p, s, 5) + e] : v.r[lstm_slot(v, p, s, 3) + e] * v.r[lstm_slot(v, p, s, 4) + e];
}

// MODE 0: x[t][b][d * P + j] = what layer `layer` handed up (its outputs times the dropout scales, zero past the end)
// MODE 1: x[t][b][k] = the state pseudo-layer p = layer * D + dir had BEFORE its step at t (hx at an item's first step; zero past the end)
// MODE 2: x[t][b][j] = o tanh(c) of pseudo-layer p at t (the projection's input; zero past the end)
template <int MODE>
__global__ void __launch_bounds__(256) lstm_gather_kernel(const lstm_view_t v, const int layer, const int dir, const float* const hx, const int* const lens, float* const x)
{
	const int W = MODE == 0 ? v.D * v.P : MODE == 1 ? v.P : v.H;
	const size_t n = (size_t)v.T * v.B * W;
	const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (e >= n) return;
	const int c = (int)(e % W), b = (int)((e / W) % v.B), t = (int)(e / ((size_t)W * v.B));
	const int len = lens ? lens[b] : v.T;
	if (t >= len) { x[e] = 0.f; return; }
	if (MODE == 0) {
		const int d = c / v.P, j = c % v.P, p = layer * v.D + d, s = d ? v.T - 1 - t : t;
		const float h = lstm_h_of(v, p, s, b, j);
		x[e] = v.drop ? h * v.r[lstm_slot(v, p, s, v.S - 1) + (size_t)b * v.H + j] : h;
	} else if (MODE == 1) {
		const int p = layer * v.D + dir, s = dir ? v.T - 1 - t : t;
		const bool first = dir ? t == len - 1 : t == 0;
		x[e] = first ? (hx ? hx[((size_t)p * v.B + b) * v.P + c] : 0.f) : lstm_h_of(v, p, s - 1, b, c);
	} else {
		const int p = layer * v.D + dir, s = dir ? v.T - 1 - t : t;
		const size_t q = (size_t)b * v.H + c;
		x[e] = v.r[lstm_slot(v, p, s, 3) + q] * v.r[lstm_slot(v, p, s, 4) + q];
	}
}

__host__ __device__ __forceinline__ unsigned lstm_mix32(unsigned x)
{
	x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
	return x;
}
// forward: draw the scales of layer `layer`'s outputs (0 or 1 / (1 - p)), keep them in the reserved space, scale y in place; backward (DRAW = false): dy *= the kept scales
template <bool DRAW>
__global__ void __launch_bounds__(256) lstm_dropout_kernel(const lstm_view_t v, float* const r, const int layer, float* const y, const unsigned seed0, const unsigned* const tick, const float p, const float inv_keep)
{
	const unsigned seed = tick ? seed0 + 0x9e3779b9U * tick[0] : seed0; // (a captured step: the replay's tick, common.h capture_tick_of)
	const int W = v.D * v.P;
	const size_t n = (size_t)v.T * v.B * W;
	const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (e >= n) return;
	const int c = (int)(e % W), b = (int)((e / W) % v.B), t = (int)(e / ((size_t)W * v.B));
	const int d = c / v.P, j = c % v.P, q = layer * v.D + d, s = d ? v.T - 1 - t : t;
	const size_t at = lstm_slot(v, q, s, v.S - 1) + (size_t)b * v.H + j;
	if (DRAW) {
		const unsigned h = lstm_mix32(lstm_mix32((unsigned)e ^ seed) + (unsigned)(e >> 32) + 0x9e3779b9U * (unsigned)(layer + 1));
		const float m = (float)(h >> 8) * (1.f / 16777216.f) <= p ? 0.f : inv_keep;
		r[at] = m;
		y[e] *= m;
	} else y[e] *= v.r[at];
}

// out[j][i][f] = in[i][j][f]: batch-first tensors <-> the sequence-major order the steps walk
__global__ void __launch_bounds__(256) lstm_swap01_kernel(const float* const in, float* const out, const int A, const int Bn, const int F)
{
	const size_t n = (size_t)A * Bn * F;
	const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (e >= n) return;
	const int f = (int)(e % F), j = (int)((e / F) % Bn), i = (int)(e / ((size_t)F * Bn));
	out[((size_t)j * A + i) * F + f] = in[e];
}
// out[c][r] = in[r][c]
__global__ void __launch_bounds__(256) lstm_transpose_kernel(const float* const in, float* const out, const int R, const int C)
{
	const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (e >= (size_t)R * C) return;
	const int c = (int)(e % C), r = (int)(e / C);
	out[(size_t)c * R + r] = in[e];
}

static inline unsigned blocks_of(const size_t n) { return (unsigned)((n + 255) / 256); }
static inline size_t al(const size_t n) { return (n * sizeof(float) + 255) & ~(size_t)255; }

static int lstm_lens(const ccv_nnc_tensor_t* const xs, const lstm_geom_t& g, int* const dev, hipStream_t stream)
{ // the lengths live in host memory (ccv_nnc_lstm_gpu_cudnn.cu:67-74)
	if (CCV_GET_DATA_TYPE(xs->info.datatype) != CCV_32S || tensor_nd(xs->info.dim) != 1 || xs->info.dim[0] != g.B || CCV_TENSOR_GET_MEMORY(xs->info.type) != CCV_TENSOR_CPU_MEMORY) return CCV_NNC_EXEC_INVALID;
	for (int b = 0; b < g.B; b++) if (xs->data.i32[b] < 0 || xs->data.i32[b] > g.T) return CCV_NNC_EXEC_INVALID;
	HIP_ENFORCE(hipMemcpyAsync(dev, xs->data.i32, sizeof(int) * g.B, hipMemcpyHostToDevice, stream)); // pageable source: copied before return
	return CCV_NNC_EXEC_SUCCESS;
}

static size_t lstm_inner_bytes(const lstm_geom_t& g)
{ // what the contractions and the column sum made underneath may ask of the workspace (split-K slabs: the bound is not monotonic in the shape, so every shape is listed)
	const long TB = (long)g.T * g.B, G4 = 4L * g.H;
	const int ins[2] = { g.I, g.D * g.P };
	size_t inner = sizeof(float) * ((size_t)device_cu_count() * 4 + 64) * 4 * g.H; // colsum_f32's partials
	auto take = [&](const long M, const long N, const long K) { const size_t v = gemm_workspace_bound(M, N, K); if (v > inner) inner = v; };
	for (int i = 0; i < 2; i++) {
		take(TB, G4, ins[i]); // gx = X W^T
		take(TB, ins[i], G4); // dX = dG W
		take(G4, ins[i], TB); // dW = dG^T X
	}
	take(G4, g.P, TB);  // dR = dG^T H'
	take(g.P, g.H, TB); // dW_p
	take(g.B, g.P, G4); // a wide layer's state gradient, one step
	return inner + 4096;
}

static lstm_view_t lstm_view(const lstm_geom_t& g, const float* const r)
{
	lstm_view_t v = { r, g.T, g.B, g.H, g.P, g.D, g.S, g.proj, g.dropout > 0.f };
	return v;
}

// The host's per-stream generator when the reference host is linked in (lib/nnc/ccv_nnc_stream.c:262), else a process counter (as DROPOUT_FORWARD, cmd_ew.cpp).
extern "C" uint32_t ccv_nnc_stream_context_genrand_uint32(ccv_nnc_stream_context_t* const stream_context) __attribute__((weak));
static unsigned lstm_seed(ccv_nnc_stream_context_t* const ctx)
{
	if (ccv_nnc_stream_context_genrand_uint32) return ccv_nnc_stream_context_genrand_uint32(ctx);
	static unsigned counter = 0x13198a2eU;
	return __sync_add_and_fetch(&counter, 0x9e3779b9U);
}

// inputs: x, [xs], [hx], [cx], w    outputs: y, [hy], [cy], r      (lib/nnc/cmd/rnn/ccv_nnc_lstm.c:8-15)
static int _lstm_forw(EXEC_ARGS_L)
{
	if (input_size < 5 || output_size < 1 || !inputs[0] || !inputs[4] || !outputs[0]) return CCV_NNC_EXEC_INVALID;
	const ccv_nnc_tensor_t* const x = inputs[0];
	const ccv_nnc_tensor_t* const xs = inputs[1];
	const ccv_nnc_tensor_t* const hx = inputs[2];
	const ccv_nnc_tensor_t* const cx = inputs[3];
	const ccv_nnc_tensor_t* const w = inputs[4];
	ccv_nnc_tensor_t* const y = outputs[0];
	ccv_nnc_tensor_t* const hy = output_size > 1 ? outputs[1] : 0;
	ccv_nnc_tensor_t* const cy = output_size > 2 ? outputs[2] : 0;
	ccv_nnc_tensor_t* const r = output_size > 3 ? outputs[3] : 0;
	lstm_geom_t g;
	if (!lstm_geometry(cmd, x, &g)) return CCV_NNC_EXEC_INVALID;
	const size_t TB = (size_t)g.T * g.B, LD = (size_t)g.L * g.D;
	if (!dense_f32(x, TB * g.I) || !dense_f32(y, TB * g.D * g.P) || !dense_f32(w, g.w_total()) || !dense_f32(hx, LD * g.B * g.P) || !dense_f32(cx, LD * g.B * g.H) || !dense_f32(hy, LD * g.B * g.P) || !dense_f32(cy, LD * g.B * g.H)) return CCV_NNC_EXEC_INVALID;
	const bool train = !g.is_test;
	if (train && !dense_reserve(r, g.reserve_total())) return CCV_NNC_EXEC_INVALID; // (ccv_nnc_lstm_gpu_cudnn.cu:114-119)
	MarkerScope marker(cmd.cmd);
	hipStream_t stream = stream_of(stream_context);
	const int DP = g.D * g.P;
	// scratch: [ lengths | x, y in sequence-major order (batch-first tensors) | two layer outputs | the input half of the gates | R k-major | W_p k-major | state x 2 x 2 | o tanh(c) | the persistent kernel's state words ]
	const size_t n_len = (sizeof(int) * g.B + 255) & ~(size_t)255, n_xs = g.batch_first ? al(TB * g.I) : 0, n_ys = g.batch_first ? al(TB * DP) : 0, n_lay = g.L > 1 ? al(TB * DP) : 0;
	const size_t n_gx = al(TB * 4 * g.H), n_rt = al((size_t)4 * g.H * g.P), n_wpt = g.proj ? al((size_t)g.P * g.H) : 0, n_h = al((size_t)g.B * g.P), n_c = al(g.BH()), n_raw = g.proj ? al(g.BH()) : 0;
	// the whole sequence in one launch: every workgroup must be resident at once (they wait for each other) -- the grid stays within the CU count
	// hidden size <= 128: the whole of R fits one workgroup's registers -- two batch rows per workgroup, no traffic between workgroups (lstm_rows_forw_kernel)
	const bool rows = tune(TUNE_LSTM_PERSISTENT) && tune(TUNE_LSTM_ROWS) && !g.proj && g.H <= 128;
	const bool persistent = !rows && tune(TUNE_LSTM_PERSISTENT) && !g.proj && g.H <= 512 && (long)((g.H + 15) / 16) * ((g.B + 15) / 16) <= device_cu_count();
	const size_t n_xch = persistent ? (sizeof(unsigned long long) * 2 * g.B * g.H + 255) & ~(size_t)255 : 0;
	WorkspaceScope ws(stream_context, n_len + n_xs + n_ys + 2 * n_lay + n_gx + n_rt + n_wpt + 2 * n_h + 2 * n_c + n_raw + n_xch, lstm_inner_bytes(g));
	char* at = (char*)ws.prefix();
	if (!at) return CCV_NNC_EXEC_OOM;
	int* const lens = xs ? (int*)at : 0; at += n_len;
	float* const xseq = (float*)at; at += n_xs;
	float* const yseq = (float*)at; at += n_ys;
	float* const lay[2] = { (float*)at, (float*)(at + n_lay) }; at += 2 * n_lay;
	float* const gx = (float*)at; at += n_gx;
	float* const rt = (float*)at; at += n_rt;
	float* const wpt = (float*)at; at += n_wpt;
	float* const hs[2] = { (float*)at, (float*)(at + n_h) }; at += 2 * n_h;
	float* const cs[2] = { (float*)at, (float*)(at + n_c) }; at += 2 * n_c;
	float* const hraw = (float*)at; at += n_raw;
	unsigned long long* const xch = (unsigned long long*)at;
	unsigned* timeout_word = 0;
	if (persistent) { unsigned epoch; if (!cluster_sync_of(stream_context, 0, &epoch, &timeout_word)) return CCV_NNC_EXEC_OOM; }
	// the one-launch kernels' workgroups wait for each other: this command's launches are one TURN of the device's spinning launches (common.h) -- another
	// stream's persistent or cluster kernel can no longer hold part of the CUs while this one waits for the rest
	std::optional<ClusterTurn> turn;
	if (persistent) turn.emplace(stream_context);
	if (xs) { const int ret = lstm_lens(xs, g, lens, stream); if (ret != CCV_NNC_EXEC_SUCCESS) return ret; }
	const float* xin = x->data.f32;
	if (g.batch_first) { hipLaunchKernelGGL(lstm_swap01_kernel, dim3(blocks_of(TB * g.I)), dim3(256), 0, stream, x->data.f32, xseq, g.B, g.T, g.I); xin = xseq; }
	float* const rsv = train ? (float*)r->data.u8 : 0;
	const lstm_view_t view = lstm_view(g, rsv);
	const float* const W = w->data.f32;
	const dim3 step_grid((g.H + 63) / 64, (g.B + LS_ROWS - 1) / LS_ROWS), proj_grid((g.P + 63) / 64, (g.B + LS_ROWS - 1) / LS_ROWS);
	for (int l = 0; l < g.L; l++) {
		const int in = g.in_of(l);
		float* const yl = l == g.L - 1 ? (g.batch_first ? yseq : y->data.f32) : lay[l & 1];
		for (int d = 0; d < g.D; d++) {
			const int p = l * g.D + d;
			const float* const Wc = W + g.w_off(p);
			const float* const Rc = Wc + (size_t)4 * g.H * in;
			const float* const Wp = Rc + (size_t)4 * g.H * g.P;
			const float* const bw = g.bias ? W + g.b_off(p) : 0;
			// the input half of every step's gates: gx[T B][4H] = X W^T, one contraction on the matrix cores
			const MatOperand A = { xin, in, 1, (int)TB, in };
			const MatOperand Bm = { Wc, in, 1, 4 * g.H, in };
			const GemmOut out = { gx, 4L * g.H, 1, 0, 1.f, 0, 0 };
			const int ret = gemm_strided<float>("lstm_gx", A, Bm, out, 1, 0, 0, 0, 0, 0, stream_context);
			if (ret != CCV_NNC_EXEC_SUCCESS) return ret;
			if (!persistent && !rows) hipLaunchKernelGGL(lstm_transpose_kernel, dim3(blocks_of((size_t)4 * g.H * g.P)), dim3(256), 0, stream, Rc, rt, 4 * g.H, g.P);
			if (g.proj) hipLaunchKernelGGL(lstm_transpose_kernel, dim3(blocks_of((size_t)g.P * g.H)), dim3(256), 0, stream, Wp, wpt, g.P, g.H);
			if (rows) { // one launch for the whole sequence, nothing passes between its workgroups (lstm_rows_forw_kernel)
				const lstm_seq_t a = { gx, Rc, bw, hx ? hx->data.f32 + (size_t)p * g.BH() : 0, cx ? cx->data.f32 + (size_t)p * g.BH() : 0, yl + (size_t)d * g.P, hy ? hy->data.f32 + (size_t)p * g.BH() : 0, cy ? cy->data.f32 + (size_t)p * g.BH() : 0,
					rsv, 0, lens, 0, g.T, g.B, g.H, d, DP, g.S, rsv ? g.slot(p, 0, 0) : 0, rsv && g.T > 1 ? g.cslot(p, 0) : 0 };
				const int rb = (g.H & 3) == 0 && tune(TUNE_LSTM_ROWS) != 2 ? 1 : 2; // rows per workgroup (tuning key LSTM_ROWS = 2: the two-row form for every size)
				const dim3 rows_grid((g.B + rb - 1) / rb);
				if (rb == 1) {
					if (g.H <= 32) hipLaunchKernelGGL((lstm_rows_forw_kernel<32, 1>), rows_grid, dim3(128), 0, stream, a);
					else if (g.H <= 64) hipLaunchKernelGGL((lstm_rows_forw_kernel<64, 1>), rows_grid, dim3(256), 0, stream, a);
					else if (g.H <= 96) hipLaunchKernelGGL((lstm_rows_forw_kernel<96, 1>), rows_grid, dim3(384), 0, stream, a);
					else hipLaunchKernelGGL((lstm_rows_forw_kernel<128, 1>), rows_grid, dim3(512), 0, stream, a);
				} else {
					if (g.H <= 32) hipLaunchKernelGGL((lstm_rows_forw_kernel<32, 2>), rows_grid, dim3(128), 0, stream, a);
					else if (g.H <= 64) hipLaunchKernelGGL((lstm_rows_forw_kernel<64, 2>), rows_grid, dim3(256), 0, stream, a);
					else if (g.H <= 96) hipLaunchKernelGGL((lstm_rows_forw_kernel<96, 2>), rows_grid, dim3(384), 0, stream, a);
					else hipLaunchKernelGGL((lstm_rows_forw_kernel<128, 2>), rows_grid, dim3(512), 0, stream, a);
				}
				HIP_ENFORCE(hipGetLastError());
				note_kernel("lstm_rows_forw");
				continue;
			}
			if (persistent) { // one launch for the whole sequence (lstm_seq_forw_kernel)
				HIP_ENFORCE(hipMemsetAsync(xch, 0, n_xch, stream));
				const lstm_seq_t a = { gx, Rc, bw, hx ? hx->data.f32 + (size_t)p * g.BH() : 0, cx ? cx->data.f32 + (size_t)p * g.BH() : 0, yl + (size_t)d * g.P, hy ? hy->data.f32 + (size_t)p * g.BH() : 0, cy ? cy->data.f32 + (size_t)p * g.BH() : 0,
					rsv, xch, lens, timeout_word, g.T, g.B, g.H, d, DP, g.S, rsv ? g.slot(p, 0, 0) : 0, rsv && g.T > 1 ? g.cslot(p, 0) : 0 };
				const dim3 seq_grid((g.H + 15) / 16, (g.B + 15) / 16);
				if (g.H <= 128) NNC_LAUNCH_CONCURRENT(lstm_seq_forw_kernel<32>, seq_grid, dim3(256), LSTM_SEQ_LDS(32), stream, a);
				else if (g.H <= 256) NNC_LAUNCH_CONCURRENT(lstm_seq_forw_kernel<64>, seq_grid, dim3(256), LSTM_SEQ_LDS(64), stream, a);
				else NNC_LAUNCH_CONCURRENT(lstm_seq_forw_kernel<128>, seq_grid, dim3(256), LSTM_SEQ_LDS(128), stream, a);
				HIP_ENFORCE(hipGetLastError());
				note_kernel("lstm_seq_forw");
				continue;
			}
			note_kernel("lstm_step_forw");
			if (hx) HIP_ENFORCE(hipMemcpyAsync(hs[0], hx->data.f32 + (size_t)p * g.B * g.P, sizeof(float) * g.B * g.P, hipMemcpyDeviceToDevice, stream));
			else HIP_ENFORCE(hipMemsetAsync(hs[0], 0, sizeof(float) * g.B * g.P, stream));
			if (cx) HIP_ENFORCE(hipMemcpyAsync(cs[0], cx->data.f32 + (size_t)p * g.BH(), sizeof(float) * g.BH(), hipMemcpyDeviceToDevice, stream));
			else HIP_ENFORCE(hipMemsetAsync(cs[0], 0, sizeof(float) * g.BH(), stream));
			for (int s = 0; s < g.T; s++) {
				const int t = d ? g.T - 1 - s : s;
				float* const yt = yl + (size_t)t * g.B * DP + (size_t)d * g.P;
				float* const gates = rsv ? rsv + g.slot(p, s, 0) : 0;
				float* const cstore = rsv && s < g.T - 1 ? rsv + g.cslot(p, s) : 0;
				hipLaunchKernelGGL(lstm_step_forw_kernel, step_grid, dim3(256), 0, stream, (const float*)(gx + (size_t)t * g.B * 4 * g.H), (const float*)rt, bw, bw ? bw + 4 * g.H : (const float*)0, (const float*)hs[s & 1], (const float*)cs[s & 1],
					g.proj ? hraw : hs[(s + 1) & 1], cs[(s + 1) & 1], gates, cstore, g.proj ? (float*)0 : yt, DP, (const int*)lens, t, g.B, g.H, g.P, g.proj ? 0 : 1);
				if (g.proj)
					hipLaunchKernelGGL(lstm_rowmat_kernel<0>, proj_grid, dim3(256), 0, stream, (const float*)hraw, g.H, (const float*)wpt, g.P, g.H, g.P, g.B, hs[(s + 1) & 1], g.P, (const float*)hs[s & 1], yt, DP, gates ? gates + 5 * g.BH() : (float*)0, g.H, (const int*)lens, t);
			}
			HIP_ENFORCE(hipGetLastError());
			if (hy) HIP_ENFORCE(hipMemcpyAsync(hy->data.f32 + (size_t)p * g.B * g.P, hs[g.T & 1], sizeof(float) * g.B * g.P, hipMemcpyDeviceToDevice, stream));
			if (cy) HIP_ENFORCE(hipMemcpyAsync(cy->data.f32 + (size_t)p * g.BH(), cs[g.T & 1], sizeof(float) * g.BH(), hipMemcpyDeviceToDevice, stream));
		}
		if (l < g.L - 1 && g.dropout > 0.f)
			hipLaunchKernelGGL(lstm_dropout_kernel<true>, dim3(blocks_of(TB * DP)), dim3(256), 0, stream, view, rsv, l, yl, lstm_seed(stream_context), capture_tick_of(stream), g.dropout, 1.f / (1.f - g.dropout));
		xin = yl;
	}
	if (g.batch_first) hipLaunchKernelGGL(lstm_swap01_kernel, dim3(blocks_of(TB * DP)), dim3(256), 0, stream, (const float*)yseq, y->data.f32, g.T, g.B, DP);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}

// inputs: dy, [dhy], [dcy], [dr], x, [xs], [hx], [cx], w, y, [hy], [cy], r     outputs: dx, [dxs], [dhx], [dcx], [dw]     (ccv_nnc_lstm.c:19-33)
static int _lstm_back(EXEC_ARGS_L)
{
	if (input_size < 13 || output_size < 1 || !inputs[0] || !inputs[8] || !inputs[12] || !outputs[0]) return CCV_NNC_EXEC_INVALID;
	const ccv_nnc_tensor_t* const dy = inputs[0];
	const ccv_nnc_tensor_t* const dhy = inputs[1];
	const ccv_nnc_tensor_t* const dcy = inputs[2];
	const ccv_nnc_tensor_t* const x = inputs[4];
	const ccv_nnc_tensor_t* const xs = inputs[5];
	const ccv_nnc_tensor_t* const hx = inputs[6];
	const ccv_nnc_tensor_t* const cx = inputs[7];
	const ccv_nnc_tensor_t* const w = inputs[8];
	const ccv_nnc_tensor_t* const r = inputs[12];
	ccv_nnc_tensor_t* const dx = outputs[0];
	ccv_nnc_tensor_t* const dhx = output_size > 2 ? outputs[2] : 0;
	ccv_nnc_tensor_t* const dcx = output_size > 3 ? outputs[3] : 0;
	ccv_nnc_tensor_t* const dw = output_size > 4 ? outputs[4] : 0;
	lstm_geom_t g;
	if (!lstm_geometry(cmd, dx, &g) || g.is_test) return CCV_NNC_EXEC_INVALID;
	if (dw && !x) return CCV_NNC_EXEC_INVALID;
	const size_t TB = (size_t)g.T * g.B, LD = (size_t)g.L * g.D;
	const int DP = g.D * g.P;
	if (!dense_f32(dy, TB * DP) || !dense_f32(dx, TB * g.I) || !dense_f32(x, TB * g.I) || !dense_f32(w, g.w_total()) || !dense_f32(dw, g.w_total()) || !dense_reserve(r, g.reserve_total())) return CCV_NNC_EXEC_INVALID;
	if (!dense_f32(dhy, LD * g.B * g.P) || !dense_f32(dcy, LD * g.B * g.H) || !dense_f32(hx, LD * g.B * g.P) || !dense_f32(cx, LD * g.B * g.H) || !dense_f32(dhx, LD * g.B * g.P) || !dense_f32(dcx, LD * g.B * g.H)) return CCV_NNC_EXEC_INVALID;
	MarkerScope marker(cmd.cmd);
	hipStream_t stream = stream_of(stream_context);
	const int in_max = g.I > DP ? g.I : DP;
	// scratch: [ lengths | x, dy, dx in sequence-major order (batch-first) | two layer gradients | dG | the layer's input | h' | dh x 2 | dc | a zero cell state | projection: dhp (all steps), dh of o tanh(c), o tanh(c) (all steps) ]
	const size_t n_len = (sizeof(int) * g.B + 255) & ~(size_t)255, n_xs = g.batch_first ? al(TB * g.I) : 0, n_dys = g.batch_first ? al(TB * DP) : 0, n_lay = g.L > 1 ? al(TB * DP) : 0;
	const size_t n_dg = al(TB * 4 * g.H), n_in = g.L > 1 ? al(TB * in_max) : 0, n_hp = al(TB * g.P), n_h = al((size_t)g.B * g.P), n_c = al(g.BH());
	const size_t n_dhp = g.proj ? al(TB * g.P) : 0, n_draw = g.proj ? al(g.BH()) : 0, n_raw = g.proj ? al(TB * g.H) : 0;
	const bool rows = tune(TUNE_LSTM_PERSISTENT) && tune(TUNE_LSTM_ROWS) && !g.proj && g.H <= 128; // (lstm_rows_back_kernel: as the forward command)
	const bool persistent = !rows && tune(TUNE_LSTM_PERSISTENT) && !g.proj && g.H <= 512 && (long)((g.H + 15) / 16) * ((g.B + 15) / 16) <= device_cu_count();
	const size_t n_xch = persistent ? (sizeof(unsigned long long) * 2 * g.B * 4 * g.H + 255) & ~(size_t)255 : 0;
	WorkspaceScope ws(stream_context, n_len + 2 * n_xs + n_dys + 2 * n_lay + n_dg + n_in + n_hp + 2 * n_h + 2 * n_c + n_dhp + n_draw + n_raw + n_xch, lstm_inner_bytes(g));
	char* at = (char*)ws.prefix();
	if (!at) return CCV_NNC_EXEC_OOM;
	int* const lens = xs ? (int*)at : 0; at += n_len;
	float* const xseq = (float*)at; at += n_xs;
	float* const dxseq = (float*)at; at += n_xs;
	float* const dyseq = (float*)at; at += n_dys;
	float* const lay[2] = { (float*)at, (float*)(at + n_lay) }; at += 2 * n_lay;
	float* const dG = (float*)at; at += n_dg;
	float* const xl = (float*)at; at += n_in;
	float* const hp = (float*)at; at += n_hp;
	float* const dh[2] = { (float*)at, (float*)(at + n_h) }; at += 2 * n_h;
	float* const dc = (float*)at; at += n_c;
	float* const czero = (float*)at; at += n_c; // the cell state a sequence starts from when the host passes none
	float* const dhp = (float*)at; at += n_dhp;
	float* const draw = (float*)at; at += n_draw;
	float* const raw = (float*)at; at += n_raw;
	unsigned long long* const xch = (unsigned long long*)at;
	unsigned* timeout_word = 0;
	if (persistent) { unsigned epoch; if (!cluster_sync_of(stream_context, 0, &epoch, &timeout_word)) return CCV_NNC_EXEC_OOM; }
	// the one-launch kernels' workgroups wait for each other: this command's launches are one TURN of the device's spinning launches (common.h) -- another
	// stream's persistent or cluster kernel can no longer hold part of the CUs while this one waits for the rest
	std::optional<ClusterTurn> turn;
	if (persistent) turn.emplace(stream_context);
	if (xs) { const int ret = lstm_lens(xs, g, lens, stream); if (ret != CCV_NNC_EXEC_SUCCESS) return ret; }
	if (!cx) HIP_ENFORCE(hipMemsetAsync(czero, 0, sizeof(float) * g.BH(), stream));
	const float* x0 = x ? x->data.f32 : 0;
	const float* dytop = dy->data.f32;
	if (g.batch_first) {
		if (x) { hipLaunchKernelGGL(lstm_swap01_kernel, dim3(blocks_of(TB * g.I)), dim3(256), 0, stream, x->data.f32, xseq, g.B, g.T, g.I); x0 = xseq; }
		hipLaunchKernelGGL(lstm_swap01_kernel, dim3(blocks_of(TB * DP)), dim3(256), 0, stream, dy->data.f32, dyseq, g.B, g.T, DP);
		dytop = dyseq;
	}
	float* const dx0 = g.batch_first ? dxseq : dx->data.f32;
	const lstm_view_t view = lstm_view(g, (float*)r->data.u8);
	const float* const rsv = (const float*)r->data.u8;
	const float* const W = w->data.f32;
	float* const DW = dw ? dw->data.f32 : 0;
	const dim3 rec_grid((g.P + 63) / 64, (g.B + LS_ROWS - 1) / LS_ROWS), raw_grid((g.H + 63) / 64, (g.B + LS_ROWS - 1) / LS_ROWS);
	for (int l = g.L - 1; l >= 0; l--) {
		const int in = g.in_of(l);
		const float* const dyl = l == g.L - 1 ? dytop : lay[l & 1];
		float* const dxl = l == 0 ? dx0 : lay[(l - 1) & 1];
		const float* xin = x0;
		if (l > 0) { hipLaunchKernelGGL(lstm_gather_kernel<0>, dim3(blocks_of(TB * DP)), dim3(256), 0, stream, view, l - 1, 0, (const float*)0, (const int*)lens, xl); xin = xl; }
		for (int d = 0; d < g.D; d++) {
			const int p = l * g.D + d;
			const size_t wo = g.w_off(p);
			const float* const Wc = W + wo;
			const float* const Rc = Wc + (size_t)4 * g.H * in;
			const float* const Wp = Rc + (size_t)4 * g.H * g.P;
			if (!persistent && !rows) {
				if (dhy) HIP_ENFORCE(hipMemcpyAsync(dh[g.T & 1], dhy->data.f32 + (size_t)p * g.B * g.P, sizeof(float) * g.B * g.P, hipMemcpyDeviceToDevice, stream));
				else HIP_ENFORCE(hipMemsetAsync(dh[g.T & 1], 0, sizeof(float) * g.B * g.P, stream));
				if (dcy) HIP_ENFORCE(hipMemcpyAsync(dc, dcy->data.f32 + (size_t)p * g.BH(), sizeof(float) * g.BH(), hipMemcpyDeviceToDevice, stream));
				else HIP_ENFORCE(hipMemsetAsync(dc, 0, sizeof(float) * g.BH(), stream));
			}
			if (rows) { // one launch for the whole sequence, nothing passes between its workgroups (lstm_rows_back_kernel)
				const lstm_seq_back_t a = { Rc, rsv, cx ? cx->data.f32 + (size_t)p * g.BH() : 0, dyl + (size_t)d * g.P, dhy ? dhy->data.f32 + (size_t)p * g.BH() : 0, dcy ? dcy->data.f32 + (size_t)p * g.BH() : 0,
					dG, dhx ? dhx->data.f32 + (size_t)p * g.BH() : 0, dcx ? dcx->data.f32 + (size_t)p * g.BH() : 0, 0, lens, 0, g.T, g.B, g.H, d, DP, g.S, g.slot(p, 0, 0), g.T > 1 ? g.cslot(p, 0) : 0 };
				const int rb = (g.H & 3) == 0 && tune(TUNE_LSTM_ROWS) != 2 ? 1 : 2; // (as the forward command)
				const dim3 rows_grid((g.B + rb - 1) / rb);
				if (rb == 1) {
					if (g.H <= 32) hipLaunchKernelGGL((lstm_rows_back_kernel<32, 1>), rows_grid, dim3(128), 0, stream, a);
					else if (g.H <= 64) hipLaunchKernelGGL((lstm_rows_back_kernel<64, 1>), rows_grid, dim3(256), 0, stream, a);
					else if (g.H <= 96) hipLaunchKernelGGL((lstm_rows_back_kernel<96, 1>), rows_grid, dim3(384), 0, stream, a);
					else hipLaunchKernelGGL((lstm_rows_back_kernel<128, 1>), rows_grid, dim3(512), 0, stream, a);
				} else {
					if (g.H <= 32) hipLaunchKernelGGL((lstm_rows_back_kernel<32, 2>), rows_grid, dim3(128), 0, stream, a);
					else if (g.H <= 64) hipLaunchKernelGGL((lstm_rows_back_kernel<64, 2>), rows_grid, dim3(256), 0, stream, a);
					else if (g.H <= 96) hipLaunchKernelGGL((lstm_rows_back_kernel<96, 2>), rows_grid, dim3(384), 0, stream, a);
					else hipLaunchKernelGGL((lstm_rows_back_kernel<128, 2>), rows_grid, dim3(512), 0, stream, a);
				}
				HIP_ENFORCE(hipGetLastError());
			} else
			if (persistent) { // one launch for the whole sequence (lstm_seq_back_kernel)
				HIP_ENFORCE(hipMemsetAsync(xch, 0, n_xch, stream));
				const lstm_seq_back_t a = { Rc, rsv, cx ? cx->data.f32 + (size_t)p * g.BH() : 0, dyl + (size_t)d * g.P, dhy ? dhy->data.f32 + (size_t)p * g.BH() : 0, dcy ? dcy->data.f32 + (size_t)p * g.BH() : 0,
					dG, dhx ? dhx->data.f32 + (size_t)p * g.BH() : 0, dcx ? dcx->data.f32 + (size_t)p * g.BH() : 0, xch, lens, timeout_word, g.T, g.B, g.H, d, DP, g.S, g.slot(p, 0, 0), g.T > 1 ? g.cslot(p, 0) : 0 };
				const dim3 seq_grid((g.H + 15) / 16, (g.B + 15) / 16);
				if (g.H <= 128) NNC_LAUNCH_CONCURRENT(lstm_seq_back_kernel<1>, seq_grid, dim3(256), LSTM_SEQ_BACK_LDS, stream, a);
				else if (g.H <= 256) NNC_LAUNCH_CONCURRENT(lstm_seq_back_kernel<2>, seq_grid, dim3(256), LSTM_SEQ_BACK_LDS, stream, a);
				else if (g.H <= 384) NNC_LAUNCH_CONCURRENT(lstm_seq_back_kernel<3>, seq_grid, dim3(256), LSTM_SEQ_BACK_LDS, stream, a);
				else NNC_LAUNCH_CONCURRENT(lstm_seq_back_kernel<4>, seq_grid, dim3(256), LSTM_SEQ_BACK_LDS, stream, a);
				HIP_ENFORCE(hipGetLastError());
			} else
			for (int s = g.T - 1; s >= 0; s--) {
				const int t = d ? g.T - 1 - s : s;
				const float* const dyt = dyl + (size_t)t * g.B * DP + (size_t)d * g.P;
				const float* const cprev = s == 0 ? (cx ? cx->data.f32 + (size_t)p * g.BH() : (const float*)czero) : rsv + g.cslot(p, s - 1);
				float* const dgt = dG + (size_t)t * g.B * 4 * g.H;
				const float* const dh_in = dh[(s + 1) & 1];
				if (g.proj) {
					float* const dhpt = dhp + (size_t)t * g.B * g.P;
					hipLaunchKernelGGL(lstm_dhp_kernel, dim3(blocks_of((size_t)g.B * g.P)), dim3(256), 0, stream, dh_in, dyt, DP, dhpt, (const int*)lens, t, g.B, g.P);
					hipLaunchKernelGGL(lstm_rowmat_kernel<1>, raw_grid, dim3(256), 0, stream, (const float*)dhpt, g.P, Wp, g.H, g.P, g.H, g.B, draw, g.H, (const float*)0, (float*)0, 0, (float*)0, 0, (const int*)lens, t);
					hipLaunchKernelGGL(lstm_step_back_kernel, dim3(blocks_of(g.BH())), dim3(256), 0, stream, rsv + g.slot(p, s, 0), cprev, (const float*)draw, (const float*)0, 0, dc, dgt, (const int*)lens, t, g.B, g.H);
				} else
					hipLaunchKernelGGL(lstm_step_back_kernel, dim3(blocks_of(g.BH())), dim3(256), 0, stream, rsv + g.slot(p, s, 0), cprev, dh_in, dyt, DP, dc, dgt, (const int*)lens, t, g.B, g.H);
				// the state gradient the step before receives: dG R (+ this one's own where the step was past the end)
				if (!lens && g.H >= 256) { // wide layers, every sequence full length: a [B][4H] x [4H][P] contraction worth the matrix cores (the lane-per-column kernel walks 4H terms one after the other)
					const MatOperand A = { dgt, 4L * g.H, 1, g.B, 4 * g.H };
					const MatOperand Bm = { Rc, 1, g.P, g.P, 4 * g.H };
					const GemmOut out = { dh[s & 1], g.P, 1, 0, 1.f, 0, 0 };
					const int ret = gemm_strided<float>("lstm_dh", A, Bm, out, 1, 0, 0, 0, 0, 0, stream_context);
					if (ret != CCV_NNC_EXEC_SUCCESS) return ret;
				} else
				hipLaunchKernelGGL(lstm_rowmat_kernel<2>, rec_grid, dim3(256), 0, stream, (const float*)dgt, 4 * g.H, Rc, g.P, 4 * g.H, g.P, g.B, dh[s & 1], g.P, dh_in, (float*)0, 0, (float*)0, 0, (const int*)lens, t);
			}
			HIP_ENFORCE(hipGetLastError());
			if (!persistent && !rows && dhx) HIP_ENFORCE(hipMemcpyAsync(dhx->data.f32 + (size_t)p * g.B * g.P, dh[0], sizeof(float) * g.B * g.P, hipMemcpyDeviceToDevice, stream));
			if (!persistent && !rows && dcx) HIP_ENFORCE(hipMemcpyAsync(dcx->data.f32 + (size_t)p * g.BH(), dc, sizeof(float) * g.BH(), hipMemcpyDeviceToDevice, stream));
			int ret;
			{ // dX (+)= dG W: [T B][4H] x [4H][in]; the second direction adds to the first
				const MatOperand A = { dG, 4L * g.H, 1, (int)TB, 4 * g.H };
				const MatOperand Bm = { Wc, 1, in, in, 4 * g.H };
				const GemmOut out = { dxl, in, 1, 0, 1.f, d > 0, 0 };
				if ((ret = gemm_strided<float>("lstm_dx", A, Bm, out, 1, 0, 0, 0, 0, 0, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
			}
			if (!DW) continue;
			{ // dW = dG^T X: [4H][T B] x [T B][in]
				const MatOperand A = { dG, 1, 4L * g.H, 4 * g.H, (int)TB };
				const MatOperand Bm = { xin, 1, in, in, (int)TB };
				const GemmOut out = { DW + wo, in, 1, 0, 1.f, 0, 0 };
				if ((ret = gemm_strided<float>("lstm_dw", A, Bm, out, 1, 0, 0, 0, 0, 0, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
			}
			{ // dR = dG^T H': the state each step started from
				hipLaunchKernelGGL(lstm_gather_kernel<1>, dim3(blocks_of(TB * g.P)), dim3(256), 0, stream, view, l, d, hx ? hx->data.f32 : (const float*)0, (const int*)lens, hp);
				const MatOperand A = { dG, 1, 4L * g.H, 4 * g.H, (int)TB };
				const MatOperand Bm = { hp, 1, g.P, g.P, (int)TB };
				const GemmOut out = { DW + wo + (size_t)4 * g.H * in, g.P, 1, 0, 1.f, 0, 0 };
				if ((ret = gemm_strided<float>("lstm_dr", A, Bm, out, 1, 0, 0, 0, 0, 0, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
			}
			if (g.proj) { // dW_p = dhp^T (o tanh(c)): [P][T B] x [T B][H]
				hipLaunchKernelGGL(lstm_gather_kernel<2>, dim3(blocks_of(TB * g.H)), dim3(256), 0, stream, view, l, d, (const float*)0, (const int*)lens, raw);
				const MatOperand A = { dhp, 1, g.P, g.P, (int)TB };
				const MatOperand Bm = { raw, 1, g.H, g.H, (int)TB };
				const GemmOut out = { DW + wo + (size_t)4 * g.H * in + (size_t)4 * g.H * g.P, g.H, 1, 0, 1.f, 0, 0 };
				if ((ret = gemm_strided<float>("lstm_dwp", A, Bm, out, 1, 0, 0, 0, 0, 0, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
			}
			if (g.bias) { // both bias sets see the same gate gradients
				float* const db = DW + g.b_off(p);
				if ((ret = colsum_f32(dG, (long)TB, 4 * g.H, 4L * g.H, db, 0, stream_context)) != CCV_NNC_EXEC_SUCCESS) return ret;
				HIP_ENFORCE(hipMemcpyAsync(db + 4 * g.H, db, sizeof(float) * 4 * g.H, hipMemcpyDeviceToDevice, stream));
			}
		}
		if (l > 0 && g.dropout > 0.f) // what layer l - 1 handed up was scaled: so is its gradient
			hipLaunchKernelGGL(lstm_dropout_kernel<false>, dim3(blocks_of(TB * DP)), dim3(256), 0, stream, view, (float*)0, l - 1, dxl, 0u, (const unsigned*)0, 0.f, 0.f);
	}
	if (g.batch_first) hipLaunchKernelGGL(lstm_swap01_kernel, dim3(blocks_of(TB * g.I)), dim3(256), 0, stream, (const float*)dxseq, dx->data.f32, g.T, g.B, g.I);
	HIP_ENFORCE(hipGetLastError());
	note_kernel(rows ? "lstm_rows_back" : (persistent ? "lstm_seq_back" : "lstm_step_back"));
	return CCV_NNC_EXEC_SUCCESS;
}

// registry->aux of both rows: the bytes of reserved space the host must hand to the forward command (ccv_nnc_lstm.c:35,64-71 sizes output 3 with it)
static size_t _lstm_reserve_space_size(const ccv_nnc_cmd_t cmd, const int datatype, const int feature_size, const int batch_count, const int max_seq_count)
{
	if (cmd.info.rnn.is_test) return 0;
	lstm_geom_t g;
	memset(&g, 0, sizeof(g));
	g.T = max_seq_count; g.B = batch_count; g.I = feature_size; g.H = cmd.info.rnn.hidden_size; g.P = cmd.info.rnn.proj_size == 0 ? g.H : cmd.info.rnn.proj_size;
	g.L = cmd.info.rnn.num_layers; g.D = cmd.info.rnn.bidirectional ? 2 : 1;
	g.dropout = g.L < 2 ? 0.f : cmd.info.rnn.dropout;
	g.proj = g.P != g.H;
	g.S = 5 + g.proj + (g.dropout > 0.f ? 1 : 0);
	// (CCV_16F: the tape stays fp32 planes inside the half tensor -- dense_reserve above -- so the host is asked for the fp32 bytes)
	return g.reserve_total() * (CCV_GET_DATA_TYPE(datatype) == CCV_16F ? sizeof(float) : datatype_size(datatype));
}

} // namespace

extern "C" size_t nnc_mi355x_lstm_reserve_space_size(const ccv_nnc_cmd_t cmd, const int datatype, const int feature_size, const int batch_count, const int max_seq_count)
{
	return _lstm_reserve_space_size(cmd, datatype, feature_size, batch_count, max_seq_count);
}

#define NNC_REG_LSTM(CMD, EXEC) \
	extern "C" void _register_command_##CMD##_backend_CCV_NNC_BACKEND_GPU_CUDNN(ccv_nnc_cmd_backend_registry_t* const registry) \
	{ registry->tensor_formats = CCV_TENSOR_FORMAT_NCHW | CCV_TENSOR_FORMAT_NHWC; registry->tensor_datatypes = CCV_32F | CCV_32S; registry->tensor_memory = CCV_TENSOR_GPU_MEMORY | CCV_TENSOR_CPU_MEMORY; registry->algorithms = 1; \
	  registry->exec = EXEC; registry->aux = (void*)_lstm_reserve_space_size; NNC_HALF_STAGED(registry, EXEC); }
NNC_REG_LSTM(CCV_NNC_LSTM_FORWARD, _lstm_forw)
NNC_REG_LSTM(CCV_NNC_LSTM_BACKWARD, _lstm_back)
