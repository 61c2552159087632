// INDEX_SELECT (embedding gather / scatter-add) and PAD (zero / replicate, backward = crop) on gfx950.
// Oracle semantics:
//   index select  lib/nnc/cmd/index/ccv_nnc_index_select_cpu_ref.c:13-98   b[i][:] = a[indices[i]][:] (int32 indices; fp32 indices
//                 interpolate between rows j and j+1); backward h = 0, then h[indices[i]][:] += g[i][:] in row order
//   pad           lib/nnc/cmd/pad/ccv_nnc_pad_cpu_ref.c:13-140              begin = cmd.info.size.dim[], end = cmd.info.pad.end[]
// Both are HBM-bound copies.  The scatter-add keeps the reference's summation order (rows visited in order per column) so that
// repeated indices give bit-identical sums: one thread per column walks the rows; columns are the parallel dimension.
#include "common.h"

using namespace nnc;

namespace {

struct dims4_t { int d[4]; long s[4]; };
static bool dims4(const ccv_nnc_tensor_t* t, dims4_t* o)
{ // dims right-aligned to 4 with element strides
	const int nd = tensor_nd(t->info.dim);
	if (nd > 4) return false;
	int st[CCV_NNC_MAX_DIM_ALLOC];
	tensor_strides(t, st);
	for (int k = 0; k < 4; k++) {
		const int j = k - (4 - nd);
		o->d[k] = j >= 0 ? t->info.dim[j] : 1;
		o->s[k] = j >= 0 ? st[j] : 0;
	}
	return true;
}

// ---- index select -----------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) index_select_i32_kernel(const float* a, const int* idx, float* b, const int rows, const int cols, const long a_inc, const long b_inc)
{
	const size_t n = (size_t)rows * cols, stride = (size_t)gridDim.x * blockDim.x;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
		const size_t r = i / cols, c = i - r * cols;
		b[r * b_inc + c] = a[(long)idx[r] * a_inc + c];
	}
}
__global__ void __launch_bounds__(256) index_select_f32_kernel(const float* a, const float* idx, float* b, const int rows, const int cols, const long a_inc, const long b_inc, const int a_rows)
{
	const size_t n = (size_t)rows * cols, stride = (size_t)gridDim.x * blockDim.x;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
		const size_t r = i / cols, c = i - r * cols;
		const float f = idx[r];
		const int j0 = (int)f;
		const int j1 = j0 + 1 < a_rows - 1 ? j0 + 1 : a_rows - 1;
		const float w1 = f - j0, w0 = 1.f - w1;
		b[r * b_inc + c] = a[(long)j0 * a_inc + c] * w0 + a[(long)j1 * a_inc + c] * w1;
	}
}
__global__ void __launch_bounds__(256) index_scatter_add_kernel(const float* g, const int* idx, float* h, const int g_rows, const int cols, const long g_inc, const long h_inc)
{
	const int c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= cols) return;
	for (int i = 0; i < g_rows; i++) h[(long)idx[i] * h_inc + c] += g[(long)i * g_inc + c];
}
__global__ void __launch_bounds__(256) zero_rows_kernel(float* h, const int rows, const int cols, const long h_inc)
{
	const size_t n = (size_t)rows * cols, stride = (size_t)gridDim.x * blockDim.x;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
		const size_t r = i / cols;
		h[r * h_inc + (i - r * cols)] = 0.f;
	}
}
static void rows_cols(const ccv_nnc_tensor_t* t, int* rows, int* cols, long* inc)
{ // index_select_cpu_ref.c:22-27: dim[0] rows, dim[1] columns (1-d: one column), row increment = stride[0] for views
	const int nd = tensor_nd(t->info.dim);
	*rows = t->info.dim[0];
	*cols = nd < 2 ? 1 : t->info.dim[1];
	*inc = *cols;
	if (CCV_IS_TENSOR_VIEW(t) && nd >= 2) *inc = ((const ccv_nnc_tensor_view_t*)t)->stride[0];
}
#define EXEC_ARGS const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context

static int _index_select_forw(EXEC_ARGS)
{
	if (input_size < 2 || output_size < 1 || !inputs[0] || !inputs[1] || !outputs[0]) return CCV_NNC_EXEC_INVALID;
	const ccv_nnc_tensor_t* a = inputs[0];
	const ccv_nnc_tensor_t* ind = inputs[1];
	ccv_nnc_tensor_t* b = outputs[0];
	const int dt = CCV_GET_DATA_TYPE(a->info.datatype), it = CCV_GET_DATA_TYPE(ind->info.datatype);
	if ((dt != CCV_32F && dt != CCV_32S) || tensor_nd(a->info.dim) > 2 || tensor_nd(b->info.dim) > 2 || !tensor_contiguous(ind)) return CCV_NNC_EXEC_INVALID;
	int a_rows, a_cols, b_rows, b_cols;
	long a_inc, b_inc;
	rows_cols(a, &a_rows, &a_cols, &a_inc);
	rows_cols(b, &b_rows, &b_cols, &b_inc);
	if (a_cols != b_cols || tensor_count(ind->info) != (size_t)b_rows) return CCV_NNC_EXEC_INVALID;
	if (b_rows == 0 || b_cols == 0) return CCV_NNC_EXEC_SUCCESS;
	const unsigned grid = grid_for((size_t)b_rows * b_cols, 256);
	// int32 payloads move as raw 32-bit words through the same kernel
	if (it == CCV_32S) hipLaunchKernelGGL(index_select_i32_kernel, dim3(grid), dim3(256), 0, stream_of(stream_context), (const float*)a->data.f32, (const int*)ind->data.i32, b->data.f32, b_rows, b_cols, a_inc, b_inc);
	else if (it == CCV_32F && dt == CCV_32F) hipLaunchKernelGGL(index_select_f32_kernel, dim3(grid), dim3(256), 0, stream_of(stream_context), (const float*)a->data.f32, (const float*)ind->data.f32, b->data.f32, b_rows, b_cols, a_inc, b_inc, a_rows);
	else return CCV_NNC_EXEC_INVALID;
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}
static int _index_select_back(EXEC_ARGS)
{ // (g, _, indices) -> h (, _)
	if (input_size < 3 || output_size < 1 || !inputs[0] || !inputs[2] || !outputs[0]) return CCV_NNC_EXEC_INVALID;
	const ccv_nnc_tensor_t* g = inputs[0];
	const ccv_nnc_tensor_t* ind = inputs[2];
	ccv_nnc_tensor_t* h = outputs[0];
	if (CCV_GET_DATA_TYPE(g->info.datatype) != CCV_32F || CCV_GET_DATA_TYPE(ind->info.datatype) != CCV_32S || tensor_nd(g->info.dim) > 2 || tensor_nd(h->info.dim) > 2 || !tensor_contiguous(ind)) return CCV_NNC_EXEC_INVALID;
	int g_rows, g_cols, h_rows, h_cols;
	long g_inc, h_inc;
	rows_cols(g, &g_rows, &g_cols, &g_inc);
	rows_cols(h, &h_rows, &h_cols, &h_inc);
	if (g_cols != h_cols || tensor_count(ind->info) != (size_t)g_rows) return CCV_NNC_EXEC_INVALID;
	hipStream_t stream = stream_of(stream_context);
	if (h_rows > 0 && h_cols > 0) hipLaunchKernelGGL(zero_rows_kernel, dim3(grid_for((size_t)h_rows * h_cols, 256)), dim3(256), 0, stream, h->data.f32, h_rows, h_cols, h_inc);
	if (output_size >= 2 && outputs[1] && tensor_contiguous(outputs[1])) HIP_ENFORCE(hipMemsetAsync(outputs[1]->data.u8, 0, tensor_count(outputs[1]->info) * datatype_size(CCV_GET_DATA_TYPE(outputs[1]->info.datatype)), stream));
	if (g_rows > 0 && g_cols > 0) hipLaunchKernelGGL(index_scatter_add_kernel, dim3((g_cols + 255) / 256), dim3(256), 0, stream, (const float*)g->data.f32, (const int*)ind->data.i32, h->data.f32, g_rows, g_cols, g_inc, h_inc);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}

// ---- pad ----------------------------------------------------------------------------------------------------------------------------
enum { PAD_ZERO = 0, PAD_REPLICATE = 1 }; // CCV_NNC_PAD_* (ccv_nnc.h:98-99)
struct pad_args_t { int bd[4]; int ad[4]; int begin[4]; long sa[4], sb[4]; };
template <int TYPE>
__global__ void __launch_bounds__(256) pad_forw_kernel(const float* a, float* b, const pad_args_t m, const size_t n)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += stride) {
		size_t r = idx;
		int i[4];
		i[3] = (int)(r % m.bd[3]); r /= m.bd[3];
		i[2] = (int)(r % m.bd[2]); r /= m.bd[2];
		i[1] = (int)(r % m.bd[1]); r /= m.bd[1];
		i[0] = (int)r;
		long ao = 0, bo = 0;
		bool inside = true;
#pragma unroll
		for (int k = 0; k < 4; k++) {
			int j = i[k] - m.begin[k];
			if (TYPE == PAD_REPLICATE) j = j < 0 ? 0 : (j > m.ad[k] - 1 ? m.ad[k] - 1 : j);
			else inside = inside & (j >= 0) & (j < m.ad[k]);
			ao += (long)j * m.sa[k];
			bo += (long)i[k] * m.sb[k];
		}
		b[bo] = (TYPE == PAD_REPLICATE || inside) ? a[inside ? ao : 0] : 0.f;
	}
}
// h[i] = g[i + begin]
__global__ void __launch_bounds__(256) pad_back_kernel(const float* g, float* h, const pad_args_t m, const size_t n)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += stride) {
		size_t r = idx;
		int i[4];
		i[3] = (int)(r % m.ad[3]); r /= m.ad[3];
		i[2] = (int)(r % m.ad[2]); r /= m.ad[2];
		i[1] = (int)(r % m.ad[1]); r /= m.ad[1];
		i[0] = (int)r;
		long go = 0, ho = 0;
#pragma unroll
		for (int k = 0; k < 4; k++) { go += (long)(i[k] + m.begin[k]) * m.sb[k]; ho += (long)i[k] * m.sa[k]; }
		h[ho] = g[go];
	}
}
// small = the unpadded tensor (a / h), big = the padded one (b / g)
static int pad_args(const ccv_nnc_cmd_t& cmd, const ccv_nnc_tensor_t* small, const ccv_nnc_tensor_t* big, pad_args_t* m)
{
	dims4_t ss, sb;
	if (!dims4(small, &ss) || !dims4(big, &sb)) return 0;
	const int nd = tensor_nd(small->info.dim), offset = 4 - nd;
	for (int k = 0; k < 4; k++) {
		const int x = k - offset;
		m->begin[k] = x >= 0 ? cmd.info.size.dim[x] : 0;
		const int end = x >= 0 ? cmd.info.pad.end[x] : 0;
		if (m->begin[k] < 0 || end < 0 || sb.d[k] != ss.d[k] + m->begin[k] + end) return 0;
		m->ad[k] = ss.d[k]; m->bd[k] = sb.d[k]; m->sa[k] = ss.s[k]; m->sb[k] = sb.s[k];
	}
	return 1;
}
static int _pad_forw(EXEC_ARGS)
{
	if (input_size < 1 || output_size < 1 || !inputs[0] || !outputs[0] || CCV_GET_DATA_TYPE(inputs[0]->info.datatype) != CCV_32F) return CCV_NNC_EXEC_INVALID;
	pad_args_t m;
	if (!pad_args(cmd, inputs[0], outputs[0], &m)) return CCV_NNC_EXEC_INVALID;
	const size_t n = (size_t)m.bd[0] * m.bd[1] * m.bd[2] * m.bd[3];
	if (n == 0) return CCV_NNC_EXEC_SUCCESS;
	if (cmd.info.pad.type == PAD_ZERO) hipLaunchKernelGGL(HIP_KERNEL_NAME(pad_forw_kernel<PAD_ZERO>), dim3(grid_for(n, 256)), dim3(256), 0, stream_of(stream_context), (const float*)inputs[0]->data.f32, outputs[0]->data.f32, m, n);
	else if (cmd.info.pad.type == PAD_REPLICATE) hipLaunchKernelGGL(HIP_KERNEL_NAME(pad_forw_kernel<PAD_REPLICATE>), dim3(grid_for(n, 256)), dim3(256), 0, stream_of(stream_context), (const float*)inputs[0]->data.f32, outputs[0]->data.f32, m, n);
	else return CCV_NNC_EXEC_INVALID;
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}
static int _pad_back(EXEC_ARGS)
{ // g (padded shape) -> h (original shape): the crop (pad_cpu_ref.c:88-138)
	if (input_size < 1 || output_size < 1 || !inputs[0] || !outputs[0] || CCV_GET_DATA_TYPE(inputs[0]->info.datatype) != CCV_32F) return CCV_NNC_EXEC_INVALID;
	pad_args_t m;
	if (!pad_args(cmd, outputs[0], inputs[0], &m)) return CCV_NNC_EXEC_INVALID;
	const size_t n = (size_t)m.ad[0] * m.ad[1] * m.ad[2] * m.ad[3];
	if (n == 0) return CCV_NNC_EXEC_SUCCESS;
	hipLaunchKernelGGL(pad_back_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream_of(stream_context), (const float*)inputs[0]->data.f32, outputs[0]->data.f32, m, n);
	HIP_ENFORCE(hipGetLastError());
	return CCV_NNC_EXEC_SUCCESS;
}

} // namespace

#define NNC_REG(CMD, BACKEND, DATATYPES, EXEC) \
	extern "C" void _register_command_##CMD##_backend_##BACKEND(ccv_nnc_cmd_backend_registry_t* const registry) \
	{ registry->tensor_formats = CCV_TENSOR_FORMAT_NCHW | CCV_TENSOR_FORMAT_NHWC | CCV_TENSOR_FORMAT_CHWN; registry->tensor_datatypes = (DATATYPES); registry->tensor_memory = CCV_TENSOR_GPU_MEMORY; registry->algorithms = 1; registry->exec = EXEC; NNC_HALF_STAGED(registry, EXEC); }

NNC_REG(CCV_NNC_INDEX_SELECT_FORWARD, CCV_NNC_BACKEND_GPU_REF, CCV_32F | CCV_32S, _index_select_forw)
NNC_REG(CCV_NNC_INDEX_SELECT_BACKWARD, CCV_NNC_BACKEND_GPU_REF, CCV_32F | CCV_32S, _index_select_back)
NNC_REG(CCV_NNC_PAD_FORWARD, CCV_NNC_BACKEND_GPU_REF, CCV_32F, _pad_forw)
NNC_REG(CCV_NNC_PAD_BACKWARD, CCV_NNC_BACKEND_GPU_REF, CCV_32F, _pad_back)
