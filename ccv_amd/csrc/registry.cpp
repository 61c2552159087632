// Row table + standalone dispatch (mirrors ccv_nnc_cmd_exec / ccv_nnc_cmd_find_backend / ccv_nnc_cmd_ok,
// lib/nnc/ccv_nnc_cmd.c:117-131, 307-328, 651-693) for callers that do not link the reference host.
#include "common.h"
#include <mutex>

namespace {
struct row_t {
	uint32_t cmd, backend;
	const char* name;
	nnc::register_fn_t reg;
	ccv_nnc_cmd_backend_registry_t registry;
};
#define NNC_ROW(c, b) { (uint32_t)c, (uint32_t)b, #c "/" #b, _register_command_##c##_backend_##b, {} },
row_t g_rows[] = {
#include "../../include/nnc_mi355x_registry.def"
};
#undef NNC_ROW
constexpr int g_row_count = (int)(sizeof(g_rows) / sizeof(g_rows[0]));
std::once_flag g_once;
void init_rows()
{
	std::call_once(g_once, []() {
		for (int i = 0; i < g_row_count; i++) {
			memset(&g_rows[i].registry, 0, sizeof(g_rows[i].registry));
			g_rows[i].reg(&g_rows[i].registry);
		}
	});
}
// Backend preference order = backend_init_map order of the reference (ccv_nnc_cmd.inc: CUBLAS, CUDNN, NCCL, REF among GPU slots).
const row_t* find_row(uint32_t cmd, uint32_t backend, int memory, int formats, int datatypes)
{
	init_rows();
	for (int i = 0; i < g_row_count; i++) {
		const row_t& r = g_rows[i];
		if (r.cmd != cmd || !r.registry.exec) continue;
		if (backend != CCV_NNC_NO_BACKEND) { if (r.backend == backend) return &r; continue; }
		if ((r.registry.tensor_memory & memory) == memory && (r.registry.tensor_formats & formats) == formats && (r.registry.tensor_datatypes & datatypes) == datatypes) return &r;
	}
	return 0;
}
} // namespace

namespace nnc {
const char* command_row_name(const uint32_t cmd)
{
	for (int i = 0; i < g_row_count; i++)
		if (g_rows[i].cmd == cmd) return g_rows[i].name;
	return "CCV_NNC_(unregistered)";
}
}

extern "C" {

int nnc_mi355x_registry_count(void) { return g_row_count; }
int nnc_mi355x_registry_get(int i, uint32_t* cmd, uint32_t* backend, ccv_nnc_cmd_backend_registry_t* registry)
{
	if (i < 0 || i >= g_row_count) return -1;
	init_rows();
	*cmd = g_rows[i].cmd;
	*backend = g_rows[i].backend;
	*registry = g_rows[i].registry;
	return 0;
}
const char* nnc_mi355x_registry_name(int i) { return (i < 0 || i >= g_row_count) ? 0 : g_rows[i].name; }

int nnc_mi355x_cmd_ok(const uint32_t cmd, const uint32_t backend)
{
	if (cmd == CCV_NNC_NOOP) return 1;
	init_rows();
	for (int i = 0; i < g_row_count; i++)
		if (g_rows[i].cmd == cmd && g_rows[i].backend == backend) return !!g_rows[i].registry.exec;
	return 0;
}

int nnc_mi355x_cmd_exec(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* const inputs, const int input_size, ccv_nnc_tensor_t* const* const outputs, const int output_size, ccv_nnc_stream_context_t* const stream_context)
{
	if (cmd.cmd == CCV_NNC_NOOP) return CCV_NNC_EXEC_SUCCESS;
	int memory = 0, formats = 0, datatypes = 0, device = -1;
	for (int i = 0; i < input_size; i++)
		if (inputs[i]) {
			memory |= CCV_TENSOR_GET_MEMORY(inputs[i]->info.type), formats |= inputs[i]->info.format, datatypes |= CCV_GET_DATA_TYPE(inputs[i]->info.datatype);
			if (device < 0 && CCV_TENSOR_GET_MEMORY(inputs[i]->info.type) == CCV_TENSOR_GPU_MEMORY) device = CCV_TENSOR_GET_DEVICE_ID(inputs[i]->info.type);
		}
	for (int i = 0; i < output_size; i++)
		if (outputs[i]) {
			memory |= CCV_TENSOR_GET_MEMORY(outputs[i]->info.type), formats |= outputs[i]->info.format, datatypes |= CCV_GET_DATA_TYPE(outputs[i]->info.datatype);
			if (device < 0 && CCV_TENSOR_GET_MEMORY(outputs[i]->info.type) == CCV_TENSOR_GPU_MEMORY) device = CCV_TENSOR_GET_DEVICE_ID(outputs[i]->info.type);
		}
	// _ccv_nnc_cmd_set_device_id (ccv_nnc_cmd.c:332-342): without a stream, run on the tensors' device.
	if (!stream_context && device >= 0) nnc_mi355x_set_device(device);
	const row_t* r = find_row(cmd.cmd, cmd.backend, memory, formats, datatypes);
	if (!r) return CCV_NNC_EXEC_NO_KERNEL;
	return r->registry.exec(cmd, hint, flags, inputs, input_size, outputs, output_size, stream_context);
}

} // extern "C"
