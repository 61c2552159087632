/* Debug aid: LD_PRELOAD shim that logs every ccv_nnc_cmd_exec the reference host issues (command id, stream, tensor types). */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include "nnc_mi355x.h"
typedef int (*exec_f)(const ccv_nnc_cmd_t, const ccv_nnc_hint_t, const int, ccv_nnc_tensor_t* const*, const int, ccv_nnc_tensor_t* const*, const int, ccv_nnc_stream_context_t*);
int ccv_nnc_cmd_exec(const ccv_nnc_cmd_t cmd, const ccv_nnc_hint_t hint, const int flags, ccv_nnc_tensor_t* const* inputs, const int n, ccv_nnc_tensor_t* const* outputs, const int m, ccv_nnc_stream_context_t* s)
{
	static exec_f real; if (!real) real = (exec_f)dlsym(RTLD_NEXT, "ccv_nnc_cmd_exec");
	fprintf(stderr, "EXEC cmd %08x stream %p flags %d in %d out %d:", cmd.cmd, (void*)s, flags, n, m);
	for (int i = 0; i < n; i++) if (inputs[i]) fprintf(stderr, " i[%p mem=%x]", (void*)inputs[i]->data.u8, inputs[i]->info.type); else fprintf(stderr, " i[null]");
	for (int i = 0; i < m; i++) if (outputs[i]) fprintf(stderr, " o[%p mem=%x]", (void*)outputs[i]->data.u8, outputs[i]->info.type); else fprintf(stderr, " o[null]");
	fprintf(stderr, "\n");
	return real(cmd, hint, flags, inputs, n, outputs, m, s);
}
