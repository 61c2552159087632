/* Backend registration entry points of libnnc_mi355x.so: one per row of nnc_mi355x_registry.def.
 * Signature and naming are those of the reference's REGISTER_COMMAND_BACKEND macro
 * (lib/nnc/ccv_nnc_internal.h:196-202); the reference host calls them from _ccv_nnc_cmd_init()
 * (lib/nnc/cmd/ccv_nnc_cmd.inc:944-). */
#ifndef NNC_MI355X_REGISTRY_H
#define NNC_MI355X_REGISTRY_H
#ifdef __cplusplus
extern "C" {
#endif
#define NNC_ROW(cmd, backend) void _register_command_##cmd##_backend_##backend(ccv_nnc_cmd_backend_registry_t* const registry);
#include "nnc_mi355x_registry.def"
#undef NNC_ROW
#ifdef __cplusplus
}
#endif
#endif
