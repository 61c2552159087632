"""The drop-in boundary itself (CPU tier, no compute calls):
  * libnnc_mi355x.so (cross-compiled for gfx950) exports every symbol include/nnc_mi355x.h + the registry .def files
    declare, plus the legacy-spelled names the unmodified reference host links against;
  * the ABI mirrors of include/nnc_mi355x.h have the reference's sizes / offsets -- checked against literal numbers always,
    and against the REAL reference headers (a compiled sizeof/offsetof probe) whenever /root/reference is present;
  * the reference host's GPU registration table and ours + the generated stubs cover each other exactly."""
import os
import re
import subprocess
import ctypes as C
import pytest
from ccv_amd import nnc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "ccv_amd", "lib", "libnnc_mi355x.so")
REF = "/root/reference"


def _exported():
    out = subprocess.check_output(["nm", "-D", "--defined-only", LIB], text=True)
    return {l.split()[-1] for l in out.splitlines() if l.strip()}


def _declared():
    hdr = open(os.path.join(ROOT, "include", "nnc_mi355x.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b((?:nnc_mi355x|ccv_nnc)_\w+)\s*\(", hdr))
    names -= {n for n in names if n.endswith("_f") or n.endswith("_t")}
    rows = re.findall(r"NNC_ROW\((\w+), (\w+)\)", open(os.path.join(ROOT, "include", "nnc_mi355x_registry.def")).read())
    stubs = re.findall(r"NNC_STUB_ROW\((\w+), (\w+)\)", open(os.path.join(ROOT, "include", "nnc_mi355x_registry_stubs.def")).read())
    return names, rows, stubs


@pytest.mark.skipif(not os.path.exists(LIB), reason="library not built (run __graft_entry__.build())")
def test_library_exports_every_declared_symbol():
    exp = _exported()
    names, rows, stubs = _declared()
    missing = sorted(n for n in names if n not in exp)
    assert not missing, "declared in include/nnc_mi355x.h but not exported: %s" % missing
    for c, b in rows + stubs:
        assert "_register_command_%s_backend_%s" % (c, b) in exp, (c, b)
    # the names the reference host's objects leave undefined under its GPU configuration (lib/nnc/gpu/ccv_nnc_compat.h:23-59)
    for n in ("cumalloc", "cufree", "cudevice", "cumemcpy", "cuhostalloc", "cuhostfree", "curegister", "cuunregister", "curegmp", "cuunregmp",
              "cusetprofiler", "ccv_nnc_gpu_device_count", "co_stream_compat_await", "ccv_nnc_compat_depalettize",
              "ccv_nnc_init_stream_context", "ccv_nnc_deinit_stream_context", "ccv_nnc_synchronize_stream_context",
              "ccv_nnc_stream_compat_get_workspace", "ccv_nnc_stream_compat_drain", "ccv_nnc_stream_compat_add_callback",
              "ccv_nnc_init_stream_signal", "ccv_nnc_deinit_stream_signal", "ccv_nnc_stream_compat_emit_signal", "ccv_nnc_stream_compat_wait_signal"):
        assert n in exp, n


def test_registry_rows_are_disjoint_and_complete():
    _, rows, stubs = _declared()
    assert not (set(rows) & set(stubs))
    assert len(rows) + len(stubs) == 130  # the reference host's GPU registration table (ccv_nnc_cmd.inc:944-1075)
    if os.path.isdir(REF):
        inc = open(os.path.join(REF, "lib/nnc/cmd/ccv_nnc_cmd.inc")).read()
        host = set(re.findall(r"_register_command_(CCV_NNC_\w+?)_backend_(CCV_NNC_BACKEND_\w+)\(", re.search(r"#ifdef HAVE_CUDA\n(.*?)#endif", inc, re.S).group(1)))
        assert host == set(rows) | set(stubs)


def test_struct_sizes_match_the_reference_literals():
    # SURVEY.md 8(b) [probed]: sizes on x86-64
    assert C.sizeof(nnc.TensorParam) == 64
    assert C.sizeof(nnc.TensorStruct) == 112
    assert C.sizeof(nnc.TensorViewStruct) == 176
    assert C.sizeof(nnc.Cmd) == 152
    assert C.sizeof(nnc.Hint) == 144
    assert C.sizeof(nnc.BackendRegistry) == 40


PROBE = r'''
#include <stdio.h>
#include <stddef.h>
#include "%s"
#define S(t) printf(#t " %%zu\n", sizeof(t));
#define O(t, f) printf(#t "." #f " %%zu\n", offsetof(t, f));
int main(void) {
	S(ccv_nnc_tensor_param_t) S(ccv_nnc_tensor_t) S(ccv_nnc_tensor_view_t) S(ccv_nnc_cmd_param_t) S(ccv_nnc_hint_t) S(ccv_nnc_cmd_t) S(ccv_nnc_cmd_backend_registry_t)
	O(ccv_nnc_tensor_t, data) O(ccv_nnc_tensor_t, info) O(ccv_nnc_tensor_view_t, stride) O(ccv_nnc_tensor_view_t, contiguous)
	O(ccv_nnc_cmd_t, info) O(ccv_nnc_cmd_t, algorithm) O(ccv_nnc_cmd_param_t, bnorm.epsilon) O(ccv_nnc_cmd_param_t, bnorm.momentum) O(ccv_nnc_cmd_param_t, blas.a)
	O(ccv_nnc_cmd_param_t, sgd.rate) O(ccv_nnc_cmd_param_t, convolution.dilation) O(ccv_nnc_cmd_param_t, label_smoothing.trim1) O(ccv_nnc_hint_t, border.end)
	O(ccv_nnc_cmd_backend_registry_t, exec) O(ccv_nnc_cmd_backend_registry_t, autotune)
%s
	return 0;
}
'''


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference headers")
def test_struct_layout_matches_the_real_reference_headers(tmp_path):
    cc = "/opt/rocm/lib/llvm/bin/clang"
    outs = []
    for tag, header, inc, extra in (("ours", os.path.join(ROOT, "include", "nnc_mi355x.h"), [], 'S(struct ccv_nnc_stream_context_s) S(struct ccv_nnc_stream_signal_s)'),
                                    ("ref", "nnc/ccv_nnc.h", ["-I", os.path.join(REF, "lib")], '')):
        src = tmp_path / (tag + ".c")
        if tag == "ref":
            body = PROBE % (header, 'S(struct ccv_nnc_stream_context_s) S(struct ccv_nnc_stream_signal_s)')
            body = body.replace('#include "nnc/ccv_nnc.h"', '#include "nnc/ccv_nnc.h"\n#include "nnc/ccv_nnc_internal.h"\n#include "nnc/co.h"\n#include "nnc/_ccv_nnc_stream.h"')
        else:
            body = PROBE % (header, extra)
        src.write_text(body)
        exe = tmp_path / tag
        subprocess.check_call([cc, "-w", "-o", str(exe), str(src)] + inc + (["-DHAVE_SSE2", "-DHAVE_PTHREAD"] if tag == "ref" else []))
        outs.append(subprocess.check_output([str(exe)], text=True))
    assert outs[0] == outs[1], "\n--- ours ---\n%s--- reference ---\n%s" % tuple(outs)
