#!/bin/bash
# One rank of the process-per-GPU form on a one-GPU box (RCCL communicator of one): the harness's multi-stage step with and without the round-6 paths
# (multi-tensor SGD, overlapped gradient buckets).  usage: tools/force_comm_probe.sh <dtype 16|32> <model full|dawn|mini> <batch> <hw> <steps> [gpu|emu]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
dt=${1:-16}; model=${2:-full}; batch=${3:-256}; hw=${4:-224}; steps=${5:-3}; kind=${6:-gpu}
python - "$dt" "$model" "$batch" "$hw" "$steps" "$kind" <<'PY'
import ctypes as C, json, os, subprocess, sys
dt, model, batch, hw, steps, kind = sys.argv[1:7]
root = os.getcwd()
sys.path.insert(0, root)
from ccv_amd import nnc
def fresh_id():  # (an id makes ONE communicator)
    if kind == "gpu":
        b = C.create_string_buffer(128)
        assert nnc.load().dll.nnc_mi355x_comm_unique_id(b) == 0
        return b.raw.hex()
    import binascii
    return binascii.hexlify(os.urandom(16)).decode() + "5a" * 112
for sgd, ov in (("0", "0"), ("1", "0"), ("1", "1")):
    cid = fresh_id()
    env = dict(os.environ, HOST_BENCH_WORLD="1", HOST_BENCH_RANK="0", HOST_BENCH_DEVICE="0", HOST_BENCH_COMM_ID=cid, NNC_MI355X_SGD_BATCH=sgd, NNC_MI355X_COMM_OVERLAP=ov, OMP_NUM_THREADS="4")
    r = subprocess.run([os.path.join(root, "oracle", "_ref", "host_resnet_bench." + kind), batch, hw, steps, "1", dt, model], capture_output=True, text=True, timeout=900, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"replica')]
    print("== SGD_BATCH=%s COMM_OVERLAP=%s  exit %d" % (sgd, ov, r.returncode))
    if not lines:
        print("   no result line; stderr tail:", r.stderr[-400:])
        continue
    l = lines[-1].replace("nan", "NaN").replace("inf", "Infinity")
    try:
        d = json.loads(l)
        print("  ", {k: d.get(k) for k in ("replica_probe_sumsq", "comm_overlap", "ms_per_step", "images_per_s", "outputs_finite", "softmax_worst_row_sum_err", "device_out_sumsq")})
    except Exception as e:
        print("   unparsable:", e, l[:200])
PY
