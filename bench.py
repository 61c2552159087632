#!/usr/bin/env python
"""bench.py -- BASELINE.json metric: images/sec, VGG-D forward+backward(+SGD), fp32, batch 256 per GPU, on N MI355X.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One process per GPU.  A "step" is one pass of the hot path over one batch of synthetic input that is already resident
in HBM: every forward command, every backward command, (N > 1: RCCL all-reduce of every parameter gradient over xGMI),
one SGD command per parameter tensor.  Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events around
every launch of the contraction kernel (in-library, on the launching stream); `cpu_baseline` times the reference's own
lib/nnc CPU backend (oracle/_ref) on one image of the same network on this host.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
from ccv_amd import nnc  # noqa: E402
from ccv_amd.vgg import VGGD, vgg_d_flops_per_image, hash_unit  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 2.4 GHz


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(image, label, fwd_only=False):
    """The reference's own CPU path on this host: VGG-D at batch 1 on `image` (the GPU run's image 0,
    same seed-0 weights), all host cores (OpenMP), one warm-up + three repetitions each of
      * the whole step (forward + backward + SGD) through CPU_REF -- the reported `value` (median);
      * forward only through CPU_REF, and forward only with the convolutions on CPU_OPT (the reference's Winograd / direct
        fast paths, cpu_opt/_ccv_nnc_conv_cpu_opt.c) -- reported beside it.
    The warm-up step runs on the initial weights: its loss is what bench.py checks the GPU's step-1 loss of image 0 against."""
    from oracle_bind import oracle_lib
    O, backend, per_image = oracle_lib()
    kind = "reference" if O.kind == "reference" else "port"
    from oracle_vgg import make_vggd
    net = make_vggd(O, 1, memory=nnc.CPU_MEMORY, backend=backend, pool_per_image=per_image, init="hash")
    net.set_input(image[None], [label])
    REPS = 3

    def timed(fn):
        t0 = time.time()
        fn()
        return time.time() - t0
    (net.forward if fwd_only else net.step)()  # warm-up (pages, OpenMP pool) on the initial weights
    loss0 = float(net.loss.numpy()[0])
    fwd_ref = sorted(timed(net.forward) for _ in range(REPS))
    steps = fwd_ref if fwd_only else sorted(timed(net.step) for _ in range(REPS))  # (config 2 is forward only: no training step is timed for it)
    out = {"value": 1.0 / steps[REPS // 2], "unit": "images/s", "cores": os.cpu_count() if kind == "reference" else 1, "kind": kind,
           "cpu": cpu_model(), "forward_only_cpu_ref_images_per_s": 1.0 / fwd_ref[REPS // 2],
           "sample": "1 image, VGG-D %s through lib/nnc CPU_REF (oracle/_ref, clang -O3 -fopenmp, no BLAS: the fc layers run "
                     "the reference's own loops), warm-up + %d repetitions, median %.2f s (min %.2f, max %.2f)" % ("forward" if fwd_only else "fwd+bwd+SGD", REPS, steps[REPS // 2], steps[0], steps[-1])}
    if kind == "reference":
        net.conv_fwd_backend = nnc.BACKEND_CPU_OPT
        try:
            net.forward()
            fwd_opt = sorted(timed(net.forward) for _ in range(REPS))
            out["forward_only_cpu_opt_images_per_s"] = 1.0 / fwd_opt[REPS // 2]
        except RuntimeError as e:
            out["forward_only_cpu_opt_images_per_s"] = None
            out["cpu_opt_note"] = str(e)
    return out, loss0


def via_host(args, losses, params, fwd_only=False):
    """The same training step through the REFERENCE HOST (oracle/_ref/host_vgg_bench.gpu = tools/host_vgg_bench.c against the
    unmodified reference host + this backend): symbolic graph, ccv_nnc_symbolic_graph_minimize, compile (tensor arena),
    ccv_nnc_graph_autotune, static schedule -- timed there, and its first step compared with this process's.  fwd_only (config 2):
    the forward graph alone -- the node sequence of test/int/nnc/graph.vgg.d.tests.c:14-90 on synthetic images."""
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "host_vgg_bench.gpu")
    if not os.path.exists(exe):
        return {"status": "oracle/_ref/host_vgg_bench.gpu not built"}
    try:
        r = subprocess.run([exe, str(args.batch), "225", str(min(args.steps, 6)), "2", "full"] + (["fwd"] if fwd_only else []), capture_output=True, text=True, timeout=900, env=dict(os.environ, NNC_MI355X_PEEPHOLE_STATS="1"))
        if r.returncode != 0:
            return {"status": "failed (%d): %s" % (r.returncode, (r.stdout + r.stderr)[-300:])}
        h = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        return {"status": "failed: %s" % e}
    hl = np.array(h["loss"], dtype=np.float64)
    loss_err = float(np.max(np.abs(hl - losses[:len(hl)]) / np.maximum(np.abs(losses[:len(hl)]), 1e-30)))
    perr = qerr = 0.0
    if not fwd_only:
        perr = max(abs(a - b[0]) / max(abs(b[0]), 1e-3 * b[1] ** 0.5, 1e-30) for a, b in zip(h["updated_param_sum"], params))
        qerr = max(abs(a - b[1]) / max(b[1], 1e-30) for a, b in zip(h["updated_param_sumsq"], params))
    import re
    m = re.search(r"look-ahead: (\d+) commands recorded, (\d+) completed by their ReLU, (\d+) launched as they were", r.stderr)
    return {"images_per_s": h["images_per_s"], "ms_per_step": h["ms_per_step"], "autotune_ms": h["autotune_ms"],
            "relu_look_ahead": {"recorded": int(m.group(1)), "folded": int(m.group(2)), "launched_plain": int(m.group(3))} if m else None,
            "step1_max_rel_err_vs_command_driver": {"loss": loss_err} if fwd_only else {"loss": loss_err, "updated_param_sum": perr, "updated_param_sumsq": qerr},
            "equal": bool(loss_err <= 1e-4 and qerr <= 1e-4), "driver": h["driver"]}


def resnet_config(args, half, dawn=False):
    """BASELINE config 4 on one MI355X: ResNet-50 v1d (bin/nnc/imagenet.c:17-98), NCHW, batch 256, forward + backward + Nesterov SGD,
    through the REFERENCE HOST's own model API (tools/host_resnet_bench.c -> oracle/_ref/host_resnet_bench.gpu: ccv_cnnp_model_fit on
    this backend; cnnp, autodiff, compile and the scheduler are the reference's unmodified code).  The timing is the harness's
    (wall clock around K fit calls bracketed by stream waits); `roofline` is the batch-norm command (HBM-bound, SURVEY 8(d):
    2 |x| forward, 3 |x| backward algorithmic bytes) from the backend's HIP-event records of one more step."""
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "host_resnet_bench.gpu")
    if not os.path.exists(exe):
        raise SystemExit("oracle/_ref/host_resnet_bench.gpu not built (oracle/build_ref_host.sh)")
    gpus = max(1, args.gpus)
    cmd = [exe, str(args.batch), "32" if dawn else "224", str(args.steps), str(max(args.warmup, 1)), "16" if half else "32"]
    env = dict(os.environ, NNC_MI355X_PEEPHOLE_STATS="1")
    rank_lines = None
    if (gpus > 1 and args.dp == "process") or os.environ.get("NNC_BENCH_FORCE_COMM") == "1":  # (FORCE_COMM: the same code on a communicator of one, for a 1-GPU box)
        # ONE PROCESS PER GPU (the default for N > 1 since round 4): N harness processes, rank r on device r, RCCL bootstrapped from an id made here; the
        # ranks meet in the reference's multi-stage training API with one in-place all-reduce per parameter gradient (tools/host_resnet_bench.c).  One
        # host thread feeds ~10 devices on config 4 fp32 but only ~6 on config 4-f16 and ~4 on config 5: this form does not care.
        import ctypes as C
        buf = C.create_string_buffer(128)
        if nnc.load().dll.nnc_mi355x_comm_unique_id(buf) != 0:
            raise SystemExit("ncclGetUniqueId failed")
        devices = 1
        env.setdefault("NNC_MI355X_COMM_OVERLAP", "1")  # the gradient all-reduces in buckets on a communication stream, each behind its own gradients' writers (cmd_comm.cpp "Overlap")
        procs = [subprocess.Popen(cmd + (["dawn"] if dawn else ["full"]), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                  env=dict(env, HOST_BENCH_WORLD=str(gpus), HOST_BENCH_RANK=str(k), HOST_BENCH_DEVICE=str(k), HOST_BENCH_COMM_ID=buf.raw.hex())) for k in range(gpus)]
        outs = [p.communicate(timeout=3000) for p in procs]
        for k, (p, (so, se)) in enumerate(zip(procs, outs)):
            if p.returncode != 0:
                raise SystemExit("host_resnet_bench rank %d failed (%d): %s" % (k, p.returncode, (so + se)[-600:]))
        rank_lines = [json.loads(so.strip().splitlines()[-1]) for so, _ in outs]
        r = type("R", (), {"stdout": outs[0][0], "stderr": outs[0][1], "returncode": 0})()
    else:
        devices = gpus  # the reference's single-process data parallelism (ccv_cnnp_model_set_data_parallel): `batch` per device, one host thread for all of them
        if devices == 1 and not args.no_capture:
            # SURVEY.md section 8(f)3, HIP-graph capture of the compiled schedule: BOTH forms are timed in the one process -- K steps issued command by command by the
            # reference host's scheduler, then the same step recorded once (nnc_mi355x_capture_begin / _end around the host's step call) and replayed K times
            env["HOST_BENCH_CAPTURE"] = "2"
        r = subprocess.run(cmd + (["dawn", str(devices)] if dawn else ["full", str(devices)]), capture_output=True, text=True, timeout=3000, env=env)
        if r.returncode != 0:
            raise SystemExit("host_resnet_bench failed (%d): %s" % (r.returncode, (r.stdout + r.stderr)[-600:]))
    h = json.loads(r.stdout.strip().splitlines()[-1])
    dp_check = None
    if rank_lines:  # the slowest rank's clock (every rank brackets its timed steps with a cross-rank barrier + stream wait)
        h = dict(rank_lines[0], ms_per_step=max(l["ms_per_step"] for l in rank_lines))
        h["images_per_s"] = gpus * args.batch / (h["ms_per_step"] * 1e-3)
        devices = gpus
        # the data-parallel check of this form: after the timed steps every rank holds the same parameters (the harness reads the first and the last parameter
        # tensor back: sum of squares), and every rank's communicator has all the ranks in it
        probes = [l.get("replica_probe_sumsq") for l in rank_lines]
        if any(p is None or any(v is None for v in p) for p in probes):  # (the harness prints null for a non-finite probe: ResNet-50 in f16 diverges in its hand-made multi-stage step, tools/host_resnet_bench.c)
            dp_check = {"replica_param_sumsq_max_rel_diff": None, "rccl_ranks_per_rank": [l["process_per_gpu"]["rccl_ranks"] for l in rank_lines], "ok": False, "note": "non-finite parameters after the timed steps"}
            print("bench.py: a rank's parameters are not finite after the timed steps", file=sys.stderr)
        elif all(p for p in probes):
            rep = max(abs(p[k] - probes[0][k]) / max(abs(probes[0][k]), 1e-30) for p in probes for k in range(2))
            ranks_ok = all(l["process_per_gpu"]["rccl_ranks"] == gpus for l in rank_lines)
            dp_check = {"replica_param_sumsq_max_rel_diff": rep, "rccl_ranks_per_rank": [l["process_per_gpu"]["rccl_ranks"] for l in rank_lines], "ok": bool(rep <= 1e-6 and ranks_ok)}
            if not dp_check["ok"]:
                print("bench.py: the ranks' replicas differ after the timed steps: %r" % (dp_check,), file=sys.stderr)
    gflop = 25.97  # ResNet-50 v1d forward + backward per image (1 MAC = 2 FLOP)
    if dawn:  # DawnNet: 3x3 convolutions 3->64 @32^2, 64->128 @32^2, 2 x 128->128 @16^2, 128->256 @16^2, 256->512 @8^2, 2 x 512->512 @4^2, dense 512->10; x3 for fwd + bwd
        macs = 9 * (3 * 64 * 1024 + 64 * 128 * 1024 + 2 * 128 * 128 * 256 + 128 * 256 * 256 + 256 * 512 * 64 + 2 * 512 * 512 * 16) + 5120
        gflop = 3 * 2 * macs / 1e9
    out = {"metric": ("images/sec fwd+bwd CIFAR-10 DawnNet 32x32 bs%d NCHW" if dawn else "images/sec fwd+bwd ResNet-50 v1d 224x224 bs%d NCHW") % args.batch, "value": h["images_per_s"], "unit": "images/s", "n_gpus": devices,
           "steps": args.steps, "warmup": max(args.warmup, 1), "ms_per_step": h["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f16" if half else "f32", "data": "synthetic",
           "config": {"workload": ("CIFAR-10 DawnNet (bin/nnc/cifar-10.c:76-127) NCHW, the trainer's own step (evaluate, softmax cross-entropy, backward, apply gradients; Nesterov SGD), batch %d, random-init weights, driven by the reference host's model API" if dawn else
                                   "ResNet-50 v1d (bin/nnc/imagenet.c) NCHW forward+backward+Nesterov SGD, batch %d, random-init weights, driven by the reference host's ccv_cnnp_model_fit") % args.batch,
                      "global_batch": args.batch * devices, "parallelism": "dp%d%s" % (devices, (" (one process per GPU; the reference's evaluate / backward / parameter_gradients_map(COMM_ALLREDUCE) / apply_gradients; RCCL ranks %s)" % [l["process_per_gpu"]["rccl_ranks"] for l in rank_lines]) if rank_lines else (" (one process, ccv_cnnp_model_set_data_parallel; gradients all-reduced by the COMM_ALLREDUCE rows over RCCL)" if devices > 1 else "")), "gflop_per_image": gflop, "whole_step_tflops_per_gpu": h["images_per_s"] / devices * gflop / 1e3,
                      "first_step_ms": h["first_step_ms"], "outputs_finite": h["outputs_finite"], "softmax_worst_row_sum_err": h["softmax_worst_row_sum_err"], "memory_gib": h["memory_gib"]}}
    cap = h.get("capture") or {}
    if cap.get("on"):
        # `value` / `ms_per_step` above are the REPLAYED step (every kernel of the step runs at every replay; parameters bit-identical to the issued form:
        # tests/test_via_host.py); the step issued command by command is reported beside it
        issued = cap.get("issued_per_command_ms_per_step") or 0.0
        out["config"]["step_form"] = "HIP-graph replay of the captured step (nnc_mi355x_capture_begin / _end around the reference host's step call; one hipGraphLaunch per step)"
        out["config"]["capture"] = {"graph_nodes": cap["graph_nodes"], "capture_ms": cap["capture_ms"], "launch_host_ms_median": cap.get("launch_host_ms_median"),
                                    "issued_per_command_ms_per_step": issued, "issued_per_command_images_per_s": (args.batch * devices / (issued * 1e-3)) if issued > 0 else None,
                                    "devices_one_thread_can_feed": (h["ms_per_step"] / cap["launch_host_ms_median"]) if cap.get("launch_host_ms_median") else None}
    if rank_lines:
        out["config"]["rccl_ranks"] = rank_lines[0]["process_per_gpu"]["rccl_ranks"]
        out["config"]["comm_overlap"] = rank_lines[0].get("comm_overlap")  # all-reduces that went out in buckets beside the backward pass (0 / 0: NNC_MI355X_COMM_OVERLAP=0)
    if dp_check is not None:
        out["config"]["data_parallel_check"] = dp_check
    ks = h.get("kernels", [])
    bn = [k for k in ks if k["bytes"] > 0 and k["ms"] > 0]
    import re
    m = re.search(r"look-ahead: (\d+) commands recorded, (\d+) completed by their ReLU, (\d+) launched as they were", r.stderr)
    if m:  # the library's ReLU look-ahead (peephole.cpp): the host's graph is unchanged, batch norm + ReLU and pool-gradient + ReLU-backward pairs fold
        out["config"]["relu_look_ahead"] = {"recorded": int(m.group(1)), "folded": int(m.group(2)), "launched_plain": int(m.group(3))}
    if bn:
        byts, ms, n = sum(k["bytes"] for k in bn), sum(k["ms"] for k in bn), sum(k["launches"] for k in bn)
        ach = byts / (ms * 1e-3) / 1e9
        # HBM bytes per batch-norm command from the committed PMC pass of this configuration (tools/pmc_pass.sh with PMC_BENCH_ARGS="--config ..." ->
        # profiles/pmc_traffic_<config>.json; counters cannot be collected inside the timed run): every batch-norm kernel (cluster kernels, plane
        # kernels), read + written, over the forward + backward commands of the recorded step
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic_%s.json" % args.config)
        if os.path.exists(tpath):
            pk = json.load(open(tpath))["kernels"]
            bnk = [v for k, v in pk.items() if "bn_" in k]  # (half-precision kernels appear under their mangled names)
            fw_cmds_per_step = sum(k["launches"] for k in bn if "apply" in k["name"])  # the forward commands of the recorded step
            fw_in_pass = sum(v["launches"] for k, v in pk.items() if "bn_cluster_forw" in k or "bn_apply_planes" in k)  # one of these per forward command
            if bnk and fw_cmds_per_step and fw_in_pass:
                traffic = sum((v["read_bytes_per_launch"] + v["write_bytes_per_launch"]) * v["launches"] for v in bnk) / (fw_in_pass / fw_cmds_per_step) / n
        out["roofline"] = {"bound": "hbm", "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0, "traffic": traffic,
                           "kernel": "batch norm forward + backward commands (%s)" % "; ".join(sorted(set(k["name"] for k in bn))), "launches": n, "avg_ms": ms / n,
                           "ms_per_step": ms, "recorded_kernels": {k["name"][-100:]: {"ms": k["ms"], "launches": k["launches"], "tflops": (k["flops"] / (k["ms"] * 1e-3) / 1e12 if k["ms"] > 0 else 0.0)} for k in ks}}
    # half precision: the f16 contractions of the recorded step, each LAUNCH against the bound of its own shape: algorithmic
    # FLOP per algorithmic byte under the machine balance 2.5 PFLOP/s / 8 TB/s = 312 -> HBM-bound (the 1 x 1 convolutions with 64 .. 512 channels), else bound by
    # the dense f16 MFMA peak (MI355X_MICROARCH.md: ~2.5 PFLOP/s).  Both fractions are reported; `bound` names the class with more of the step's time.
    f16k = [k for k in ks if "mfma_gemm_f16" in k["name"] and k["ms"] > 0 and k["flops"] > 0]
    if half and f16k:
        top = max(f16k, key=lambda k: k["ms"])
        allf = sum(k["flops"] for k in f16k) / (sum(k["ms"] for k in f16k) * 1e-3) / 1e12
        by = h.get("f16_contractions_by_bound") or {}
        mf, hb = by.get("mfma") or {}, by.get("hbm") or {}
        cls = {}
        if mf.get("ms", 0) > 0:
            a = mf["flops"] / (mf["ms"] * 1e-3) / 1e12
            cls["mfma_bound"] = {"achieved": a, "peak": 2500.0, "unit": "TFLOP/s", "frac": a / 2500.0, "ms": mf["ms"], "launches": mf["launches"]}
        if hb.get("ms", 0) > 0:
            a = hb["bytes"] / (hb["ms"] * 1e-3) / 1e9
            cls["hbm_bound"] = {"achieved": a, "peak": 8000.0, "unit": "GB/s", "frac": a / 8000.0, "ms": hb["ms"], "launches": hb["launches"], "tflops": hb["flops"] / (hb["ms"] * 1e-3) / 1e12}
        lead = "hbm_bound" if hb.get("ms", 0) > mf.get("ms", 0) else "mfma_bound"
        if not cls:  # a harness built before the per-launch classes existed: the kernel with the most time against the matrix peak, labelled as such
            ach = top["flops"] / (top["ms"] * 1e-3) / 1e12
            out["roofline_f16_contractions"] = {"bound": "mfma (unclassified: rebuild oracle/_ref/host_resnet_bench.gpu for the per-shape classes)", "achieved": ach, "peak": 2500.0, "unit": "TFLOP/s", "frac": ach / 2500.0, "traffic": None,
                                                "kernel": top["name"][-100:], "launches": top["launches"], "avg_ms": top["ms"] / top["launches"], "all_f16_contractions": {"achieved_tflops": allf, "ms": sum(k["ms"] for k in f16k)}}
        if lead in cls:
            out["roofline_f16_contractions"] = dict(cls[lead], bound="hbm" if lead == "hbm_bound" else "mfma", traffic=None, kernel="every f16 contraction launch whose shape is %s (FLOP / byte %s 312)" % ("HBM-bound" if lead == "hbm_bound" else "MFMA-bound", "<" if lead == "hbm_bound" else ">="),
                                                    by_bound=cls, top_kernel={"name": top["name"][-100:], "ms": top["ms"], "launches": top["launches"], "tflops": top["flops"] / (top["ms"] * 1e-3) / 1e12},
                                                    all_f16_contractions={"achieved_tflops": allf, "ms": sum(k["ms"] for k in f16k)})
    # one host thread enqueues for every device of the reference's single-process data parallelism (lib/nnc/ccv_nnc_graph_run.c:581-675): N x the
    # enqueue time of a step must stay under the GPU time of a step for the N-device form to scale
    he = h.get("host_enqueue")
    if he and he.get("commands_per_step"):
        issued_ms = (cap.get("issued_per_command_ms_per_step") or 0.0) if cap.get("on") else 0.0
        out["config"]["host_enqueue"] = dict(he, devices_one_thread_can_feed=(issued_ms or h["ms_per_step"]) / he["ms_per_step_median"] if he["ms_per_step_median"] > 0 else None,
                                             note="wall time of the step call on drained streams, one host thread, one device; the step's GPU time / this = how many devices that thread keeps busy")
    if not args.no_cpu_baseline and devices == 1:
        out["cpu_baseline"], out["config"]["oracle_gate"] = host_cpu_baseline(args, half, dawn)
    emit(out)


def lstm_config(args):
    """The recurrent half of the reference's NLP trainer (test/int/nnc/imdb.tests.c:972-980, :1278): ccv_cnnp_lstm -- 2 layers, 128 hidden units, masked,
    batch-first -- at the IMDB classifier's shape (batch 64, sequences of 512 steps, 128 embedding features), evaluate + backward through the reference
    host's model API on this backend's LSTM rows (tools/host_lstm_check.c with HOST_LSTM_BENCH: wall clock between two blocking read-backs).  The reference
    has NO CPU implementation of the row (lib/nnc/cmd/rnn registers the GPU backend only), so there is no cpu_baseline; the first pass of the same process is
    replayed against the numpy restatement by tests/test_via_host.py (anchored to torch.nn.LSTM in float64)."""
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "oracle", "_ref", "host_lstm_check.gpu")
    if not os.path.exists(exe):
        raise SystemExit("oracle/_ref/host_lstm_check.gpu not built (oracle/build_ref_host.sh)")
    nnc.load()  # fail loudly without the HIP library / a GPU
    B = args.batch if args.batch != 256 else 64
    T, I, H, layers = 512, 128, 128, 2
    with tempfile.TemporaryDirectory() as d:
        r = subprocess.run([exe, str(T), str(B), str(I), str(H), str(layers), "0", "1", "1", os.path.join(d, "lstm.bin")], capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, HOST_LSTM_BENCH=str(max(args.steps, 1))))
    if r.returncode != 0:
        raise SystemExit("host_lstm_check failed (%d): %s" % (r.returncode, (r.stdout + r.stderr)[-600:]))
    h = [json.loads(l) for l in r.stdout.strip().splitlines() if l.startswith("{")][-1]
    out = {"metric": "sequences/sec fwd+bwd ccv_cnnp_lstm (IMDB classifier shape) bs%d" % B, "value": h["sequences_per_s"], "unit": "sequences/s", "n_gpus": 1, "steps": h["steps"], "warmup": 2,
           "ms_per_step": h["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "ccv_cnnp_lstm (2 layers, 128 hidden, masked, batch-first) evaluate + backward, batch %d x 512 steps x 128 features (test/int/nnc/imdb.tests.c:972-980), through the reference host's model API" % B,
                      "global_batch": B, "parallelism": "dp1", "timesteps_per_s": h["timesteps_per_s"], "gflop_per_step": h["gflop_per_step"]},
           # the row is bound by the hand-over between the workgroups of the one-launch kernel once per time step, not by a pipe: the
           # line reports its algorithmic rate against the fp32 matrix peak for scale
           "roofline": {"bound": "latency (the per-step chain of dependent LDS round trips; the fp32 matrix peak is quoted for scale only)", "achieved": h["tflops"], "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": h["tflops"] / FP32_MFMA_PEAK_TFLOPS, "traffic": None,
                        "kernel": "lstm_rows_forw_kernel / lstm_rows_back_kernel (hidden size <= 128: a workgroup per batch row, R in registers) + the batched input contractions", "launches": None, "avg_ms": None},
           "cpu_baseline": None}
    out["config"]["cpu_baseline_note"] = "the reference has no CPU LSTM (lib/nnc/cmd/rnn: GPU backend only): nothing to time beside it"
    emit(out)


def host_cpu_baseline(args, half, dawn):
    """configs 4 / 5 next to the reference's own CPU path on this box's host cores: the SAME harness built against the reference's
    CPU backend (oracle/_ref/host_resnet_bench.cpu: the reference's unmodified host, its CPU_REF rows, every tensor in CPU memory, fp32) -- timed on a
    bounded sample (2 images of ResNet-50 / 16 of the DawnNet per step, one warm-up + two timed steps), and, with HOST_BENCH_CHECK=1, both builds run ONE
    step from identical parameters: the per-image losses of this backend against the CPU's are the line's oracle gate (tests/test_via_host.py holds the
    parameters and steps too)."""
    import subprocess
    cpu = os.path.join(ROOT, "oracle", "_ref", "host_resnet_bench.cpu")
    gpu = os.path.join(ROOT, "oracle", "_ref", "host_resnet_bench.gpu")
    if not os.path.exists(cpu):
        return None, None
    n = 16 if dawn else 2
    tail = ["dawn"] if dawn else ["full"]
    hw = "32" if dawn else "224"
    base = gate = None
    # the CPU sample does not depend on the GPU side's precision: within ONE default run (extra_configs) the fp32 and f16 lines of a network share it
    cache = os.environ.get("NNC_BENCH_CPU_CACHE")
    cpath = os.path.join(cache, "cpu_%s_%d.json" % ("dawn" if dawn else "resnet50", n)) if cache else None
    want = None
    if cpath and os.path.exists(cpath):
        c = json.load(open(cpath))
        base, want = c["base"], c["want"]
    try:
        env = dict(os.environ, HOST_BENCH_CHECK="1")
        if base is None:
            r = subprocess.run([cpu, str(n), hw, "2", "1", "32"] + tail, capture_output=True, text=True, timeout=900)
            if r.returncode == 0:
                c = json.loads(r.stdout.strip().splitlines()[-1])
                base = {"value": c["images_per_s"], "unit": "images/s", "cores": os.cpu_count(), "kind": "reference",
                        "sample": "%s forward + backward + SGD on %d images per step (fp32, the reference host on its CPU_REF rows, OpenMP over the host's hardware threads, no BLAS in the image), one warm-up + two timed steps: %.1f ms per step" % ("DawnNet" if dawn else "ResNet-50 v1d 224 x 224", n, c["ms_per_step"])}
            rc = subprocess.run([cpu, str(n), hw, "0", "1", "32"] + tail, capture_output=True, text=True, timeout=900, env=env)
            if rc.returncode == 0:
                want = json.loads(rc.stdout.strip().splitlines()[-1])["check"]
            if cpath and base is not None and want is not None:
                json.dump({"base": base, "want": want}, open(cpath, "w"))
        rg = subprocess.run([gpu, str(n), hw, "0", "1", "16" if half else "32"] + tail, capture_output=True, text=True, timeout=900, env=env)
        if want is not None and rg.returncode == 0:
            got = json.loads(rg.stdout.strip().splitlines()[-1])["check"]
            rel = max(abs(a - b) / max(abs(b), 1e-30) for a, b in zip(got["loss"], want["loss"]))
            tol = 5e-3 if half else 1e-4
            gate = {"step1_loss_per_image_max_rel_err": rel, "tolerance": tol, "ok": bool(rel <= tol), "images": n, "loss_image0": {"gpu": got["loss"][0], "cpu_ref": want["loss"][0]},
                    "out_sumsq_rel_err": abs(got["out_sumsq"] - want["out_sumsq"]) / want["out_sumsq"]}
            if not gate["ok"]:
                raise SystemExit("bench.py: step-1 losses differ from the reference host's CPU step by %.3g (> %.1g)" % (rel, tol))
    except subprocess.TimeoutExpired:
        pass
    return base, gate


def pmc_traffic(symbol, batch):
    """HBM bytes per launch of the dominant kernel, from the committed PMC passes (profiles/pmc_traffic_bs<batch>.json, written
    by tools/pmc_pass.sh + tools/pmc_traffic.py from `rocprofv3 --pmc` runs of this same command: counters cannot be
    collected inside the timed run).  `symbol` is the part of a record's name behind '|': the kernel symbol itself, or -- for
    the contraction template -- the tail of its signature (LA, LB, WM, WN) plus the epilogue.  None when no pass exists."""
    import re
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic_bs%d.json" % batch)
    if not os.path.exists(path):
        return None
    kernels = json.load(open(path))["kernels"]
    m = re.search(r"LA = (nnc::\w+<[^>]*>), LB = nnc::(\w+)<.*WM = (\d), WN = (\d)\] EPI = (\w+)", symbol)
    tot, n = 0.0, 0
    for k, v in kernels.items():
        if m:
            la, lb, wm, wn, epi = m.groups()
            hit = k.startswith("nnc::mfma_gemm_f32_kernel<" + la.replace(">", "")) and ("nnc::" + lb + "<") in k and ("nnc::" + epi + ",") in k and k.endswith("%s, %s, 0>" % (wm, wn))
        else:
            hit = k.startswith(symbol.rstrip(">"))
        if hit:
            tot += (v["read_bytes_per_launch"] + v["write_bytes_per_launch"]) * v["launches"]
            n += v["launches"]
    return tot / n if n else None


EXTRA_CONFIGS = ["vggd-fwd-bs64", "resnet50-nchw-bs256", "resnet50-nchw-bs256-f16", "cifar10-dawn-f16-bs512", "imdb-lstm-bs64"]


def extra_configs(args):
    """BASELINE configs 2, 4, 4 at the trainer's own precision, and 5 under the SAME clock as the headline: the default run (N = 1) ends with one short run of
    each -- a fresh `bench.py --config <name> --steps 5 --warmup 2` process after the headline's timed region and legs are over (its autotune and warm-up
    are untimed there as here) -- and carries their lines, reduced to value / ms_per_step / roofline / cpu_baseline / oracle gate, in `extra_configs`.  A
    config that fails or times out is reported as such; it never takes the headline down."""
    import subprocess
    import tempfile
    res = {}
    cache = tempfile.mkdtemp(prefix="nnc_bench_cpu_")
    for name in EXTRA_CONFIGS:
        t0 = time.time()
        cmd = [sys.executable, os.path.abspath(__file__), "--config", name, "--steps", "5", "--warmup", "2", "--no-extra-configs", "--no-alt-leg"]
        if args.no_cpu_baseline:
            cmd.append("--no-cpu-baseline")
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, NNC_BENCH_CPU_CACHE=cache))
            lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
            if r.returncode != 0 or not lines:
                res[name] = {"status": "failed (%d): %s" % (r.returncode, (r.stderr or r.stdout)[-300:]), "wall_s": time.time() - t0}
                continue
            d = json.loads(lines[-1])
        except Exception as e:  # noqa: BLE001
            res[name] = {"status": "failed: %s" % e, "wall_s": time.time() - t0}
            continue
        cfg = d.get("config", {})
        e = {k: d.get(k) for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "via_host_images_per_s") if d.get(k) is not None}
        e["workload"] = cfg.get("workload")
        for rk in ("roofline", "roofline_f16_contractions"):
            if rk in d:
                e[rk] = {k: d[rk].get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "launches", "avg_ms", "by_bound", "ms") if k in d[rk]}
                if isinstance(e[rk].get("kernel"), str):
                    e[rk]["kernel"] = e[rk]["kernel"][:120]
        if "cpu_baseline" in d and d["cpu_baseline"]:
            e["cpu_baseline"] = {k: d["cpu_baseline"].get(k) for k in ("value", "unit", "cores", "kind", "sample")}
        gate = cfg.get("oracle_gate") or cfg.get("step1_loss_image0")
        if gate:
            e["oracle_gate"] = gate
        for k in ("relu_look_ahead", "host_enqueue", "capture", "whole_step_tflops_per_gpu", "memory_gib"):
            if k in cfg:
                e[k] = cfg[k]
        e["wall_s"] = time.time() - t0
        res[name] = e
    return res


_RESULT = None


def claim_stdout():
    """The contract is ONE JSON line on stdout.  RCCL prints a version banner to C stdout when a communicator is created (and child
    processes or libraries may print too): file descriptor 1 is pointed at stderr for the life of the process and the JSON line goes
    to a private duplicate of the real stdout."""
    global _RESULT
    if _RESULT is None:
        sys.stdout.flush()
        _RESULT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


LINE_LIMIT = 7400  # the driver keeps an 8 KB tail of stdout: the ONE JSON line must fit it whole (the head of an 11.6 KB line was cut)


def compact(out):
    """The line as printed: every contract key, `roofline`, `cpu_baseline` and `config` in full; the extra configurations reduced to their numbers (value,
    ms_per_step, roofline fractions, CPU baseline value, oracle gate verdict).  The full record goes to stderr (`bench.py full record: {...}`)."""
    def slim_roof(r):
        keep = {k: r.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "launches", "avg_ms") if k in r}
        if isinstance(r.get("kernel"), str):
            keep["kernel"] = r["kernel"][:60]
        if "by_bound" in r:
            keep["by_bound"] = {k: {kk: v.get(kk) for kk in ("achieved", "unit", "frac", "ms")} for k, v in r["by_bound"].items()}
        return keep
    o = json.loads(json.dumps(out))
    if len(json.dumps(o)) <= LINE_LIMIT:
        return o
    for name, e in (o.get("extra_configs") or {}).items():
        if not isinstance(e, dict) or "value" not in e:
            continue
        n = {k: e[k] for k in ("value", "unit", "ms_per_step", "dtype", "via_host_images_per_s", "whole_step_tflops_per_gpu", "wall_s") if k in e}
        for rk in ("roofline", "roofline_f16_contractions"):
            if rk in e:
                n[rk] = slim_roof(e[rk])
        if e.get("cpu_baseline"):
            n["cpu_baseline"] = {k: e["cpu_baseline"].get(k) for k in ("value", "unit", "cores", "kind")}
        g = e.get("oracle_gate")
        if isinstance(g, dict):
            n["oracle_gate"] = {k: g[k] for k in ("ok", "step1_loss_per_image_max_rel_err", "rel_err", "tolerance", "bound") if k in g}
        he = e.get("host_enqueue")
        if isinstance(he, dict):
            n["host_enqueue"] = {k: he.get(k) for k in ("ms_per_step_median", "commands_per_step", "devices_one_thread_can_feed")}
        cp = e.get("capture")
        if isinstance(cp, dict):  # value / ms_per_step are the replayed step; the step issued command by command beside it
            n["capture"] = {k: cp.get(k) for k in ("graph_nodes", "launch_host_ms_median", "issued_per_command_images_per_s", "devices_one_thread_can_feed")}
        o["extra_configs"][name] = n
    if len(json.dumps(o)) > LINE_LIMIT and isinstance(o.get("roofline"), dict) and "all_contractions" in o["roofline"]:
        ac = o["roofline"]["all_contractions"]
        o["roofline"]["all_contractions"] = {k: ac[k] for k in ("achieved", "ms") if k in ac}
    if len(json.dumps(o)) > LINE_LIMIT and isinstance(o.get("roofline"), dict) and "recorded_kernels" in o["roofline"]:
        del o["roofline"]["recorded_kernels"]
    if len(json.dumps(o)) > LINE_LIMIT and isinstance(o.get("cpu_baseline"), dict) and isinstance(o["cpu_baseline"].get("sample"), str):
        o["cpu_baseline"]["sample"] = o["cpu_baseline"]["sample"][:200]
    return o


def emit(out):
    claim_stdout()
    line = json.dumps(compact(out))
    full = json.dumps(out)
    if line != full:
        print("bench.py full record: " + full, file=sys.stderr)
    _RESULT.write(line + "\n")
    _RESULT.flush()


def self_launch(args):
    """`python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset): this process becomes the launcher -- one rank per GPU, each a
    fresh `python bench.py <same arguments>` with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set the way torch.distributed.run sets them (the ranks
    rendezvous over ccv_amd/ctl.py's Unix socket, named after MASTER_PORT and the run id; RCCL is bootstrapped from rank 0's id).  Fails loudly when
    fewer devices are visible than ranks were asked for.  Rank 0's JSON line is the run's; any rank failing takes the run down with its code.
    The driver's contract launches N ranks itself; this covers the other spelling so that a SCALE record can never silently be a 1-GPU number."""
    import socket
    import subprocess
    L = nnc.load()
    have = L.device_count()
    if have < args.gpus:
        raise SystemExit("bench.py: --gpus %d but only %d HIP device(s) visible" % (args.gpus, have))
    with socket.socket() as s:  # a free port: names the control socket (and is what a launcher would have passed)
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    run_id = "self%d_%d" % (os.getpid(), int(time.time()))
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   TORCHELASTIC_RUN_ID=run_id, NNC_BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=_RESULT.fileno() if r == 0 else 2))
    rc = 0
    for r, p in enumerate(procs):
        try:
            code = p.wait(timeout=3000)
        except subprocess.TimeoutExpired:
            code = -9
        if code != 0 and rc == 0:
            rc = code
            print("bench.py: rank %d exited with %d; stopping the other ranks" % (r, code), file=sys.stderr)
            for q in procs:
                if q.poll() is None:
                    q.terminate()  # (exactly the processes started above)
    if rc != 0:
        raise SystemExit(rc if rc > 0 else 1)


def main():
    claim_stdout()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (the metric is quoted at 256)")
    ap.add_argument("--config", default="vggd-train-bs256", choices=["vggd-train-bs256", "vggd-fwd-bs64", "resnet50-nchw-bs256", "resnet50-nchw-bs256-f16", "cifar10-dawn-f16-bs512", "cifar10-dawn-f32-bs512", "imdb-lstm-bs64"],
                    help="BASELINE.json configs: 3 (default, the metric), 2 (VGG-D forward only, batch 64), 4 (ResNet-50 v1d NCHW through the reference host; -f16 = the trainer's own precision), 5 (CIFAR-10 DawnNet fp16, batch 512, through the reference host)")
    ap.add_argument("--dp", default="process", choices=["process", "host"], help="configs 4 / 5 at --gpus N > 1: one harness process per GPU (default), or the reference host's own single-process data parallelism (one host thread enqueues for all N devices)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt-leg", action="store_true", help="skip the second timed leg with the other ReLU setting (profiling runs: one kind of step in the trace)")
    ap.add_argument("--no-fuse-relu", action="store_true", help="issue RELU_FORWARD as its own command behind every convolution (the reference host's graph does) instead of letting the convolution's epilogue rectify (NNC_MI355X_CONV_ALGO_FUSE_RELU); the other setting is always timed beside the headline one")
    ap.add_argument("--no-extra-configs", action="store_true", help="the default run (config 3, one GPU) ends with a short run of configs 2, 4, 4-f16 and 5 whose lines it carries in `extra_configs`; this skips them")
    ap.add_argument("--no-capture", action="store_true", help="skip the HIP-graph capture legs (SURVEY.md 8(f)3): time only the steps issued command by command")
    ap.add_argument("--no-via-host", action="store_true", help="skip the second driver: the same step through the reference host's symbolic graph / autotune / static schedule (tools/host_vgg_bench.c)")
    ap.add_argument("--records", default=None, help="also write the per-launch contraction records (name, dims, ms, TFLOP/s) of the roofline leg to this file")
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    fwd_only = args.config == "vggd-fwd-bs64"
    if fwd_only and args.batch == 256:
        args.batch = 64
    if args.config.startswith("cifar10") and args.batch == 256:
        args.batch = 512
    if args.config == "imdb-lstm-bs64":
        if rank == 0:
            lstm_config(args)
        return
    if args.config.startswith("resnet50") or args.config.startswith("cifar10"):
        if world > 1 and rank != 0:  # started under torch.distributed.run: this path is ONE process driving all the GPUs (the reference host's own data parallelism)
            return
        have = nnc.load().device_count()  # fail loudly without the HIP library / a GPU
        if have < args.gpus:
            raise SystemExit("bench.py: --gpus %d but only %d HIP device(s) visible" % (args.gpus, have))
        return resnet_config(args, "f16" in args.config, dawn=args.config.startswith("cifar10"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE %d" % (args.gpus, world))
    if (args.gpus > 1 or os.environ.get("NNC_BENCH_FORCE_SELF_LAUNCH") == "1") and "WORLD_SIZE" not in os.environ:
        return self_launch(args)  # (FORCE_SELF_LAUNCH: the launcher path at N = 1, for the one-GPU test box)

    L = nnc.load()  # raises without the HIP library / a GPU: there is no fallback path
    L.set_device(local_rank)
    dist = None
    comm = None
    force_comm = world == 1 and os.environ.get("NNC_BENCH_FORCE_COMM") == "1"  # exercise the N > 1 code path (RCCL communicator of ONE) on a 1-GPU box
    if world > 1 or force_comm:
        # one node: RCCL's bootstrap sockets go over loopback, no InfiniBand probing (the boxes have no external network)
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        os.environ.setdefault("NCCL_IB_DISABLE", "1")
        # control plane only (RCCL id hand-over, barriers, max over the ranks' clocks): ccv_amd/ctl.py, a Unix socket between the
        # ranks torch.distributed.run started -- this process never imports torch (its wheel's second HIP runtime, see ctl.py)
        from ccv_amd.ctl import LocalControl
        from ccv_amd.comm import ProcessComm
        dist = LocalControl(rank, world, deadline_s=float(os.environ.get("NNC_MI355X_CTL_DEADLINE_S", "3000")))  # a bench run is a bounded job: a rank that is alive but stuck ends it (exit 71)
        comm = ProcessComm(L, dist, rank, world)

    # parameters, images and labels from the counter hash tools/host_vgg_bench.c uses too: the command driver (this process),
    # the driver through the reference host (--via-host) and the CPU oracle all run on identical numbers
    net = VGGD(L, args.batch, device=local_rank, init="hash", flat_grads=(world > 1 or force_comm), fuse_relu=not args.no_fuse_relu,
               sgd=(0, 0.001, 1.0 / (args.batch * world), 0.0005, 0.9, 0.9))
    imgs = hash_unit(args.batch * 225 * 225 * 3, 1000 + 2 * rank).reshape(args.batch, 225, 225, 3)
    labels = (hash_unit(args.batch, 1001 + 2 * rank) * np.float32(1000)).astype(np.int32)
    net.set_input(imgs, labels)
    image0, label0 = imgs[0].copy(), int(labels[0])
    del imgs
    stream = L.stream_new(local_rank)
    L.stream_wait(None)  # construction-time SET commands ran on the default stream
    comm_stream = None
    if comm:
        comm.broadcast_params(net, stream)
        L.stream_wait(stream)
        comm_stream = L.stream_new(local_rank)  # the exchange overlaps backward on its own HIP stream
        comm.plan_overlap(net, comm_stream)

    def step():
        net.forward(stream)
        if fwd_only:
            return
        if comm:
            net.backward(stream, after_node=lambda i: comm.after_backward_node(net, i, stream))
            comm.finish_overlap(stream)
        else:
            net.backward(stream)
        net.update(stream)

    def barrier():
        L.stream_wait(stream)
        if dist:
            dist.barrier()

    # Step 1 runs on the initial (seed 0) weights: image 0's loss after it is checked against the oracle's further down
    # (read back between warm-up steps, outside the timed region; with --warmup 0 there is no untimed step and no check).
    step1_loss = None
    step1_losses = step1_params = None
    dp_check = None

    def checked_first_step():
        """N > 1, warm-up step 1, NOT overlapped: the exchange's sum identity on real gradients -- the all-reduced gradient of two
        probe tensors (the last layer's bias, the first layer's filters) must equal the sum of the ranks' local gradients
        (gathered over the control plane), and after the update every rank must hold the same parameters (sum / sum of squares per tensor)."""
        net.forward(stream)
        net.backward(stream)
        L.stream_wait(stream)
        probes = [net.params[-1][1], net.params[0][1]]
        local = [t.numpy().astype(np.float64) for t in probes]
        comm.allreduce_grads(net, stream)
        L.stream_wait(stream)
        reduced = [t.numpy().astype(np.float64) for t in probes]
        net.update(stream)
        L.stream_wait(stream)
        sums = [(float(p.numpy().astype(np.float64).sum()), float((p.numpy().astype(np.float64) ** 2).sum())) for p, _, _ in net.params]
        gathered = [None] * world
        dist.all_gather_object(gathered, (local, sums))
        try:
            err = 0.0
            for k in range(len(probes)):
                want = sum(g[0][k] for g in gathered)
                err = max(err, float(np.abs(reduced[k] - want).max() / max(np.abs(want).max(), 1e-30)))
            rep = max(abs(g[1][j][1] - gathered[0][1][j][1]) / max(gathered[0][1][j][1], 1e-30) for g in gathered for j in range(len(sums)))
            return {"allreduce_vs_sum_of_local_gradients_max_rel_err": err, "replica_param_sumsq_max_rel_diff": rep, "ok": bool(err <= 1e-5 and rep <= 1e-6)}
        except Exception as e:  # the check must never take the benchmark down
            return {"ok": False, "error": str(e)}

    for i in range(args.warmup):
        if i == 0 and comm and not fwd_only:
            dp_check = checked_first_step()
        else:
            step()
        if i == 0:
            L.stream_wait(stream)
            step1_losses = net.loss.numpy()[:8].astype(np.float64)
            step1_loss = float(step1_losses[0])
            if rank == 0 and world == 1 and not args.no_via_host and fwd_only:
                step1_params = []
            elif rank == 0 and world == 1 and not args.no_via_host:
                step1_params = [(float(p.numpy().astype(np.float64).sum()), float((p.numpy().astype(np.float64) ** 2).sum())) for p, _, _ in net.params]
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if dist:
        dt = dist.reduce_max(dt)
    loss = float(net.loss.numpy().mean())

    # HIP-graph capture of the step: the same step recorded once and replayed K times -- one runtime call per step instead of ~150 commands.
    # One GPU only here (the N-GPU form's gradient exchange goes through torch.distributed, outside this library's streams).  Reported beside the headline.
    dt_cap = cap_nodes = None
    if world == 1 and not args.no_capture:
        if L.capture_begin(stream) == 0:
            step()
            graph = L.capture_end(stream)
            if graph:
                cap_nodes = L.graph_node_count(graph)
                L.graph_launch(graph, stream)
                barrier()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    L.graph_launch(graph, stream)
                barrier()
                dt_cap = time.perf_counter() - t0
                L.graph_free(graph)

    # the same K steps with the other ReLU setting (results are bit-identical: tests/test_vgg_step.py), reported beside the headline.
    # The library's look-ahead (peephole.cpp) would fold the separately issued ReLUs as well -- it is what gives the reference host the
    # same saving (via_host) -- so it is switched off for this leg: what is timed is the step with every ReLU as its own pass.
    dt_alt = None
    if not args.no_alt_leg:
        net.fuse_relu = not net.fuse_relu
        L.dll.nnc_mi355x_set_peephole(0)
        step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        dt_alt = time.perf_counter() - t0
        if dist:
            dt_alt = dist.reduce_max(dt_alt)
        net.fuse_relu = not net.fuse_relu
        L.dll.nnc_mi355x_set_peephole(1)

    # roofline leg: three more steps with every contraction launch bracketed by HIP events on its stream; per launch position
    # the MEDIAN of the three (one stalled launch -- an allocator call, a clock dip -- would otherwise skew a kernel's average)
    LEG = 3
    L.profile_enable(1)
    for _ in range(LEG):
        step()
    L.stream_wait(stream)
    allrecs = L.profile_records()
    L.profile_enable(0)
    per = len(allrecs) // LEG
    recs = []
    for i in range(per):
        trio = [allrecs[j * per + i] for j in range(LEG)]
        assert all(t[0] == trio[0][0] and t[4] == trio[0][4] for t in trio), "launch sequence differs between steps"
        recs.append(sorted(trio, key=lambda t: t[3])[LEG // 2])
    if args.records and rank == 0:
        with open(args.records, "w") as f:
            for name, fl, _by, ms, dims in recs:
                f.write("%-12s M=%-8d N=%-6d K=%-8d Z=%d S=%-3d %9.3f ms %7.2f TFLOP/s  %s\n" % (name.split("|")[0], dims[0], dims[1], dims[2], dims[3], dims[4], ms, fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0, name.split("|")[-1][-60:]))
    # per KERNEL (the symbol behind '|'): the fused Winograd kernel serves forward and the data gradient under two command names
    # (and, with ReLU folded in, two instantiations of one source: the data gradient's epilogue-mask variant <.., 0, true> is filed with
    # its plain sibling -- same kernel, one more epilogue option, like the forward's max(0, .))
    by = {}
    for name, fl, _by, ms, dims in recs:
        k = by.setdefault(name.split("|", 1)[-1].replace(", 0, true>", ">"), [0.0, 0.0, 0])
        k[0] += fl
        k[1] += ms
        k[2] += 1
    dom = max(by.items(), key=lambda kv: kv[1][1]) if by else None
    total_fl = sum(v[0] for v in by.values())
    total_ms = sum(v[1] for v in by.values())

    if rank == 0:
        ff, fb = vgg_d_flops_per_image()
        value = world * args.batch * args.steps / dt
        out = {
            "metric": "images/sec fwd VGG-D 224x224 bs%d" % args.batch if fwd_only else "images/sec fwd+bwd VGG-D 224x224 bs256",
            "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            # fp32 tensors and fp32 accumulation everywhere; the plain-matrix products of the Winograd domain and the fc layers run on the bf16 matrix pipe with every
            # operand split EXACTLY into three bf16 values and all nine partial products kept (mfma_gemm_bf16x3.h; NNC_MI355X_GEMM_BF16X3=0: the fp32 instructions)
            "dtype": "f32 (Winograd-domain / fc GEMMs: exact bf16x3 split, all 9 products, fp32 accumulate)" if L.tune_get("GEMM_BF16X3") > 0 else "f32", "data": "synthetic",
            "config": {"workload": "VGG-D (ccv vgg_d_params, 225x225x3 crop, NHWC) %s, batch %d per GPU, random-init weights" % ("forward only" if fwd_only else "forward+backward+SGD", args.batch),
                       "global_batch": args.batch * world, "parallelism": "dp%d" % world,
                       "captured_step": ({"images_per_s": world * args.batch * args.steps / dt_cap, "ms_per_step": 1e3 * dt_cap / args.steps, "graph_nodes": cap_nodes, "note": "one hipGraphLaunch per step; `value` is the step issued command by command"} if dt_cap else None),
                       "gflop_per_image": (ff if fwd_only else fb) / 1e9, "whole_step_tflops_per_gpu": value / world * (ff if fwd_only else fb) / 1e12, "final_loss": loss,
                       "conv_relu": "convolution epilogue (NNC_MI355X_CONV_ALGO_FUSE_RELU)" if net.fuse_relu else "separate RELU_FORWARD commands"},
        }
        if dt_alt:
            out["relu_as_separate_passes" if net.fuse_relu else "relu_folded_in"] = {"value": world * args.batch * args.steps / dt_alt, "unit": "images/s", "ms_per_step": 1e3 * dt_alt / args.steps}
        if dom:
            name, (fl, ms, cnt) = dom
            traffic = pmc_traffic(name, args.batch)
            ach = fl / (ms * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "achieved": ach, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP32_MFMA_PEAK_TFLOPS,
                               "traffic": traffic, "kernel": name, "launches": cnt, "avg_ms": ms / cnt,
                               "all_contractions": {"achieved": total_fl / (total_ms * 1e-3) / 1e12, "ms": total_ms,
                                                    "by_kernel": {k: {"tflops": v[0] / (v[1] * 1e-3) / 1e12, "ms": v[1], "launches": v[2]} for k, v in by.items()}}}
        if comm is not None:
            out["config"]["rccl_ranks"] = int(L.dll.nnc_mi355x_comm_count())  # ncclCommCount of the communicator the gradients went through
            out["config"]["launcher"] = "self (bench.py started the ranks)" if os.environ.get("NNC_BENCH_SELF_LAUNCHED") == "1" else "external (torch.distributed.run)"
        if dp_check is not None:
            out["config"]["data_parallel_check"] = dp_check
            if not dp_check.get("ok"):
                print("bench.py: the data-parallel exchange FAILED its sum identity / replica equality check: %r" % (dp_check,), file=sys.stderr)
        if step1_params is not None:
            out["config"]["via_host"] = via_host(args, step1_losses, step1_params, fwd_only)
            # the drop-in number (the unmodified reference host driving this backend), next to `value` (ccv_amd/vgg.py, the command driver)
            out["via_host_images_per_s"] = out["config"]["via_host"].get("images_per_s")
            out["config"]["via_host_images_per_s"] = out["via_host_images_per_s"]  # (also inside `config`: the driver's parsed record keeps that object whole)
        if not args.no_cpu_baseline and world > 1:
            out["cpu_baseline"] = {"value": None, "unit": "images/s", "cores": 0, "kind": "reference", "sample": "timed at N = 1 only (python bench.py --gpus 1)"}
        elif not args.no_cpu_baseline:
            try:
                out["cpu_baseline"], oracle_loss = cpu_baseline(image0, label0, fwd_only)
            except Exception as e:  # the checker is optional for the bench line, never for the parity tests
                out["cpu_baseline"], oracle_loss = {"value": None, "unit": "images/s", "cores": 0, "kind": "reference", "sample": "unavailable: %s" % e}, None
            if oracle_loss is not None and step1_loss is not None:
                # parity gate on the benchmarked configuration itself: image 0's loss after the first forward at batch 256 (the
                # convolutions under the algorithms the timed steps use) vs the reference CPU backend on the same image and weights
                rel = abs(step1_loss - oracle_loss) / max(abs(oracle_loss), 1e-30)
                out["config"]["step1_loss_image0"] = {"gpu": step1_loss, "oracle": oracle_loss, "rel_err": rel, "bound": 1e-4}
                if not rel <= 1e-4:
                    emit(out)
                    raise SystemExit("bench.py: step-1 loss of image 0 differs from the oracle's: %r vs %r (rel %.3g > 1e-4)" % (step1_loss, oracle_loss, rel))
        if args.config == "vggd-train-bs256" and world == 1 and not args.no_extra_configs and not force_comm:
            L.stream_wait(stream)
            out["extra_configs"] = extra_configs(args)
        emit(out)
    if dist:
        L.stream_wait(stream)
        if comm_stream is not None:
            L.stream_wait(comm_stream)
        dist.barrier()  # every rank is past its last collective
        L.dll.nnc_mi355x_comm_destroy()
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
