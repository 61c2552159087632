#!/usr/bin/env python
"""Whole-network parity, for the record: the reference host on its own CPU backends (oracle/_ref/host_resnet_bench.cpu) against the same host on
this backend (host_resnet_bench.gpu), one training step from identical parameters (tests/test_via_host.py holds the bounds; this prints what
was achieved).  Run on the GPU box: python tools/whole_net_parity.py > gpurun_out/whole_net_parity.txt"""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_via_host as T  # noqa: E402

for name, batch, hw, model in (("ResNet-50 v1d 224x224 N=2 (BASELINE config 4's network)", 2, 224, "full"), ("CIFAR-10 DawnNet 32x32 N=8 (config 5's network and trainer step)", 8, 32, "dawn"),
                               ("ResNet bottlenecks (mini) 32x32 N=2", 2, 32, "mini")):
    want = T.run_check("cpu", batch, hw, 32, model)
    for dtype in (32, 16):
        got = T.run_check("gpu", batch, hw, dtype, model)
        print("%s, GPU %s vs CPU_REF fp32: %s" % (name, "f16" if dtype == 16 else "f32", json.dumps(T.check_report(want, got))))
