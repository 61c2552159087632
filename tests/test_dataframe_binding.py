"""The GPU data pipeline bound into the reference's dataframe.

integration/nnc_mi355x_dataframe.c is HOST-SIDE glue -- compiled into the reference host, against its own headers and its SFMT generator -- that stands where
the trainers chain ccv_cnnp_dataframe_image_random_jitter + one_hot + combine_new + copy_to_gpu (bin/nnc/imagenet.c:389-406, test/int/nnc/cifar.tests.c:100-126):
a sampled column whose rows are whole batches already on the device (decisions from the reference's generator in the reference's order, raw images through the
pinned staging ring, pixels by nnc_mi355x_jitter_batch, labels by nnc_mi355x_one_hot_batch).  tools/host_dataframe_test.c iterates the reference's dataframe
twice over the same synthetic images with the same seed -- the reference's own CPU stages row by row, then this stage batch by batch -- and compares.
Bounds: fp32 images to 2e-5 of their range (measured: bit-identical without colour jitter, 2e-6 with it), half to 2e-3; one-hot rows exact in fp32."""
import json
import os
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(kind, count, batch, mode, dtype=32, timeout=1500):
    exe = os.path.join(ROOT, "oracle", "_ref", "host_dataframe_test." + kind)
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/host_dataframe_test.%s not built (oracle/build_ref_host.sh needs /root/reference)" % kind)
    r = subprocess.run([exe, str(count), str(batch), mode, str(dtype)], capture_output=True, text=True, timeout=timeout, env=dict(os.environ, OMP_NUM_THREADS="4"))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def check(out, count, batch, half=False):
    assert out["images"] == count and out["batches"] == (count + batch - 1) // batch
    scale = max(1.0, out["reference_abs_max"])
    assert out["max_abs_diff"] <= (2e-3 if half else 2e-5) * scale, out
    assert out["one_hot_max_abs_diff"] <= (1e-3 if half else 0.0), out


@pytest.mark.parametrize("mode,count,batch", [("cifar", 7, 3), ("pad", 5, 2), ("imagenet", 4, 2)])
def test_dataframe_gpu_stage_matches_the_reference_pipeline_on_emulator(mode, count, batch):
    check(run("emu", count, batch, mode), count, batch)


def test_dataframe_gpu_stage_half_precision_on_emulator():
    check(run("emu", 5, 4, "cifar", 16), 5, 4, half=True)


@pytest.mark.gpu
@pytest.mark.parametrize("mode,count,batch,dtype", [("cifar", 1000, 128, 32), ("imagenet", 96, 32, 32), ("imagenet", 64, 32, 16), ("pad", 40, 16, 32)])
def test_dataframe_gpu_stage_matches_the_reference_pipeline_on_gpu(mode, count, batch, dtype):
    check(run("gpu", count, batch, mode, dtype), count, batch, half=dtype == 16)
