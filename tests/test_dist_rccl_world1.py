"""GPU: the sharded search's collectives executed through RCCL once, with a process group of one rank.

`dist.py`'s helpers short-circuit at world size 1 and the gloo tests stage device tensors through the host, so before this test the
calls an 8-GPU run makes -- `all_gather_into_tensor` on device tensors, the device branch of `require_equal_rows`, asynchronous
`work.wait()` stream ordering, `all_reduce`, the merge over [world, Q, k] candidates -- had never executed.  The child process
initialises backend "nccl" (= RCCL on ROCm) BEFORE anything else touches the GPU, forces the collectives on, and compares
`ShardedGallery.search / search_many / ranks` bit for bit with the same calls made without a process group."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_sharded_gallery_through_rccl_at_world_size_one(device):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), MASTER_ADDR="127.0.0.1")
    env.pop("KEMR_DIST_BACKEND", None)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_world1_child.py")], capture_output=True, text=True,
                          timeout=600, env=env, cwd=ROOT)
    assert proc.returncode == 0, (proc.returncode, proc.stdout[-2000:], proc.stderr[-4000:])
    report = json.loads([l for l in proc.stdout.splitlines() if l.startswith("{")][-1])
    assert report["backend"] == "nccl" and report["world"] == 1
    assert all(report["equal"].values()) and report["helpers_ok"], report
    calls = report["calls"]
    # every collective ran on DEVICE tensors (no host staging), some of the all-gathers asynchronously
    assert calls["all_gather_into_tensor"]["n"] >= 20 and calls["all_gather_into_tensor"]["cuda"] == calls["all_gather_into_tensor"]["n"]
    assert calls["all_gather_into_tensor"]["async"] >= 10
    assert calls["all_reduce"]["n"] >= 8 and calls["all_reduce"]["cuda"] == calls["all_reduce"]["n"]
    assert report["top1_hit"] > 0.9
    out = os.environ.get("KEMR_RCCL_JSON")
    if out:
        with open(out, "w") as f:
            json.dump(report, f, indent=1)
