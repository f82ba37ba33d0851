"""Rehearsal of `bench.py --gpus 2` on a one-GPU box: two ranks launched by
torch.distributed.run exactly as the driver does, both on cuda:0, with the gloo backend
standing in for RCCL (two ranks cannot share one device under RCCL).  Everything else is
the real N>1 path: stripe sharding in the engines, the gather, the assembly, the counters."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, nproc, tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable]
    if nproc > 1:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
                "--master-port", "29571"]
    cmd += [os.path.join(ROOT, "bench.py")] + args
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_two_ranks_reproduce_the_single_rank_frame(tmp_path):
    common = ["--workload", "c1", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0"]
    one = _run(common + ["--gpus", "1", "--dump-frame", str(tmp_path / "one.npy")], 1, tmp_path)
    two = _run(common + ["--gpus", "2", "--backend", "gloo", "--same-device", "--dump-frame", str(tmp_path / "two.npy")],
               2, tmp_path)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert one["config"]["segments_per_step"] == two["config"]["segments_per_step"]
    assert two["config"]["paths_per_step"] == 512 * 512 * 64
    for j in (one, two):
        assert j["unit"] == "Msamples/s" and j["value"] > 0
        roof = j["roofline"]
        assert roof["kernel_ms"] > 0 and roof["algorithmic"]["GBps"] > 0
        # the binding ceiling comes from a PMC profile of THIS build or not at all: never above 1
        assert roof["frac"] is None or 0.0 < roof["frac"] <= 1.0
        assert j["steps"] == 2 and j["warmup"] == 1 and j["scaling"] == "strong"
    assert np.array_equal(np.load(tmp_path / "one.npy"), np.load(tmp_path / "two.npy"))
    # the self-verification fields the first real multi-GPU run will be read by
    v1, v2 = one["verify"], two["verify"]
    assert v1["frame_crc32"] == v2["frame_crc32"] and v1["frame_crc32"] is not None
    assert len(v2["trace_ms_per_step_by_rank"]) == 2 and min(v2["trace_ms_per_step_by_rank"]) > 0
    assert v2["trace_ms_per_step_min"] <= v2["trace_ms_per_step_max"]
    assert v2["rccl_ranks_reported"] == [0, 0]          # the gloo rehearsal: nothing went through RCCL, and the line says so
    assert "torch.distributed" in v2["exchange"]
    e2e = one["end_to_end"]
    assert two["end_to_end"] is None and e2e["frame_equals_timed_frame"] is True
    assert e2e["total_ms"] >= e2e["render_ms"] > 0 and e2e["readback_ms"] > 0 and e2e["upload_ms"] > 0


@pytest.mark.parametrize("workload", ["c4", "c5"])
def test_the_eight_gpu_configurations_shard_too(workload, tmp_path):
    # BASELINE configs 4 and 5 are the ones quoted on 8 GPUs: one sample per pixel of their real frames, two ranks on one
    # device, against the single-rank frame
    common = ["--workload", workload, "--spp", "1", "--steps", "1", "--warmup", "0", "--cpu-seconds", "0", "--no-stats", "--no-end-to-end"]
    one = _run(common + ["--gpus", "1"], 1, tmp_path)
    two = _run(common + ["--gpus", "2", "--backend", "gloo", "--same-device"], 2, tmp_path)
    assert one["verify"]["frame_crc32"] == two["verify"]["frame_crc32"] is not None
    assert one["config"]["segments_per_step"] == two["config"]["segments_per_step"] > 0
    assert two["n_gpus"] == 2 and len(two["verify"]["trace_ms_per_step_by_rank"]) == 2
