"""bench.py's N > 1 path on a one-GPU box: two ranks on cuda:0 (torch.distributed over gloo, `--test-one-gpu`) cut ONE stream into two key
ranges -- global ticks, flush events, no data-path collective, one all-reduce of the Stats blocks -- and must report the Stats of the same
stream run through a single engine."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def run(cmd):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert p.returncode == 0 and lines, (p.returncode, p.stdout[-2000:], p.stderr[-3000:])
    return json.loads(lines[-1])


def test_two_ranks_equal_one_engine(built):
    common = ["--workload", "cfg4s", "--scale", "0.02", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"]
    one = run([sys.executable, "bench.py", "--pairs", "300000"] + common)
    # `bench.py --gpus 2` by itself: it starts its own two ranks (torch.distributed.run) -- the form the driver uses for N = 1
    two = run([sys.executable, "bench.py", "--gpus", "2", "--pairs", "150000", "--test-one-gpu"] + common)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "weak"
    a, b = one["stats_whole_stream"], two["stats_whole_stream"]
    assert a["pre"]["reads"] == 600000 or a["pre"]["reads"] > 0
    for blk in ("pre", "post"):
        assert a[blk] == b[blk], (blk, a[blk], b[blk])
    assert a["pre_hist_sum"] == b["pre_hist_sum"] and a["post_hist_sum"] == b["post_hist_sum"]
    # the depth bins and BED region counts travel in the same buffer (gce_stats_payload_device): sums and position-weighted checksums of the four vectors
    assert a["depth"] == b["depth"] and a["depth"]["pre_depth"]["sum"] > a["depth"]["post_depth"]["sum"] > 0 and a["depth"]["pre_bed"]["sum"] > 0
    assert b["depth"]["payload_bytes"] == 8 * (228 + 2 * b["depth"]["depth_bins"] + 2 * b["depth"]["bed_regions"])
    assert two["value"] > 0 and two["config"]["pairs_per_gpu"] > 0


def test_driver_form_under_the_launcher(built):
    """the driver's N > 1 form: bench.py under torch.distributed.run; --gpus must agree with the ranks the launcher started"""
    common = ["--workload", "cfg4s", "--scale", "0.02", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--pairs", "60000", "--test-one-gpu"]
    two = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29533",
               "bench.py", "--gpus", "2"] + common)
    assert two["n_gpus"] == 2
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29534",
                        "bench.py", "--gpus", "4"] + common, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode != 0 and "--gpus 4 but the launcher started 2 ranks" in (p.stdout + p.stderr)


def test_world_of_one_over_rccl_equals_one_engine(built):
    """the N > 1 code path as a world of ONE rank over the nccl backend (= RCCL on ROCm): the key-range plan, global ticks + flush events and -- what no
    gloo test reaches -- the RCCL int64 all-reduce of the Stats payload straight from device memory, loaded and executed on this one GPU"""
    common = ["--workload", "cfg4s", "--scale", "0.02", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--pairs", "200000"]
    one = run([sys.executable, "bench.py"] + common)
    w1 = run([sys.executable, "bench.py", "--nccl-world1"] + common)
    assert w1["n_gpus"] == 1 and w1["backend"].startswith("nccl") and w1["stats_merge_ms_per_step"] > 0
    a, b = one["stats_whole_stream"], w1["stats_whole_stream"]
    for blk in ("pre", "post"):
        assert a[blk] == b[blk], (blk, a[blk], b[blk])
    assert a["depth"] == b["depth"] and a["pre_hist_sum"] == b["pre_hist_sum"] and a["post_hist_sum"] == b["post_hist_sum"]
    assert "all-reduce" in w1["step_includes"]


def test_record_layouts_in_device_memory_give_the_same_stream_statistics(built):
    """bench.py --layout record / record64: ONE device blob, both base pointers of the batch the same buffer (gce_submit_device, zero copy) -- the engine must
    give the Stats blocks, histogram sums, depth and BED digests of the two-blob layout (the record-level parity of free-form offsets: test_gpu_parity.py)"""
    common = ["--workload", "cfg3", "--scale", "0.02", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--pairs", "200000"]
    base = run([sys.executable, "bench.py"] + common)["stats_whole_stream"]
    for lay in ("record", "record64"):
        got = run([sys.executable, "bench.py", "--layout", lay] + common)["stats_whole_stream"]
        assert got == base, lay
