"""GPU (-m gpu): the process-per-GPU launch path of bench.py on the ONE GPU of the test box.  RCCL refuses two ranks on one
device, so what can execute here is the one-rank case under torch.distributed.run: process-group creation over RCCL
(backend "nccl"), the device all-gather of walker positions in every step, the barrier / max-over-ranks timing and the
gathered device identities -- the same code the N = 2, 4, 8 runs execute, minus the inter-GPU transport.  The multi-rank
logic itself (sharding, all-gather of ragged shards, rank-count invariance) is covered over gloo in test_ensemble_gloo.py."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_under_torchrun_runs_rccl_with_one_rank(pkg):
    if pkg.lib().cf_device_count() < 1:
        pytest.fail("GPU tests need an MI355X; no HIP device visible (there is no fallback path)")
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1",
           "--no-cpu-baseline", "--precondition-ms", "0", "--walkers-per-gpu", "1024", "--n-sn", "300"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "stdout must carry ONE JSON line (RCCL's banner goes to stderr):\n" + r.stdout[:800]
    d = json.loads(lines[0])
    assert d["backend"] == "rccl (torch.distributed nccl)" and d["world_size"] == 1 and d["n_gpus"] == 1
    assert "RCCL all-gather of positions per step" in d["config"]["parallelism"]
    assert d["distinct_devices"] == 1 and len(d["device_ids"]) == 1 and d["value"] > 0
    # round 4: every rank's own time and the exchange alone are in the line; one per-GPU shape whatever N
    assert d["mode"] == "ranks" and len(d["ms_per_step_by_rank"]) == 1 and d["ms_per_step_by_rank"][0] <= d["ms_per_step"] * 1.0001
    assert d["allgather_ms_per_step"] is not None and 0 < d["allgather_ms_per_step"] < d["ms_per_step"]
    assert d["config"]["walkers_per_gpu"] == 1024 and d["scaling"] == "weak"


def test_bench_two_ranks_rehearsed_over_gloo_on_one_gpu(pkg):
    """The N = 2 code path of bench.py (sharding of the walkers, per-rank times gathered, max over ranks) with two rank PROCESSES
    sharing the box's one GPU: RCCL refuses that, so the rehearsal backend stages the all-gather through the host (labelled in the
    line).  What this cannot show is inter-GPU transport."""
    if pkg.lib().cf_device_count() < 1:
        pytest.fail("GPU tests need an MI355X; no HIP device visible (there is no fallback path)")
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["BENCH_DIST_BACKEND"] = "gloo"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
           "--no-cpu-baseline", "--precondition-ms", "0", "--walkers-per-gpu", "512", "--n-sn", "300"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[:800]
    d = json.loads(lines[0])
    assert d["world_size"] == 2 and d["n_gpus"] == 2 and d["backend"] == "gloo" and "rehearsal" in d["config"]["parallelism"]
    assert d["config"]["walkers_per_gpu"] == 512 and d["config"]["walkers_total"] == 1024
    assert len(d["ms_per_step_by_rank"]) == 2 and max(d["ms_per_step_by_rank"]) == pytest.approx(d["ms_per_step"], rel=1e-9)
    assert d["allgather_ms_per_step"] is None and d["distinct_devices"] == 1 and len(d["device_ids"]) == 2



def test_bench_inprocess_mode_over_repeated_ordinals(pkg):
    """SURVEY 8e form (1) as bench.py measures it: ONE process, one handle over k replicas, host buffers through cf_eval.  On the
    one-GPU box the ordinal repeats (a rehearsal, labelled as such in the line); the replicas must not change a walker's result
    (bench.py itself asserts bit-equality with a one-device handle) and the host CPU time per call is reported."""
    if pkg.lib().cf_device_count() < 1:
        pytest.fail("GPU tests need an MI355X; no HIP device visible (there is no fallback path)")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "inprocess", "--devices", "0,0,0", "--walkers-total", "6144",
           "--n-sn", "300", "--steps", "3", "--warmup", "1", "--precondition-ms", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[:800]
    d = json.loads(lines[0])
    assert d["mode"] == "inprocess" and d["n_gpus"] == 1 and d["distinct_devices"] == 1 and d["config"]["replicas"] == [0, 0, 0]
    assert d["config"]["rehearsal"] and d["config"]["walkers_total"] == 6144 and d["value"] > 0 and d["scaling"] == "strong"
    assert 0 < d["host_cpu_seconds_per_call"] and d["roofline"]["frac"] > 0 and d["vs_baseline"] is None
