"""CPU: bench.py's launch logic.  `python bench.py --gpus N` must start its N ranks itself (torch.distributed.run from a parent
that never touches a GPU) and, on a box without a GPU, every rank must refuse loudly -- there is no CPU path to time instead."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _run(args, env=None, timeout=180):
    e = dict(os.environ)
    e.pop("RANK", None), e.pop("WORLD_SIZE", None), e.pop("LOCAL_RANK", None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=e)


def test_single_process_refuses_without_a_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    r = _run(["--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and "needs an MI355X" in (r.stderr + r.stdout)
    assert "{" not in r.stdout, "no JSON line may be printed without a measurement"


def test_gpus_n_self_launches_its_ranks():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    out = r.stderr + r.stdout
    assert r.returncode != 0
    assert "must be launched with torch.distributed.run" not in out, "round-1 behaviour: bench.py could not start N > 1 by itself"
    # both child ranks are started and each refuses for lack of a GPU; torchrun ends the surviving sibling as soon as the first
    # child has failed, so the second refusal may not get printed: one refusal + torchrun's own failure report is the proof
    assert out.count("needs an MI355X") >= 2 or ("needs an MI355X" in out and "ChildFailedError" in out), out[-1500:]


def test_strong_scaling_needs_whole_panels_per_rank():
    r = _run(["--help"])
    assert r.returncode == 0 and "--scaling" in r.stdout and "--walkers-total" in r.stdout and "--fde" in r.stdout


def test_one_per_gpu_shape_at_every_world_size_and_the_inprocess_arguments():
    """the weak-scaling default keeps 4096 walkers per GPU at N = 1, 2, 4 AND 8 (round 3 switched to 8192 at
    N = 8 under the same label); configs[3] (65536 walkers over 8 GPUs) is `--scaling strong`; `--mode inprocess` is one process over
    a device list, and a repeated ordinal is labelled a rehearsal."""
    sys.path.insert(0, ROOT)
    import bench

    for n in (1, 2, 4, 8):
        p = bench.plan("ranks", "weak", n)
        assert p["walkers_per_gpu"] == 4096 and p["walkers_total"] == 4096 * n and p["scaling"] == "weak"
    assert bench.plan("ranks", "weak", 8, walkers_per_gpu=8192)["walkers_total"] == 65536
    p = bench.plan("ranks", "strong", 8)
    assert p["walkers_per_gpu"] == 8192 and p["walkers_total"] == 65536
    with pytest.raises(ValueError, match="multiple of 32"):
        bench.plan("ranks", "strong", 3, walkers_total=65536)
    p = bench.plan("inprocess", "weak", 1, devices="all")
    assert p["devices"] == "all" and p["walkers_total"] == 65536 and not p["rehearsal"]
    p = bench.plan("inprocess", "weak", 1, walkers_total=8192, devices="0,0,1")
    assert p["devices"] == [0, 0, 1] and p["rehearsal"] and p["walkers_total"] == 8192
    assert not bench.plan("inprocess", "weak", 1, devices="0,1,2,3")["rehearsal"]
    for bad in ("", "a", "-1"):
        with pytest.raises(ValueError):
            bench.plan("inprocess", "weak", 1, devices=bad)
    with pytest.raises(ValueError, match="ONE process"):
        bench.plan("inprocess", "weak", 2)
    # the arithmetic behind roofline.useful_over_executed: N = 1701 -> 27 row blocks, 64 (rb + 1) MFMAs per 16-walker panel and block
    assert bench.executed_mfmas_solve(1701, 4096, skip_padding=False) == 256 * 64 * 378 == 6193152  # (= the PMC count of round 4's first pass)
    assert abs(bench.flops_per_eval_solve(1701) * 16 / (64 * 378 * 2048.0) - 0.9355) < 1e-4
    # less one all-padding tile of row block 26 (16 x 27 K-steps) and six zero tile-groups of 4 K-steps per diagonal block
    assert bench.executed_mfmas_solve(1701, 16) == 64 * 378 - 432 - 27 * 24 == 23112
    assert bench.executed_mfmas_solve(1820, 16) == 64 * 435 - 2 * 16 * 29 - 4 * (28 * 6 + 5) == 26220


def test_inprocess_mode_refuses_without_a_gpu_and_under_torchrun():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    r = _run(["--mode", "inprocess", "--devices", "0,0", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and "needs an MI355X" in (r.stderr + r.stdout) and "{" not in r.stdout
    r = _run(["--mode", "inprocess", "--steps", "1"], env={"RANK": "0", "WORLD_SIZE": "1"})
    assert r.returncode != 0 and "ONE process" in (r.stderr + r.stdout)
