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
