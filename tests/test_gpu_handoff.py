"""GPU (-m gpu): the inter-workgroup hand-off of the inverse-GEMM solve (tri_gemm_chi2_kernel / tri_gemm_small_kernel: agent-scope
write-through stores, s_waitcnt vmcnt(0), ONE relaxed agent-scope add, the last arriver's acquire fence).

* the relaxed arrival add rests on hand-written ordering (sc1 stores + an asm wait), not on the C++ memory model alone;
  a build whose add is a RELEASE (-DCF_HANDOFF_RELEASE, 15-20 % slower: profiles/r02_handoff_and_traffic_ab.txt) must give
  bit-identical chi^2 at W = 4096 and W = 65536 -- a compiler or ROCm upgrade that reorders the relaxed form shows up here.
* the hand-off under a deliberately SKEWED co-runner: a second stream keeps part of the chip busy with
  unrelated FP64 matrix and streaming kernels while evaluations run; every word of every result must equal the quiet run's."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
PKG_DIR = os.path.join(ROOT, "cosmology-model-fit_amd")


@pytest.fixture(scope="module")
def gpu(pkg):
    if pkg.lib().cf_device_count() < 1:
        pytest.fail("GPU tests need an MI355X; no HIP device visible (there is no fallback path)")
    return pkg


def _chi2_with(lib_path, W, tmp_path, tag, tune=None):
    out = str(tmp_path / f"{tag}_{W}.npy")
    env = dict(os.environ, COSMOFIT_LIB=lib_path)
    if tune is not None:
        env["CF_TUNE"] = tune
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "handoff_worker.py"), str(W), out], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(out)


def test_release_ordered_arrival_gives_the_same_bits(gpu, tmp_path):
    variant = os.path.join(PKG_DIR, "libcosmofit_hip_handoff_release.so")
    if not os.path.exists(variant):  # normally built by __graft_entry__.build(); hipcc cross-compiles in ~30 s
        subprocess.run([os.path.join(ROOT, "tools", "build_variant.sh"), "handoff_release", "-DCF_HANDOFF_RELEASE"], check=True,
                       stdout=subprocess.DEVNULL)
    default = os.path.join(PKG_DIR, "libcosmofit_hip.so")
    for W in (16, 4096, 65536):  # the small-batch kernel, the throughput kernel in one panel group and in 16
        a = _chi2_with(default, W, tmp_path, "relaxed")
        b = _chi2_with(variant, W, tmp_path, "release")
        assert a.shape == (W,) and np.all(np.isfinite(a)) and np.all(a > 0)
        assert np.array_equal(a, b), f"W = {W}: the relaxed and the release-ordered hand-off disagree"


def test_skipping_zero_tiles_and_splitting_units_changes_no_bit(gpu, tmp_path):
    """Round 4: the throughput solve kernel does not multiply the all-padding tiles of the last row block nor the zero tiles of the
    diagonal blocks, and works the lowest row blocks of a panel as two half units at ~900-1800 walkers.  With every one of these
    switched off through CF_TUNE (a fresh process: the string is read once) the chi^2 must be the same BITS -- an accumulator that
    receives 0 x b keeps its value, a half unit computes its walkers exactly as the whole unit would."""
    default = os.path.join(PKG_DIR, "libcosmofit_hip.so")
    for W in (1040, 1500, 4096):  # half units with an empty / a partly filled second half; whole units
        a = _chi2_with(default, W, tmp_path, "as_shipped")
        b = _chi2_with(default, W, tmp_path, "everything_multiplied", tune="gemm_trim=0,gemm_diag_skip=0,gemm_split=0")
        c = _chi2_with(default, W, tmp_path, "twelve_split_levels", tune="gemm_split=12")
        assert a.shape == (W,) and np.all(np.isfinite(a)) and np.all(a > 0)
        assert np.array_equal(a, b) and np.array_equal(a, c), f"W = {W}"


def test_handoff_under_a_skewed_co_runner(gpu):
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    syn = gpu.synthetic.pantheon_like(n_sn=1701, seed=0)
    lk = gpu.sn_pantheon.PantheonLikelihood(syn["z_cmb"], syn["z_hel"], syn["obs"], chol=syn["chol"], solve="inverse")
    f = lk.engine.torch_log_prob(gpu.CF_OUT_CHI2)
    thetas = {W: torch.from_numpy(gpu.synthetic.walkers(gpu.sn_pantheon.bounds, W, seed=W)).to(dev) for W in (16, 48, 1000, 4096, 16384)}
    quiet = {W: f(t).clone() for W, t in thetas.items()}
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev)
    a = torch.randn(3072, 3072, dtype=torch.float64, device=dev)
    big = torch.randn(1 << 26, dtype=torch.float64, device=dev)
    main = torch.cuda.current_stream(dev)
    bad = []
    for rep in range(6):
        with torch.cuda.stream(side):  # unrelated work of very different shapes: FP64 GEMMs (matrix cores), a streaming pass, a tiny op
            for _ in range(3):
                c = a @ a
                big.mul_(1.0000001)
                c[:7, :5].add_(1.0)
        for W, t in thetas.items():
            got = f(t)
            if not torch.equal(got, quiet[W]):
                bad.append((rep, W, int((got != quiet[W]).sum())))
        main.synchronize()
    torch.cuda.synchronize()
    assert not bad, f"results changed under the co-runner: (round, W, differing walkers) = {bad}"
    lk.engine.close()
