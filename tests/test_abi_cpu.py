"""CPU: the C-ABI library loads, exports every symbol include/cosmofit.h declares, and refuses loudly without a GPU."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, golden

HEADER = os.path.join(ROOT, "include", "cosmofit.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cf_[a-z_0-9]+)\s*\(", src)))


def test_header_compiles_as_c():
    subprocess.run(["gcc", "-std=c99", "-fsyntax-only", "-x", "c", HEADER], check=True)


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.lib()
    names = _declared_functions()
    assert len(names) >= 14
    for name in names:
        assert hasattr(lib, name), f"{name} declared in cosmofit.h but not exported"
    assert set(names) == set(pkg._lib.EXPORTS), "ctypes table and header disagree"


def test_desc_layout_matches_c(pkg, tmp_path):
    """sizeof / offsetof of cf_desc and cf_info as gcc sees them == the ctypes mirror."""
    fields = ["ndim", "z_max", "param", "n_sn", "sn_chol_ld", "n_bao", "rd_fit", "cmb_mode", "cmb_inv_cov", "nu_ws",
              "bounds", "gauss", "chi2_gauss", "sn_fixed_mu", "n_cc", "cc_logdet", "solve_mode", "probe_limit", "n_devices", "devices", "om_mode", "rd_wm_mode", "sn_lin_coef", "sn_dir", "n_fs8", "fs8_fid", "logl_const",
              "fs8_a_init", "sn_vel_mode", "cc_f_mode", "prior_norm_mode"]
    prog = '#include <stdio.h>\n#include <stddef.h>\n#include "cosmofit.h"\nint main(){printf("%zu %zu", sizeof(cf_desc), sizeof(cf_info));' + \
        "".join(f'printf(" %zu", offsetof(cf_desc, {f}));' for f in fields) + "return 0;}"
    src = tmp_path / "sz.c"
    src.write_text(prog)
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    vals = list(map(int, subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()))
    L = pkg._lib
    assert vals[0] == C.sizeof(L.cf_desc)
    assert vals[1] == C.sizeof(L.cf_info)
    for f, off in zip(fields, vals[2:]):
        assert getattr(L.cf_desc, f).offset == off, f


def test_split_rows_partitions_a_batch_into_whole_panels(pkg):
    """cf_eval over n replicas (cf_desc.n_devices): contiguous slices that cover the batch once, cut at 32-walker panels,
    at most one panel apart in size -- host arithmetic, checked without a GPU."""
    lib = pkg.lib()
    for W in (1, 31, 32, 33, 75, 4096, 4097, 65536, 100000):
        for n in (1, 2, 3, 8):
            cuts = []
            for k in range(n):
                b, e = C.c_int64(), C.c_int64()
                lib.cf_split_rows(W, n, k, C.byref(b), C.byref(e))
                cuts.append((b.value, e.value))
            assert cuts[0][0] == 0 and cuts[-1][1] == W
            assert all(cuts[k][1] == cuts[k + 1][0] for k in range(n - 1))
            assert all(b % 32 == 0 or b == W for b, _ in cuts) and all(0 <= b <= e <= W for b, e in cuts)
            sizes = [e - b for b, e in cuts]
            assert max(sizes) - min(sizes) <= 32 + 31, (W, n, sizes)
    b, e = C.c_int64(), C.c_int64()
    lib.cf_split_rows(65536, 8, 3, C.byref(b), C.byref(e))
    assert (b.value, e.value) == (3 * 8192, 4 * 8192)  # BASELINE configs[3]: 8192 walkers per GPU


def test_engine_rejects_a_bad_device_list(pkg):
    with pytest.raises(ValueError):
        pkg.LikelihoodEngine(ndim=2, z_max=1.0, params=dict(H0=pkg.Param(0), Om=pkg.Param(1)), devices="some")
    with pytest.raises(ValueError):
        pkg.LikelihoodEngine(ndim=2, z_max=1.0, params=dict(H0=pkg.Param(0), Om=pkg.Param(1)), devices=[])
    with pytest.raises(pkg.CosmofitError, match="CF_ERR_INVALID"):
        pkg.LikelihoodEngine(ndim=2, z_max=1.0, params=dict(H0=pkg.Param(0), Om=pkg.Param(1)), probe_limit=-1.0)


def test_no_gpu_is_a_loud_error_not_a_fallback(pkg):
    if pkg.lib().cf_device_count() > 0:
        pytest.skip("a GPU is visible")
    g = golden("interpolator")
    with pytest.raises(pkg.CosmofitError, match="CF_ERR_NO_DEVICE"):
        pkg.interpolator.interp_hermite(g["h_xq"], g["h_x"], g["h_y"], g["h_yp"])
    with pytest.raises(pkg.CosmofitError, match="CF_ERR_NO_DEVICE"):
        pkg.solve_triangular.solve_triangular(g["t_L"], g["t_b"][0])
    with pytest.raises(pkg.CosmofitError, match="CF_ERR_NO_DEVICE"):
        pkg.LikelihoodEngine(ndim=2, z_max=1.0, params=dict(H0=pkg.Param(0), Om=pkg.Param(1)))


def test_descriptor_validation(pkg):
    """Argument errors are reported before any device work (same on CPU and GPU boxes)."""
    with pytest.raises(pkg.CosmofitError, match="CF_ERR_INVALID"):
        pkg.LikelihoodEngine(ndim=0, z_max=1.0, params={})
    with pytest.raises(pkg.CosmofitError, match="CF_ERR_INVALID"):
        pkg.LikelihoodEngine(ndim=2, z_max=-1.0, params={})
    with pytest.raises(pkg.CosmofitError, match="CF_ERR_INVALID"):
        pkg.LikelihoodEngine(ndim=2, z_max=1.0, params=dict(H0=pkg.Param(5)))
    with pytest.raises(ValueError):
        pkg.LikelihoodEngine(ndim=2, z_max=1.0, params=dict(bogus=pkg.Param(0)))


@pytest.mark.parametrize("n", [1, 15, 16, 17, 129, 256, 257, 300, 531])
def test_packed_factor_replay_matches_forward_substitution(pkg, n):
    """Host packing logic: replaying the fragment streams == the oracle's forward substitution."""
    from oracle import oracle_c as oc

    rng = np.random.default_rng(n)
    M = rng.standard_normal((n, n))
    Lm = np.linalg.cholesky(M @ M.T + n * np.eye(n)) + np.triu(rng.standard_normal((n, n)), 1) * 3.0  # garbage above
    b = rng.standard_normal(n)
    chi2, nbytes = C.c_double(), C.c_int64()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    pkg._lib.check(pkg.lib().cf_selftest_pack_host(p(Lm), n, n, p(b), C.byref(chi2), C.byref(nbytes)))
    assert chi2.value == pytest.approx(oc.solve_triangular(Lm, b), rel=1e-12)
    assert nbytes.value > 0


@pytest.mark.parametrize("n", [1, 16, 63, 64, 65, 200, 531])
def test_inverse_pack_replay_matches_forward_substitution(pkg, n):
    """Latency-mode packing (explicit inverse, 64-row blocks x 4 K-quarters) replayed on the host."""
    from oracle import oracle_c as oc

    rng = np.random.default_rng(100 + n)
    M = rng.standard_normal((n, n))
    Lm = np.linalg.cholesky(M @ M.T + n * np.eye(n)) + np.triu(rng.standard_normal((n, n)), 1) * 3.0
    b = rng.standard_normal(n)
    chi2, probe = C.c_double(), C.c_double()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    pkg._lib.check(pkg.lib().cf_selftest_invpack_host(p(Lm), n, n, p(b), C.byref(chi2), C.byref(probe)))
    assert chi2.value == pytest.approx(oc.solve_triangular(Lm, b), rel=1e-12)
    assert probe.value < 1e-12


def test_pack_probe_flags_ill_conditioned_blocks(pkg):
    """The create-time probe compares the blocked streams with row-by-row substitution; the self-test entry
    exposes the same replay, so an ill-conditioned diagonal block must show a large discrepancy here."""
    from oracle import oracle_c as oc

    n = 200
    rng = np.random.default_rng(0)
    good = np.linalg.cholesky(np.diag(rng.uniform(0.01, 0.09, n)) + 1e-4 * np.ones((n, n)))
    bad = np.tril(rng.standard_normal((n, n)), -1) * 3.0 + np.diag(np.full(n, 1e-3))  # cond ~ 1e40: hopeless
    b = rng.standard_normal(n)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    chi2 = C.c_double()
    pkg._lib.check(pkg.lib().cf_selftest_pack_host(p(good), n, n, p(b), C.byref(chi2), None))
    assert chi2.value == pytest.approx(oc.solve_triangular(good, b), rel=1e-12)
    pkg._lib.check(pkg.lib().cf_selftest_pack_host(p(bad), n, n, p(b), C.byref(chi2), None))
    ref = oc.solve_triangular(bad, b)
    assert not np.isfinite(chi2.value) or abs(chi2.value - ref) > 1e-11 * abs(ref)


def test_packed_factor_rejects_bad_pivot(pkg):
    Lm = np.eye(20)
    Lm[7, 7] = 0.0
    chi2 = C.c_double()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = pkg.lib().cf_selftest_pack_host(p(Lm), 20, 20, p(np.ones(20)), C.byref(chi2), None)
    assert rc == -4


def test_ensemble_entry_points_validate_their_arguments(pkg):
    """cf_ens_*: argument errors are reported before anything is launched (so this needs no GPU)."""
    import ctypes as C

    lib, L = pkg.lib(), pkg._lib
    buf = (C.c_double * 64)()
    ids = (C.c_int64 * 8)()
    p = lambda a: C.cast(a, C.c_void_p)
    with pytest.raises(pkg.CosmofitError, match="even number"):
        L.check(lib.cf_ens_kde_prepare(p(buf), 7, 4, 2, 0, 0, p(buf), p(buf), None))
    with pytest.raises(pkg.CosmofitError, match="ndim"):
        L.check(lib.cf_ens_propose(0, p(buf), 8, 17, 2, 0, 0, p(ids), 4, 1, 2.0, 1e-5, None, None, p(buf), p(buf), None))
    with pytest.raises(pkg.CosmofitError, match="n_splits"):
        L.check(lib.cf_ens_propose(0, p(buf), 8, 4, 4, 0, 0, p(ids), 4, 1, 2.0, 1e-5, None, None, p(buf), p(buf), None))
    with pytest.raises(pkg.CosmofitError, match="split must be"):
        L.check(lib.cf_ens_propose(0, p(buf), 8, 4, 2, 2, 0, p(ids), 4, 1, 2.0, 1e-5, None, None, p(buf), p(buf), None))
    with pytest.raises(pkg.CosmofitError, match="kind"):
        L.check(lib.cf_ens_propose(3, p(buf), 8, 4, 2, 0, 0, p(ids), 4, 1, 2.0, 1e-5, None, None, p(buf), p(buf), None))
    with pytest.raises(pkg.CosmofitError, match="null"):
        L.check(lib.cf_ens_propose(2, p(buf), 8, 4, 2, 0, 0, p(ids), 4, 1, 2.0, 1e-5, None, None, p(buf), p(buf), None))  # KDE without its fit
    with pytest.raises(pkg.CosmofitError, match="null"):
        L.check(lib.cf_ens_accept(p(ids), p(ids), 4, 4, 1, p(buf), p(buf), p(buf), p(buf), p(buf), None, None))
    with pytest.raises(pkg.CosmofitError, match="shard range"):
        L.check(lib.cf_ens_active_set(5, 3, 1, 10, 4, p(ids), p(ids), None))
    assert lib.cf_ens_active_count(5, 4, 0, 0, 10) == -1 and lib.cf_ens_comp_count(5, 2, 2, 10) == -1


def test_split_counts_of_the_library_match_the_driver_and_the_tensor_statement(pkg):
    """cf_ens_active_count / cf_ens_comp_count (host functions of the library) == ensemble.active_count (Python ints) == a
    count over oracle.moves_torch.split_of, for two and three splits, fixed and re-drawn, at shard boundaries that cut groups."""
    import torch
    from oracle import moves_torch

    E, lib = pkg.ensemble, pkg.lib()
    for key in (0, E.stream_key(7, 3, 0, E._SPLIT_STREAM), E.stream_key(8, 1000003, 0, E._SPLIT_STREAM)):
        for S in (2, 3):
            for W in (6, 32, 50, 151):
                ids = torch.arange(W, dtype=torch.int64)
                sp = moves_torch.split_of(key, S, ids)
                for c in range((W + S - 1) // S):  # every split holds exactly one member of every whole group
                    grp = sp[S * c: S * c + S].tolist()
                    assert sorted(grp) == list(range(S))[: len(grp)] or len(grp) < S
                    assert tuple(grp) == E.split_perm(key, S, c)[: len(grp)]
                for s in range(S):
                    assert lib.cf_ens_comp_count(key, S, s, W) == int((sp != s).sum())
                    for start, stop in ((0, W), (1, W - 1), (2, 5), (W // 3, 2 * W // 3 + 1), (4, 4), (W - 1, W)):
                        want = int((sp[start:stop] == s).sum())
                        assert lib.cf_ens_active_count(key, S, s, start, stop) == want == E.active_count(key, S, s, start, stop)
