"""
``bao_theory(z, qty, params[, DM_interp])`` with the scripts' own signature at ARBITRARY redshifts: what the post-fit blocks hand
to ``plot_bao_predictions`` (bao/plot_predictions.py:24-45: z_smooth = linspace(0, max z, 200), one curve per quantity code).
SURVEY 8(b) lists ``bao_theory`` among the accessors a drop-in must keep.  Fixture: the curves the scripts' own functions return
(tests/golden/generate_golden.py::case_bao_plot_curves) for bao/desi.py (PCHIP D_H, fixed r_d), bao/desi_cc.py (exact D_H, free
r_d) and bao/desi_cmb_des5y.py (physical densities, fitted r_drag, F_AP, four-argument form).

The same fixture holds ``H_z(z, params)`` at the redshifts of ``plot_cc_predictions`` (ohd/plot_predictions.py:7-32) and far beyond.

CPU: the numpy oracle evaluated at the plot's redshifts.  GPU (-m gpu): ``cf_eval_bao_at`` / ``cf_eval_hz`` through the mirrors,
bar 1e-10.
"""
import dataclasses

import numpy as np
import pytest

from conftest import golden
from oracle import oracle_np as onp
from test_oracle_golden import _chol_of, lk_bao_desi, lk_bao_desi_cmb_des5y
from test_variants import lk_bao_desi_cc

RTOL = 1e-10
CASES = {"desi": "bao_desi", "desi_cc": "bao_desi_cc", "desi_cmb_des5y": "bao_desi_cmb_des5y"}


def _oracle(tag):
    g = golden(CASES[tag])
    return {"desi": lambda: lk_bao_desi(g), "desi_cc": lambda: lk_bao_desi_cc(g),
            "desi_cmb_des5y": lambda: lk_bao_desi_cmb_des5y(g, _chol_of(g))}[tag]()


@pytest.mark.parametrize("tag", sorted(CASES))
def test_oracle_reproduces_the_plotted_curves(tag):
    c = golden("bao_plot_curves")
    lk, z = _oracle(tag), c[tag + "_z"]
    for code, curve in zip(c[tag + "_codes"], c[tag + "_curves"]):
        at = dataclasses.replace(lk, bao_z=z, bao_qty=np.full(z.size, code, dtype=np.int32), bao_val=np.zeros(z.size),
                                 bao_inv_cov=np.zeros((z.size, z.size)))
        np.testing.assert_allclose(onp.bao_theory(at, c[tag + "_theta"]), curve, rtol=RTOL, atol=1e-300)


@pytest.mark.parametrize("tag", sorted(CASES))
def test_oracle_reproduces_the_plotted_hubble_curve(tag):
    c = golden("bao_plot_curves")
    np.testing.assert_allclose(onp.H_z(_oracle(tag), c[tag + "_hz_z"], c[tag + "_theta"]), c[tag + "_hz"], rtol=1e-13)


@pytest.fixture(scope="module")
def gpu(pkg):
    if pkg.lib().cf_device_count() < 1:
        pytest.fail("GPU tests need an MI355X; no HIP device visible (there is no fallback path)")
    return pkg


def _mirror(gpu, tag):
    g = golden(CASES[tag])
    bao = (g["bao_z"], g["bao_val"], g["bao_qty"], g["bao_inv_cov"])
    lkl = gpu.likelihoods
    if tag == "desi":
        return lkl.DesiBao(*bao, rd=float(g["rd"])), g
    if tag == "desi_cc":
        return lkl.DesiCc(*bao, g["cc_z"], g["cc_h"], g["cc_cov"]), g
    return lkl.DesiCmbDes5y(g["z_cmb"], g["z_hel"], g["obs"], None, *bao, chol=_chol_of(g)), g


@pytest.mark.gpu
@pytest.mark.parametrize("tag", sorted(CASES))
def test_gpu_bao_theory_with_the_scripts_signature(gpu, tag):
    c = golden("bao_plot_curves")
    lk, g = _mirror(gpu, tag)
    z, theta = c[tag + "_z"], c[tag + "_theta"]
    predictions = lambda zz, qty: lk.bao_theory(zz, qty, theta)  # the lambda of the post-fit block, bao/desi.py:204-211
    for code, curve in zip(c[tag + "_codes"], c[tag + "_curves"]):
        got = predictions(z, np.full_like(z, code, dtype=np.int32))  # bao/plot_predictions.py:39-41
        np.testing.assert_allclose(got, curve, rtol=RTOL, atol=1e-300)
    # four-argument form (DM_interp is implied by params), a scalar quantity code, and the data points themselves
    np.testing.assert_array_equal(lk.bao_theory(z[:70], 1, theta, None), lk.bao_theory(z[:70], np.ones(70, dtype=np.int32), theta))
    np.testing.assert_allclose(lk.bao_theory(g["bao_z"], g["bao_qty"], theta), lk.bao_theory(theta), rtol=1e-13)
    # H_z(z, params): the lambda of plot_cc_predictions (ohd/cc.py:95-96), also far beyond the data
    np.testing.assert_allclose(lk.H_z(c[tag + "_hz_z"], theta), c[tag + "_hz"], rtol=RTOL)
    with pytest.raises(gpu.CosmofitError):
        lk.bao_theory(z[:3], 7, theta)
    with pytest.raises(TypeError):
        lk.bao_theory(z, theta)
    lk.engine.close()
