"""
cosmology-model-fit_amd — MI355X (gfx950) batched log-likelihood engine for the emcee / nautilus
hot path of franciscotln/cosmology-model-fit.  See DESIGN.md / INTEGRATION.md at the repo root.

The directory name contains a hyphen; import it with
``importlib.import_module("cosmology-model-fit_amd")``.
"""
from . import _lib
from ._lib import CosmofitError, build, lib
from ._lib import (CF_FDE_CPL, CF_FDE_LCDM, CF_FDE_THAWING, CF_FDE_WCDM, CF_OUT_CHI2, CF_OUT_LOGL, CF_OUT_LOGP,
                   CF_SOLVE_AUTO, CF_SOLVE_BLOCKED_TRSM, CF_SOLVE_INVERSE_GEMM)
from .engine import C_KM_S, LikelihoodEngine, Param
from . import cmb_data, interpolator, laplace, likelihoods, scripts, solve_triangular, sn_pantheon, synthetic



def __getattr__(name):
    # `ensemble` needs torch; import it lazily so that ctypes-only users do not pay for it
    if name == "ensemble":
        import importlib
        return importlib.import_module(__name__ + ".ensemble")
    raise AttributeError(name)


__all__ = ["LikelihoodEngine", "Param", "CosmofitError", "build", "lib", "interpolator", "solve_triangular",
           "sn_pantheon", "C_KM_S"]
