"""Domain statistics and the Ghan-style forcing decomposition of the reference's post-processing chain, on the
device-resident output fields.

* ``domain_stats``: calc_standard_stats of analysis_scripts/NCL_extraction_package/misc_stats_library.ncl:405-458
  through ``rrtmg_hip_domain_stats`` (sort + fixed-order reductions on the GPU).
* ``decompose``: the combinations of analysis_scripts/RadDecomp_analysis_package/RadDecomp_functions.py:147-266
  (SW direct / indirect / semi-direct, LW indirect / semi-direct) from the statistics of four scenarios:
  BASE, BASE without aerosol-radiation interaction (``_nA``), ALT (emission source removed) and ALT ``_nA``.
  Errors combine in quadrature exactly as there (``error_type`` = "stddev", "SE", ...).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .engine import ERRLEN, _check, _ptr, load_library

STAT_NAMES = ["avg", "stddev", "min", "max", "median", "x25", "x75", "x05", "x95", "SE", "morans_i", "SE_corr", "N"]


def domain_stats(field, dims, mask=None, stream=None) -> dict:
    """field: 2-D torch CUDA tensor (nj_m, ni_m); dims: ims..jte (k entries ignored); mask: optional uint8 CUDA
    tensor of the same shape (nonzero = use).  Returns the 13 statistics of the reference as a dict of floats."""
    import torch
    lib = load_library()
    lib.rrtmg_hip_domain_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_int] * 8 + [C.c_void_p, C.c_void_p,
                                                                                             C.c_char_p]
    p, is_dev, size = _ptr(field)
    if not is_dev:
        raise ValueError("domain_stats needs a device-resident field")
    mp = None
    if mask is not None:
        if mask.dtype != torch.uint8 or not mask.is_cuda or not mask.is_contiguous() or mask.shape != field.shape:
            raise ValueError("mask must be a contiguous uint8 CUDA tensor of the field's shape")
        mp = mask.data_ptr()
    out = torch.zeros(13, dtype=torch.float64, device=field.device)
    s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
    err = C.create_string_buffer(ERRLEN)
    _check(lib.rrtmg_hip_domain_stats(p, mp, size, dims["ims"], dims["ime"], dims["jms"], dims["jme"], dims["its"],
                                      dims["ite"], dims["jts"], dims["jte"], out.data_ptr(), s, err), err)
    return dict(zip(STAT_NAMES, out.cpu().tolist()))


def morans_i_host(field2d, mask=None) -> float:
    """Moran's I the way calc_morans_i_2D forms it with the options of calc_standard_stats (neighbour, manhattan;
    misc_stats_library.ncl:196-370): flattened field, the eight displacements, weight 1 at Manhattan distance 1;
    masked cells are missing (their products drop out, their weights stay)."""
    x = np.asarray(field2d, dtype=np.float64)
    nj, ni = x.shape
    ok = np.ones(x.shape, bool) if mask is None else np.asarray(mask) != 0
    if not ok.any():
        return 0.0
    vals = x[ok]
    sd = float(np.std(vals, ddof=1)) if vals.size > 1 else 0.0
    if sd == 0.0:
        return 0.0
    d = np.where(ok, x - vals.mean(), 0.0).ravel()
    okf = ok.ravel()
    N = ni * nj
    n = np.arange(N)
    ii, jj = n % ni, n // ni
    au = w = 0.0
    for m in (1, ni - 1, ni, ni + 1, N - ni - 1, N - ni, N - ni + 1, N - 1):
        if m == 0:
            continue
        n1 = (n + m) % N
        near = (np.abs(ii - ii[n1]) + np.abs(jj - jj[n1])) == 1
        w += float(near.sum())
        au += float((d * d[n1])[near & okf & okf[n1]].sum())
    return au / (w * sd * sd) if w > 0 else 0.0


def domain_stats_host(values, field2d=None, mask=None) -> dict:
    """The same statistics with numpy (for tests and for host-resident data).  Moran's I needs the 2-D field
    (field2d, optional mask); without it the two slots carry the reference's -9999 marker."""
    x = np.sort(np.asarray(values, dtype=np.float64).ravel())
    n = x.size
    if n == 0:
        return dict(zip(STAT_NAMES, [0.0] * 13))
    sd = float(np.std(x, ddof=1)) if n > 1 else 0.0

    def pick(p):
        v = 0.01 * p * (n - 1)
        return x[min(max(int(v + 0.5), 0), n - 1)]

    se = sd / np.sqrt(n)
    mi = morans_i_host(field2d, mask) if field2d is not None else -9999.0
    return dict(zip(STAT_NAMES, [float(x.mean()), sd, float(x[0]), float(x[-1]), pick(50.), pick(25.), pick(75.),
                                 pick(5.), pick(95.), se, mi, se * mi if field2d is not None else -9999.0, float(n)]))


def _q(*errs):
    return float(np.sqrt(sum(e * e for e in errs)))


def decompose(base: dict, base_na: dict, alt: dict, alt_na: dict, error_type: str = "SE",
              lw_indirect_as_documented: bool = False) -> dict:
    """Each argument maps a field name (SWUPT, SWUPTCLN, LWUPT, LWUPTC) to its statistics dict (``domain_stats``).
    Returns {effect: (value, error)} for Delta_S, Delta_L, SW_DIRECT, SW_INDIRECT, SW_SEMIDIRECT, LW_INDIRECT,
    LW_SEMIDIRECT exactly as RadDecomp_functions.py:118-266 computes them (same operands, same order of operations,
    errors in quadrature) -- pinned on the reference itself: tests/golden/raddecomp.npz, tools/make_golden_raddecomp.py.

    LW_INDIRECT: the reference's code reads LWUPTC_nA where its comment says LWUPT_nA (:224-227), so its value is
    identically zero and its error counts the two LWUPTC_nA errors twice.  The default reproduces the code; with
    ``lw_indirect_as_documented=True`` the formula of the comment (:218-221) is used instead,
    (LWUPTC - LWUPT)_BASE_nA - (LWUPTC - LWUPT)_ALT_nA."""
    a, e = "avg", error_type
    out = {}
    out["Delta_S"] = (alt["SWUPT"][a] - base["SWUPT"][a], _q(alt["SWUPT"][e], base["SWUPT"][e]))
    out["Delta_L"] = (alt["LWUPT"][a] - base["LWUPT"][a], _q(alt["LWUPT"][e], base["LWUPT"][e]))
    # SW direct = (SWUPTCLN - SWUPT)_BASE - (SWUPTCLN - SWUPT)_ALT
    out["SW_DIRECT"] = (base["SWUPTCLN"][a] - base["SWUPT"][a] - alt["SWUPTCLN"][a] + alt["SWUPT"][a],
                        _q(base["SWUPTCLN"][e], base["SWUPT"][e], alt["SWUPTCLN"][e], alt["SWUPT"][e]))
    # SW indirect = SWUPTCLN_ALT_nA - SWUPTCLN_BASE_nA
    out["SW_INDIRECT"] = (alt_na["SWUPTCLN"][a] - base_na["SWUPTCLN"][a], _q(alt_na["SWUPTCLN"][e], base_na["SWUPTCLN"][e]))
    # SW semi-direct = SWUPTCLN_ALT - SWUPTCLN_BASE - SWUPT_ALT_nA + SWUPT_BASE_nA
    out["SW_SEMIDIRECT"] = (alt["SWUPTCLN"][a] - base["SWUPTCLN"][a] - alt_na["SWUPT"][a] + base_na["SWUPT"][a],
                            _q(alt["SWUPTCLN"][e], base["SWUPTCLN"][e], alt_na["SWUPT"][e], base_na["SWUPT"][e]))
    # LW indirect, as coded: every operand is LWUPTC_nA (:224-227)
    lwup_b, lwup_a = (base_na["LWUPT"], alt_na["LWUPT"]) if lw_indirect_as_documented else (base_na["LWUPTC"], alt_na["LWUPTC"])
    out["LW_INDIRECT"] = (base_na["LWUPTC"][a] - lwup_b[a] - alt_na["LWUPTC"][a] + lwup_a[a],
                          _q(base_na["LWUPTC"][e], lwup_b[e], alt_na["LWUPTC"][e], lwup_a[e]))
    # LW semi-direct = (LWUPTC - LWUPT)_BASE - (LWUPTC - LWUPT)_ALT - LW indirect (written out, :236-266)
    out["LW_SEMIDIRECT"] = (base["LWUPTC"][a] - base["LWUPT"][a] - alt["LWUPTC"][a] + alt["LWUPT"][a]
                            - base_na["LWUPTC"][a] + base_na["LWUPT"][a] + alt_na["LWUPTC"][a] - alt_na["LWUPT"][a],
                            _q(base["LWUPTC"][e], base["LWUPT"][e], alt["LWUPTC"][e], alt["LWUPT"][e],
                               base_na["LWUPTC"][e], base_na["LWUPT"][e], alt_na["LWUPTC"][e], alt_na["LWUPT"][e]))
    return out
