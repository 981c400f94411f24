"""Deterministic synthetic inputs for the RRTMG double-call path.

Two generators, both seeded with ``numpy.random.default_rng``:

* k-distribution coefficient files in the *record layout of WRF's own*
  ``RRTMG_SW_DATA`` / ``RRTMG_LW_DATA`` (Fortran unformatted sequential, one
  record per band).  The field order per record follows the READ statements
  of the reference's ``sw_kgbNN`` / ``lw_kgbNN`` routines
  (module_ra_rrtmg_sw.F:11379-12366, module_ra_rrtmg_lw.F:13021-14387).  The
  real data files are not shipped with the reference, so the numbers written
  here are synthetic; a real WRF ``run/`` file is read by the same loader.
* WRF-layout atmospheres: 3-D ``(i,k,j)`` state with i fastest, i.e. numpy
  arrays of shape ``(nj, nk, ni)`` in C order.

Nothing in this module touches the GPU or the oracle.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field

import numpy as np

# --------------------------------------------------------------------------
# Record schema of RRTMG_SW_DATA: (band, [(name, shape)...]) in READ order.
# shape () = real scalar, "i" = default INTEGER scalar; arrays are Fortran
# (column-major) order on disk.  13:59 pressure dims are written as 47.
# --------------------------------------------------------------------------
NO = 16  # original g-points per band in the files

SW_RECORDS = [
    (16, [("rayl", ()), ("strrat1", ()), ("layreffr", "i"), ("kao", (9, 5, 13, NO)),
          ("kbo", (5, 47, NO)), ("selfrefo", (10, NO)), ("forrefo", (3, NO)), ("sfluxrefo", (NO,))]),
    (17, [("rayl", ()), ("strrat", ()), ("layreffr", "i"), ("kao", (9, 5, 13, NO)),
          ("kbo", (5, 5, 47, NO)), ("selfrefo", (10, NO)), ("forrefo", (4, NO)), ("sfluxrefo", (NO, 5))]),
    (18, [("rayl", ()), ("strrat", ()), ("layreffr", "i"), ("kao", (9, 5, 13, NO)),
          ("kbo", (5, 47, NO)), ("selfrefo", (10, NO)), ("forrefo", (3, NO)), ("sfluxrefo", (NO, 9))]),
    (19, [("rayl", ()), ("strrat", ()), ("layreffr", "i"), ("kao", (9, 5, 13, NO)),
          ("kbo", (5, 47, NO)), ("selfrefo", (10, NO)), ("forrefo", (3, NO)), ("sfluxrefo", (NO, 9))]),
    (20, [("rayl", ()), ("layreffr", "i"), ("absch4o", (NO,)), ("kao", (5, 13, NO)),
          ("kbo", (5, 47, NO)), ("selfrefo", (10, NO)), ("forrefo", (4, NO)), ("sfluxrefo", (NO,))]),
    (21, [("rayl", ()), ("strrat", ()), ("layreffr", "i"), ("kao", (9, 5, 13, NO)),
          ("kbo", (5, 5, 47, NO)), ("selfrefo", (10, NO)), ("forrefo", (4, NO)), ("sfluxrefo", (NO, 9))]),
    (22, [("rayl", ()), ("strrat", ()), ("layreffr", "i"), ("kao", (9, 5, 13, NO)),
          ("kbo", (5, 47, NO)), ("selfrefo", (10, NO)), ("forrefo", (3, NO)), ("sfluxrefo", (NO, 9))]),
    (23, [("raylo", (NO,)), ("givfac", ()), ("layreffr", "i"), ("kao", (5, 13, NO)),
          ("selfrefo", (10, NO)), ("forrefo", (3, NO)), ("sfluxrefo", (NO,))]),
    (24, [("raylao", (NO, 9)), ("raylbo", (NO,)), ("strrat", ()), ("layreffr", "i"),
          ("abso3ao", (NO,)), ("abso3bo", (NO,)), ("kao", (9, 5, 13, NO)), ("kbo", (5, 47, NO)),
          ("selfrefo", (10, NO)), ("forrefo", (3, NO)), ("sfluxrefo", (NO, 9))]),
    (25, [("raylo", (NO,)), ("layreffr", "i"), ("abso3ao", (NO,)), ("abso3bo", (NO,)),
          ("kao", (5, 13, NO)), ("sfluxrefo", (NO,))]),
    (26, [("raylo", (NO,)), ("sfluxrefo", (NO,))]),
    (27, [("raylo", (NO,)), ("scalekur", ()), ("layreffr", "i"), ("kao", (5, 13, NO)),
          ("kbo", (5, 47, NO)), ("sfluxrefo", (NO,))]),
    (28, [("rayl", ()), ("strrat", ()), ("layreffr", "i"), ("kao", (9, 5, 13, NO)),
          ("kbo", (5, 5, 47, NO)), ("sfluxrefo", (NO, 5))]),
    (29, [("rayl", ()), ("layreffr", "i"), ("absh2oo", (NO,)), ("absco2o", (NO,)),
          ("kao", (5, 13, NO)), ("kbo", (5, 47, NO)), ("selfrefo", (10, NO)), ("forrefo", (4, NO)),
          ("sfluxrefo", (NO,))]),
]

# Scalars with the magnitudes of the shipped WRF data (used only to put the
# synthetic optical depths in a realistic range; not reference content).
_SW_SCAL = {
    16: dict(rayl=2.91e-10, strrat1=252.131, layreffr=18),
    17: dict(rayl=6.97e-10, strrat=0.364641, layreffr=30),
    18: dict(rayl=1.39e-09, strrat=38.9589, layreffr=6),
    19: dict(rayl=2.29e-09, strrat=5.49281, layreffr=3),
    20: dict(rayl=4.12e-09, layreffr=3),
    21: dict(rayl=9.41e-09, strrat=0.0045321, layreffr=8),
    22: dict(rayl=1.89e-08, strrat=0.022708, layreffr=2),
    23: dict(givfac=1.029, layreffr=6),
    24: dict(strrat=0.124692, layreffr=1),
    25: dict(layreffr=2),
    26: dict(),
    27: dict(scalekur=50.15 / 48.37, layreffr=32),
    28: dict(rayl=2.02e-05, strrat=6.67029e-07, layreffr=58),
    29: dict(rayl=9.41e-11, layreffr=49),
}
# typical key-species column amounts (1e20 molec/cm2 per layer) below / above
# the tropopause, used to scale k so that layer optical depths span 1e-5..3
_SW_AMOUNT = {16: (10., 1e-2), 17: (10., 1.0), 18: (10., 1e-2), 19: (20., 2.0), 20: (10., 3e-2),
              21: (10., 3e-2), 22: (40., 2e3), 23: (10., 1.0), 24: (150., 1e3), 25: (10., 1.0),
              26: (1., 1.), 27: (1e-3, 5e-3), 28: (2e-3, 5e-3), 29: (10., 2.0)}
# per-band per-g Rayleigh magnitudes for the bands that carry raylo arrays
_SW_RAYL_G = {23: 4.0e-08, 24: 1.3e-07, 25: 5.0e-07, 26: 2.0e-06, 27: 8.0e-06}
# share of the solar constant per band (sums to 1)
_SW_SOLAR = np.array([12.1, 20.4, 23.7, 22.4, 55.6, 102.9, 24.3, 345.7, 218.2,
                      347.2, 129.5, 50.2, 3.1, 12.9])
_SW_SOLAR = _SW_SOLAR / _SW_SOLAR.sum()


def _ktable(rng, shape, amount):
    """Positive k values, increasing with the (last) g index, log-uniform so
    that amount*k spans roughly 1e-5..3."""
    ng = shape[-1]
    lo, hi = np.log10(1e-5 / amount), np.log10(3.0 / amount)
    base = lo + (hi - lo) * (np.arange(ng) + 0.5) / ng
    noise = rng.uniform(-0.35, 0.35, size=shape)
    return 10.0 ** (base + noise)


def make_sw_tables(seed: int = 20260001) -> dict:
    """Synthetic contents for every field of RRTMG_SW_DATA, keyed
    ``{band: {name: ndarray|float|int}}``; arrays carry their Fortran shape."""
    rng = np.random.default_rng(seed)
    out = {}
    for ib, (band, fields) in enumerate(SW_RECORDS):
        rec = {}
        lo_amt, hi_amt = _SW_AMOUNT[band]
        for name, shape in fields:
            if shape == "i" or shape == ():
                rec[name] = _SW_SCAL[band][name]
            elif name == "kao":
                rec[name] = _ktable(rng, shape, lo_amt)
            elif name == "kbo":
                rec[name] = _ktable(rng, shape, hi_amt)
            elif name == "selfrefo":
                rec[name] = _ktable(rng, shape, 3e3)
            elif name == "forrefo":
                rec[name] = _ktable(rng, shape, 3e2)
            elif name == "sfluxrefo":
                w = rng.uniform(0.5, 1.5, size=shape)
                w = w / w.sum(axis=0, keepdims=True)
                rec[name] = 1368.22 * _SW_SOLAR[ib] * w
            elif name in ("raylo", "raylbo", "raylao"):
                rec[name] = _SW_RAYL_G[band] * rng.uniform(0.8, 1.25, size=shape)
            elif name in ("absch4o",):
                rec[name] = _ktable(rng, shape, 1e-2)
            elif name in ("abso3ao", "abso3bo"):
                rec[name] = _ktable(rng, shape, 3e-3)
            elif name in ("absh2oo",):
                rec[name] = _ktable(rng, shape, 1.0)
            elif name in ("absco2o",):
                rec[name] = _ktable(rng, shape, 3.0)
            else:  # pragma: no cover
                raise KeyError(name)
        out[band] = rec
    return out


def _pack_record(fields, rec, real_dtype):
    buf = bytearray()
    for name, shape in fields:
        v = rec[name]
        if shape == "i":
            buf += struct.pack("<i", int(v))
        elif shape == ():
            buf += np.asarray(v, dtype=real_dtype).tobytes()
        else:
            a = np.asarray(v, dtype=real_dtype)
            assert a.shape == tuple(shape), (name, a.shape, shape)
            buf += np.asfortranarray(a).tobytes(order="F")
    return bytes(buf)


def write_fortran_records(path, records, tables, real_bytes: int = 8):
    """Write ``tables`` as a Fortran unformatted sequential file (4-byte
    little-endian record markers, as gfortran/flang/ifort produce)."""
    real_dtype = {4: "<f4", 8: "<f8"}[real_bytes]
    with open(path, "wb") as f:
        for band, fields in records:
            payload = _pack_record(fields, tables[band], real_dtype)
            mark = struct.pack("<i", len(payload))
            f.write(mark)
            f.write(payload)
            f.write(mark)


def write_sw_data(path, seed: int = 20260001, real_bytes: int = 8) -> dict:
    tables = make_sw_tables(seed)
    write_fortran_records(path, SW_RECORDS, tables, real_bytes)
    return tables


# --------------------------------------------------------------------------
# Synthetic WRF-layout atmosphere (SURVEY.md §8d "Atmosphere per column")
# --------------------------------------------------------------------------
R_D = 287.0
CP = 7.0 * R_D / 2.0  # WRF module_model_constants: cp = 7.*r_d/2.
G = 9.81


@dataclass
class WrfState:
    """A tile of WRF state in WRF memory order.  3-D arrays are (nj, nk, ni)
    C-order == Fortran (ni, nk, nj); 2-D arrays are (nj, ni)."""
    ni: int
    nj: int
    nk: int           # model layers K = kte-kts+1 ; 3-D arrays have nk+1 k-slots (kme)
    p_top: float      # Pa
    a3: dict = field(default_factory=dict)   # name -> (nj, nk+1, ni) float64
    a2: dict = field(default_factory=dict)   # name -> (nj, ni) float64

    @property
    def ncol(self):
        return self.ni * self.nj


def make_wrf_state(ni, nj, nk, seed=20260002, night_fraction=0.0, aerosol=True,
                   p_top_hpa=50.0) -> WrfState:
    rng = np.random.default_rng(seed)
    kme = nk + 1
    st = WrfState(ni=ni, nj=nj, nk=nk, p_top=p_top_hpa * 100.0)
    shp2 = (nj, ni)
    psfc = rng.uniform(950.0, 1030.0, size=shp2) * 100.0            # Pa, fractional Pa kept
    eta = np.linspace(1.0, 0.0, nk + 1) ** 1.25                       # eta-like full levels
    p8w = st.p_top + eta[None, :, None] * (psfc[:, None, :] - st.p_top)   # (nj, nk+1, ni)
    p3d = 0.5 * (p8w[:, :-1, :] + p8w[:, 1:, :])                      # (nj, nk, ni)
    phpa = p3d / 100.0
    t3d = 288.0 - 65.0 * np.clip(1.0 - phpa / 1000.0, 0.0, None) ** 0.9 \
        + rng.normal(0.0, 2.0, size=p3d.shape)
    t3d = np.maximum(t3d, 205.0)
    t8w = np.empty_like(p8w)
    t8w[:, 1:-1, :] = 0.5 * (t3d[:, :-1, :] + t3d[:, 1:, :])
    t8w[:, 0, :] = t3d[:, 0, :]
    t8w[:, -1, :] = t3d[:, -1, :]
    q0 = rng.uniform(0.004, 0.018, size=shp2)
    qv = np.maximum(3.0e-6, q0[:, None, :] * (phpa / 1000.0) ** 3)
    tsk = t3d[:, 0, :] + rng.uniform(-2.0, 4.0, size=shp2)
    albedo = rng.uniform(0.05, 0.4, size=shp2)
    emiss = rng.uniform(0.9, 1.0, size=shp2)
    coszen = rng.uniform(0.05, 1.0, size=shp2)
    if night_fraction > 0.0:
        night = rng.uniform(size=shp2) < night_fraction
        coszen = np.where(night, -rng.uniform(0.0, 0.5, size=shp2), coszen)
    rho = p3d / (R_D * t3d * (1.0 + 0.608 * qv))
    dz = (p8w[:, :-1, :] - p8w[:, 1:, :]) / (rho * G)
    pi3d = (p3d / 1.0e5) ** (R_D / CP)

    def pad(a):  # (nj,nk,ni) -> (nj,kme,ni) with an unused top slot, as WRF's kme
        out = np.zeros((nj, kme, ni))
        out[:, :nk, :] = a
        return out

    st.a3.update(p8w=p8w, t8w=t8w, p3d=pad(p3d), t3d=pad(t3d), qv3d=pad(qv), rho3d=pad(rho),
                 dz8w=pad(dz), pi3d=pad(pi3d))
    st.a3["pi3d"][:, nk, :] = 1.0
    st.a2.update(tsk=tsk, albedo=albedo, emiss=emiss, xcoszen=coszen,
                 xland=np.ones(shp2), xice=np.zeros(shp2), snow=np.zeros(shp2))

    # aerosol optics at 300/400/600/999 nm (WRF-Chem compact form) + 16 LW bands
    z = np.cumsum(dz, axis=1) - 0.5 * dz
    if aerosol:
        aod = 10.0 ** rng.uniform(np.log10(0.02), np.log10(1.5), size=shp2)
        w = np.exp(-z / 2000.0) * dz
        w = w / w.sum(axis=1, keepdims=True)
        tau400 = aod[:, None, :] * w
        ang = rng.uniform(0.3, 1.8, size=shp2)[:, None, :]
        for lam, nm in ((300.0, "300"), (400.0, "400"), (600.0, "600"), (999.0, "999")):
            st.a3["tauaer" + nm] = pad(tau400 * (lam / 400.0) ** (-ang))
        w400 = rng.uniform(0.75, 0.99, size=p3d.shape)
        w600 = rng.uniform(0.75, 0.99, size=p3d.shape)
        g400 = rng.uniform(0.55, 0.8, size=p3d.shape)
        g600 = rng.uniform(0.55, 0.8, size=p3d.shape)
        st.a3.update(waer300=pad(w400), waer400=pad(w400), waer600=pad(w600), waer999=pad(w600),
                     gaer300=pad(g400), gaer400=pad(g400), gaer600=pad(g600), gaer999=pad(g600))
        for b in range(1, 17):
            st.a3[f"tauaerlw{b}"] = pad(tau400 * rng.uniform(0.02, 0.15, size=shp2)[:, None, :])
    else:
        for nm in ("300", "400", "600", "999"):
            st.a3["tauaer" + nm] = np.zeros((nj, kme, ni))
            st.a3["waer" + nm] = np.ones((nj, kme, ni))
            st.a3["gaer" + nm] = np.zeros((nj, kme, ni))
        for b in range(1, 17):
            st.a3[f"tauaerlw{b}"] = np.zeros((nj, kme, ni))
    return st


def make_wrf_rows(ni, nj, nk, j0, j1, seed=20260002, aerosol=True, p_top_hpa=50.0) -> WrfState:
    """Rows j0 <= j < j1 (0-based) of ONE synthetic ni x nj domain: every row is generated from its own seed, so any
    partition of the domain into j-slabs (one per rank) yields the same domain -- what a strong-scaling run needs
    (each rank builds only its slab).  Night columns are set by the caller (bench.py) from a per-row mask."""
    if not (0 <= j0 <= j1 <= nj):
        raise ValueError("rows outside the domain")
    rows = [make_wrf_state(ni, 1, nk, seed=seed + 7919 * j, aerosol=aerosol, p_top_hpa=p_top_hpa) for j in range(j0, j1)]
    st = WrfState(ni=ni, nj=j1 - j0, nk=nk, p_top=p_top_hpa * 100.0)
    if rows:
        for n in rows[0].a3:
            st.a3[n] = np.concatenate([r.a3[n] for r in rows], axis=0)
        for n in rows[0].a2:
            st.a2[n] = np.concatenate([r.a2[n] for r in rows], axis=0)
    return st


# --------------------------------------------------------------------------
# RRTMG_LW_DATA: record schema in READ order (lw_kgb01..16, LW:13021-14387)
# --------------------------------------------------------------------------
_KA9, _KA1 = (9, 5, 13, NO), (5, 13, NO)
_KB5, _KB1 = (5, 5, 47, NO), (5, 47, NO)
_SF = [("selfrefo", (10, NO)), ("forrefo", (4, NO))]
LW_RECORDS = [
    (1, [("fracrefao", (NO,)), ("fracrefbo", (NO,)), ("kao", _KA1), ("kbo", _KB1),
         ("kao_mn2", (19, NO)), ("kbo_mn2", (19, NO))] + _SF),
    (2, [("fracrefao", (NO,)), ("fracrefbo", (NO,)), ("kao", _KA1), ("kbo", _KB1)] + _SF),
    (3, [("fracrefao", (NO, 9)), ("fracrefbo", (NO, 5)), ("kao", _KA9), ("kbo", _KB5),
         ("kao_mn2o", (9, 19, NO)), ("kbo_mn2o", (5, 19, NO))] + _SF),
    (4, [("fracrefao", (NO, 9)), ("fracrefbo", (NO, 5)), ("kao", _KA9), ("kbo", _KB5)] + _SF),
    (5, [("fracrefao", (NO, 9)), ("fracrefbo", (NO, 5)), ("kao", _KA9), ("kbo", _KB5),
         ("kao_mo3", (9, 19, NO)), ("ccl4o", (NO,))] + _SF),
    (6, [("fracrefao", (NO,)), ("kao", _KA1), ("kao_mco2", (19, NO)), ("cfc11adjo", (NO,)), ("cfc12o", (NO,))] + _SF),
    (7, [("fracrefao", (NO, 9)), ("fracrefbo", (NO,)), ("kao", _KA9), ("kbo", _KB1),
         ("kao_mco2", (9, 19, NO)), ("kbo_mco2", (19, NO))] + _SF),
    (8, [("fracrefao", (NO,)), ("fracrefbo", (NO,)), ("kao", _KA1), ("kbo", _KB1),
         ("kao_mco2", (19, NO)), ("kbo_mco2", (19, NO)), ("kao_mn2o", (19, NO)), ("kbo_mn2o", (19, NO)),
         ("kao_mo3", (19, NO)), ("cfc12o", (NO,)), ("cfc22adjo", (NO,))] + _SF),
    (9, [("fracrefao", (NO, 9)), ("fracrefbo", (NO,)), ("kao", _KA9), ("kbo", _KB1),
         ("kao_mn2o", (9, 19, NO)), ("kbo_mn2o", (19, NO))] + _SF),
    (10, [("fracrefao", (NO,)), ("fracrefbo", (NO,)), ("kao", _KA1), ("kbo", _KB1)] + _SF),
    (11, [("fracrefao", (NO,)), ("fracrefbo", (NO,)), ("kao", _KA1), ("kbo", _KB1),
          ("kao_mo2", (19, NO)), ("kbo_mo2", (19, NO))] + _SF),
    (12, [("fracrefao", (NO, 9)), ("kao", _KA9)] + _SF),
    (13, [("fracrefao", (NO, 9)), ("fracrefbo", (NO,)), ("kao", _KA9), ("kao_mco2", (9, 19, NO)),
          ("kao_mco", (9, 19, NO)), ("kbo_mo3", (19, NO))] + _SF),
    (14, [("fracrefao", (NO,)), ("fracrefbo", (NO,)), ("kao", _KA1), ("kbo", _KB1)] + _SF),
    (15, [("fracrefao", (NO, 9)), ("kao", _KA9), ("kao_mn2", (9, 19, NO))] + _SF),
    (16, [("fracrefao", (NO, 9)), ("fracrefbo", (NO,)), ("kao", _KA9), ("kbo", _KB1)] + _SF),
]
# typical major-species amounts (1e20 molec/cm2 per layer) below / above the tropopause
_LW_AMOUNT = {1: (10., 0.03), 2: (10., 0.03), 3: (10., 0.5), 4: (10., 0.5), 5: (10., 0.5), 6: (10., 1.),
              7: (10., 5e-3), 8: (10., 5e-3), 9: (10., 1e-2), 10: (10., 0.03), 11: (10., 0.03), 12: (10., 1.),
              13: (10., 1.), 14: (2., 2.), 15: (1e-2, 1.), 16: (10., 1e-2)}
_LW_MINOR = {"kao_mn2": 3e4, "kbo_mn2": 3e4, "kao_mn2o": 2e-3, "kbo_mn2o": 2e-3, "kao_mo3": 1e-3, "kbo_mo3": 5e-3,
             "kao_mco2": 2.0, "kbo_mco2": 2.0, "kao_mo2": 3e3, "kbo_mo2": 3e2, "kao_mco": 1e-3,
             "ccl4o": 5e-7, "cfc11adjo": 1.3e-6, "cfc12o": 2.8e-6, "cfc22adjo": 9e-7}


def make_lw_tables(seed: int = 20260003) -> dict:
    rng = np.random.default_rng(seed)
    out = {}
    for band, fields in LW_RECORDS:
        rec = {}
        lo_amt, hi_amt = _LW_AMOUNT[band]
        for name, shape in fields:
            if name in ("fracrefao", "fracrefbo"):
                w = rng.uniform(0.3, 1.7, size=shape) * np.exp(-0.12 * np.arange(NO)).reshape((NO,) + (1,) * (len(shape) - 1))
                rec[name] = w / w.sum(axis=0, keepdims=True)
            elif name == "kao":
                rec[name] = _ktable(rng, shape, lo_amt)
            elif name == "kbo":
                rec[name] = _ktable(rng, shape, hi_amt)
            elif name == "selfrefo":
                rec[name] = _ktable(rng, shape, 3e2)      # selffac carries colh2o * water * scalefac
            elif name == "forrefo":
                rec[name] = _ktable(rng, shape, 3e2)
            else:
                rec[name] = 0.1 * _ktable(rng, shape, _LW_MINOR[name])
        out[band] = rec
    return out


def write_lw_data(path, seed: int = 20260003, real_bytes: int = 8) -> dict:
    tables = make_lw_tables(seed)
    write_fortran_records(path, LW_RECORDS, tables, real_bytes)
    return tables
