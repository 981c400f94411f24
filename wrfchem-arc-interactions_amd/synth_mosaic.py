"""Synthetic MOSAIC sectional aerosol for the optical-averaging stage (BASELINE configs[2], [3]: 4 and 8 bins).

Species per bin follow the reference's registry (Registry/registry.chem:2325-2447, package lists :3823-3826):
so4, no3, cl, nh4, na, oin, oc, bc, hysw, water (ug/kg-dryair) and num (#/kg-dryair).  The tracer array is laid
out like WRF-Chem's ``chem(ims:ime, kms:kme, jms:jme, num_chem)``: numpy shape (num_chem, nj, nk_mem, ni).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .engine_optics import MOSAIC_DHI_UM, MOSAIC_DLO_UM, SPECIES

REGISTRY_ORDER = ["so4", "no3", "cl", "nh4", "na", "oin", "oc", "bc", "hysw", "water", "num"]
DENSITY = dict(so4=1.8, no3=1.8, cl=2.2, nh4=1.8, na=2.2, oin=2.6, oc=1.0, bc=1.7)
N_GAS = 5   # a few leading gas-phase tracers so that aerosol indices do not start at 1


@dataclass
class MosaicState:
    nbin: int
    chem: np.ndarray      # (num_chem, nj, nk_mem, ni) float64
    alt: np.ndarray       # (nj, nk_mem, ni) inverse dry-air density, m3/kg
    dz8w: np.ndarray      # (nj, nk_mem, ni) m
    idx: np.ndarray       # [nbin][10] 1-based species numbers in engine_optics.SPECIES order
    names: list


def section_bounds(nbin, dlo=MOSAIC_DLO_UM, dhi=MOSAIC_DHI_UM):
    edges = dlo * (dhi / dlo) ** (np.arange(nbin + 1) / nbin)
    edges[-1] = dhi
    return edges[:-1], edges[1:]


def make_mosaic_state(st, nbin=4, seed=20260005, empty_fraction=0.02) -> MosaicState:
    """st: synth.WrfState (uses rho3d, dz8w).  Mass loading decays with height (scale 2 km); a fine and a coarse
    log-normal mode are split over the sections; number follows from the section's mass and a mean diameter that
    sometimes lies outside the section (to exercise the clipping rule); water from a per-cell growth factor."""
    rng = np.random.default_rng(seed)
    nj, kme, ni = st.a3["rho3d"].shape
    nk = st.nk
    rho = st.a3["rho3d"][:, :nk, :]
    dz = st.a3["dz8w"][:, :nk, :]
    z = np.cumsum(dz, axis=1) - 0.5 * dz
    shp2, shp3 = (nj, ni), (nj, nk, ni)
    lo, hi = section_bounds(nbin)
    dmid = np.sqrt(lo * hi)
    # column loadings (ug/m3 at the surface) of a fine and a coarse mode
    fine0 = 10.0 ** rng.uniform(np.log10(2.0), np.log10(60.0), size=shp2)
    coarse0 = 10.0 ** rng.uniform(np.log10(1.0), np.log10(40.0), size=shp2)
    prof = np.exp(-z / 2000.0)

    def mode_split(dg_um, sg):
        # fraction of a log-normal volume distribution falling into each section
        from math import erf, log, sqrt
        cdf = lambda d: 0.5 * (1.0 + erf(log(d / dg_um) / (sqrt(2.0) * log(sg))))
        f = np.array([cdf(h) - cdf(l) for l, h in zip(lo, hi)])
        return f / f.sum()

    f_fine, f_coarse = mode_split(0.30, 1.8), mode_split(3.0, 2.0)
    names = [f"gas{i + 1}" for i in range(N_GAS)]
    for b in range(nbin):
        names += [f"{s}_a{b + 1:02d}" for s in REGISTRY_ORDER]
    num_chem = len(names)
    chem = np.zeros((num_chem, nj, kme, ni))
    chem[:N_GAS, :, :nk, :] = rng.uniform(0.0, 1.0, size=(N_GAS,) + shp3)
    alt = np.zeros((nj, kme, ni))
    alt[:, :nk, :] = 1.0 / rho
    alt[:, nk:, :] = 1.0
    # composition of the two modes per column (mass fractions of the dry species)
    comp_f = rng.dirichlet([4.0, 2.0, 0.3, 2.0, 0.3, 0.8, 3.0, 0.6], size=shp2)    # so4 no3 cl nh4 na oin oc bc
    comp_c = rng.dirichlet([0.5, 0.5, 2.0, 0.2, 2.0, 6.0, 0.5, 0.05], size=shp2)
    growth = rng.uniform(0.0, 2.5, size=shp3)          # water mass / soluble dry mass
    idx = np.zeros((nbin, len(SPECIES)), dtype=np.int64)
    for b in range(nbin):
        base = N_GAS + b * len(REGISTRY_ORDER)
        for s, sp in enumerate(SPECIES):
            idx[b, s] = base + REGISTRY_ORDER.index(sp) + 1
        m_f = fine0[:, None, :] * prof * f_fine[b]       # ug/m3
        m_c = coarse0[:, None, :] * prof * f_coarse[b]
        dry = np.zeros(shp3)
        vol = np.zeros(shp3)                             # 1e-6 cm3 per m3 of air
        soluble = np.zeros(shp3)
        for s, sp in enumerate(["so4", "no3", "cl", "nh4", "na", "oin", "oc", "bc"]):
            m = m_f * comp_f[:, None, :, s] + m_c * comp_c[:, None, :, s]
            chem[base + REGISTRY_ORDER.index(sp), :, :nk, :] = m / rho     # ug/kg
            dry += m
            vol += m / DENSITY[sp]
            if sp in ("so4", "no3", "cl", "nh4", "na"):
                soluble += m
        water = growth * soluble
        chem[base + REGISTRY_ORDER.index("water"), :, :nk, :] = water / rho
        chem[base + REGISTRY_ORDER.index("hysw"), :, :nk, :] = 0.3 * water / rho   # carried, not used by the optics
        # number from the section's dry volume and a mean diameter around the section centre
        d_um = dmid[b] * rng.uniform(0.45, 2.2, size=shp3)
        vpart = np.pi / 6.0 * (d_um * 1.0e-4) ** 3      # cm3
        number = vol * 1.0e-6 / vpart                    # per m3
        chem[base + REGISTRY_ORDER.index("num"), :, :nk, :] = number / rho
        # some empty sections / cells
        empty = rng.uniform(size=shp3) < empty_fraction
        for sp in REGISTRY_ORDER:
            c = chem[base + REGISTRY_ORDER.index(sp), :, :nk, :]
            c[empty] = 0.0
    return MosaicState(nbin=nbin, chem=chem, alt=alt, dz8w=st.a3["dz8w"].copy(), idx=idx, names=names)


def make_mosaic_rows(ni, nj, nk, j0, j1, nbin=4, seed=20260005, wrf_seed=20260002, empty_fraction=0.02):
    """MOSAIC tracers for rows j0 <= j < j1 of one synthetic domain, row by row from per-row seeds (see
    synth.make_wrf_rows): returns (WrfState of the slab, MosaicState of the slab)."""
    from . import synth
    sts, mss = [], []
    for j in range(j0, j1):
        st = synth.make_wrf_state(ni, 1, nk, seed=wrf_seed + 7919 * j)
        sts.append(st)
        mss.append(make_mosaic_state(st, nbin=nbin, seed=seed + 7919 * j, empty_fraction=empty_fraction))
    slab = synth.WrfState(ni=ni, nj=j1 - j0, nk=nk, p_top=sts[0].p_top if sts else 5000.0)
    if not sts:
        return slab, None
    for n in sts[0].a3:
        slab.a3[n] = np.concatenate([r.a3[n] for r in sts], axis=0)
    for n in sts[0].a2:
        slab.a2[n] = np.concatenate([r.a2[n] for r in sts], axis=0)
    ms = MosaicState(nbin=nbin, chem=np.concatenate([m.chem for m in mss], axis=1),
                     alt=np.concatenate([m.alt for m in mss], axis=0),
                     dz8w=np.concatenate([m.dz8w for m in mss], axis=0), idx=mss[0].idx, names=mss[0].names)
    return slab, ms


@dataclass
class ModalState:
    chem: np.ndarray          # (num_chem, nj, kme, ni)
    idx: np.ndarray           # [3 modes][10] 1-based, SPECIES order
    sigma: tuple
    alt: np.ndarray
    dz8w: np.ndarray
    names: list


def make_made_state(st, seed=20260008, empty_fraction=0.03) -> ModalState:
    """Synthetic MADE/SORGAM-type aerosol: Aitken, accumulation and coarse mode (sigma 1.7 / 2.0 / 2.2) with number
    and species masses per mode; median diameters scatter around 0.03, 0.2 and 3 um; the coarse mode carries sea
    salt and dust only; some cells have empty modes."""
    rng = np.random.default_rng(seed)
    nj, kme, ni = st.a3["rho3d"].shape
    nk = st.nk
    rho = st.a3["rho3d"][:, :nk, :]
    dz = st.a3["dz8w"][:, :nk, :]
    z = np.cumsum(dz, axis=1) - 0.5 * dz
    shp3 = (nj, nk, ni)
    sigma = (1.7, 2.0, 2.2)
    dens = np.array([1.8, 1.8, 2.2, 1.8, 2.2, 2.6, 1.0, 1.7])            # so4 no3 cl nh4 na oin oc bc
    names = ["gas1", "gas2"]
    for m in ("i", "j", "c"):
        names += [f"{s}a{m}" for s in SPECIES[:9]] + [f"n{m}"]
    chem = np.zeros((len(names), nj, kme, ni))
    chem[:2, :, :nk, :] = rng.uniform(0, 1, (2,) + shp3)
    idx = np.zeros((3, 10), dtype=np.int64)
    load0 = [10.0 ** rng.uniform(-1.0, 0.5, shp3), 10.0 ** rng.uniform(0.3, 1.7, shp3), 10.0 ** rng.uniform(0.0, 1.5, shp3)]
    comp = [rng.dirichlet([4, 2, 0.1, 2, 0.1, 0.2, 3, 1.0], shp3), rng.dirichlet([4, 2, 0.3, 2, 0.3, 0.8, 3, 0.6], shp3),
            rng.dirichlet([0.01, 0.01, 2, 0.01, 2, 6, 0.01, 0.01], shp3)]
    dg0 = (0.03, 0.2, 3.0)
    prof = np.exp(-z / 2000.0)
    for m in range(3):
        base = 2 + m * 10
        idx[m] = np.arange(base + 1, base + 11)
        mass = load0[m] * prof / rho                                     # ug per kg of dry air
        dry = mass[..., None] * comp[m]                                   # (nj, nk, ni, 8)
        empty = rng.uniform(size=shp3) < empty_fraction
        dry[empty] = 0.0
        for s in range(8):
            chem[base + s, :, :nk, :] = dry[..., s]
        water = mass * rng.uniform(0.0, 2.0, shp3) * (~empty)
        chem[base + 8, :, :nk, :] = water
        vol = (dry / dens).sum(axis=-1) * 1.0e-6                          # cm3 per kg (ug / (g cm-3) = 1e-6 cm3)
        dgn = dg0[m] * 10.0 ** rng.uniform(-0.25, 0.25, shp3) * 1.0e-4    # cm
        ls = np.log(sigma[m])
        num = vol / (np.pi / 6.0 * dgn ** 3 * np.exp(4.5 * ls * ls))
        num[empty] = 0.0
        chem[base + 9, :, :nk, :] = num
    alt = np.zeros((nj, kme, ni))
    alt[:, :nk, :] = 1.0 / rho
    alt[:, nk:, :] = 1.0
    return ModalState(chem=chem, idx=idx, sigma=sigma, alt=alt, dz8w=st.a3["dz8w"].copy(), names=names)
