"""Synthetic clouds for a synth.WrfState ("20 % of columns with 1-3 cloudy layers"): the hydrometeor
and cloud-fraction arrays RRTMG_SWRAD / RRTMG_LWRAD receive when icloud /= 0 (SW:9901-9945), plus optional
microphysics effective radii (re_cloud / re_ice / re_snow with has_reqc / has_reqi / has_reqs)."""
from __future__ import annotations

import numpy as np


def add_clouds(st, seed=20260007, cloudy_fraction=0.2, with_radii=False, max_decks=3):
    """Adds cldfra3d, qc3d, qr3d, qi3d, qs3d (and re_cloud, re_ice, re_snow in metres when with_radii) to st.a3.
    Returns the keyword flags for rrtmg_swrad / rrtmg_lwrad."""
    rng = np.random.default_rng(seed)
    nj, kme, ni = st.a3["p3d"].shape
    nk = st.nk
    z = {n: np.zeros((nj, kme, ni)) for n in ("cldfra3d", "qc3d", "qr3d", "qi3d", "qs3d")}
    re = {n: np.zeros((nj, kme, ni)) for n in ("re_cloud", "re_ice", "re_snow")}
    t = st.a3["t3d"]
    for j in range(nj):
        for i in range(ni):
            if rng.uniform() > cloudy_fraction:
                continue
            for _ in range(int(rng.integers(1, max_decks + 1))):
                k0 = int(rng.integers(1, max(2, nk - 6)))
                depth = int(rng.integers(1, 4))
                for k in range(k0, min(k0 + depth, nk - 1)):
                    z["cldfra3d"][j, k, i] = rng.choice([1.0, rng.uniform(0.05, 0.95)])
                    ice = np.clip((268.0 - t[j, k, i]) / 25.0, 0.0, 1.0)
                    z["qc3d"][j, k, i] = (1.0 - ice) * 10.0 ** rng.uniform(-5.5, -3.3)
                    z["qi3d"][j, k, i] = ice * 10.0 ** rng.uniform(-6.0, -4.0)
                    z["qs3d"][j, k, i] = ice * 10.0 ** rng.uniform(-6.0, -4.0) * (rng.uniform() < 0.5)
                    z["qr3d"][j, k, i] = (1.0 - ice) * 10.0 ** rng.uniform(-6.0, -4.5) * (rng.uniform() < 0.3)
                    re["re_cloud"][j, k, i] = rng.uniform(3.0e-6, 18.0e-6)
                    re["re_ice"][j, k, i] = rng.uniform(8.0e-6, 110.0e-6)
                    re["re_snow"][j, k, i] = rng.uniform(30.0e-6, 160.0e-6)
    st.a3.update(z)
    flags = dict(icloud=1, has_reqc=0, has_reqi=0, has_reqs=0, warm_rain=False, progn=0, g=9.81)
    if with_radii:
        st.a3.update(re)
        flags.update(has_reqc=1, has_reqi=1, has_reqs=1)
    return flags
