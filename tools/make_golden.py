#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE Fortran (oracle/_ref,
built by oracle/build_ref.sh from /root/reference) on synthetic tables and
columns.  Runs only in the build container (needs /root/reference); the
fixtures it writes are committed and are what travels to the GPU box.

Each fixture holds inputs and the reference's outputs (data only).
"""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wrfchem_arc_amd as wa  # noqa: E402

synth = wa.synth
REF = os.path.join(ROOT, "oracle", "_ref")
GOLD = os.path.join(ROOT, "tests", "golden")

SW_COL_OUT = [("swuflx", 1), ("swdflx", 1), ("swhr", 0), ("swuflxc", 1), ("swdflxc", 1), ("swhrc", 0),
              ("swuflxcln", 1), ("swdflxcln", 1), ("sibvisdir", 1), ("sibvisdif", 1), ("sibnirdir", 1),
              ("sibnirdif", 1), ("swdkdir", 1), ("swdkdif", 1)]
WAVEMIN = np.array([3.077, 2.500, 2.150, 1.942, 1.626, 1.299, 1.242, 0.778, 0.625, 0.442, 0.345, 0.263, 0.200, 3.846])
WAVEMAX = np.array([3.846, 3.077, 2.500, 2.150, 1.942, 1.626, 1.299, 1.242, 0.778, 0.625, 0.442, 0.345, 0.263, 12.195])


def sw_columns_from_state(st, variant=0):
    """rrtmg_sw-level inputs (what RRTMG_SWRAD hands to rrtmg_sw) for every
    column of a synthetic WRF state; an independent numpy construction used
    only to obtain plausible inputs."""
    nk, ncol = st.nk, st.ncol
    nlay = nk + 1

    def col3(name, n):  # (nj,kme,ni) -> (ncol, n)
        return np.ascontiguousarray(np.transpose(st.a3[name][:, :n, :], (0, 2, 1)).reshape(ncol, n))

    plev = np.empty((ncol, nlay + 1))
    plev[:, :nk + 1] = col3("p8w", nk + 1) / 100.0
    plev[:, nk + 1] = 1.0e-5
    tlev = np.empty((ncol, nlay + 1))
    tlev[:, :nk + 1] = col3("t8w", nk + 1)
    tlev[:, nk + 1] = tlev[:, nk]
    play = np.empty((ncol, nlay))
    play[:, :nk] = col3("p3d", nk) / 100.0
    play[:, nk] = 0.5 * plev[:, nk]
    tlay = np.empty((ncol, nlay))
    tlay[:, :nk] = col3("t3d", nk)
    tlay[:, nk] = tlev[:, nk]
    h2o = np.empty((ncol, nlay))
    h2o[:, :nk] = np.maximum(col3("qv3d", nk), 1e-12) * 1.607793
    h2o[:, nk] = h2o[:, nk - 1]
    o3 = 3e-8 + 8e-6 * np.exp(-((np.log(play) - np.log(8.0)) / 1.2) ** 2)
    ones = np.ones((ncol, nlay))
    cols = dict(play=play, plev=plev, tlay=tlay, tlev=tlev, tsfc=st.a2["tsk"].reshape(ncol),
                h2ovmr=h2o, o3vmr=o3, co2vmr=379e-6 * ones, ch4vmr=1774e-9 * ones, n2ovmr=319e-9 * ones,
                o2vmr=0.209488 * ones)
    alb = st.a2["albedo"].reshape(ncol)
    cz = st.a2["xcoszen"].reshape(ncol)
    scal = np.stack([alb, alb * (0.9 if variant else 1.0), alb * (1.1 if variant else 1.0), alb,
                     cz, np.full(ncol, 1.0 if not variant else 1.017), np.full(ncol, 1370.0)], axis=1)
    cols["scal"] = scal
    wavemid = 0.5 * (WAVEMIN + WAVEMAX)
    t3, t4, t9 = col3("tauaer300", nk), col3("tauaer400", nk), col3("tauaer999", nk)
    ang = np.log(t3 / t9) / np.log(999.0 / 300.0)
    w4, w6 = col3("waer400", nk), col3("waer600", nk)
    g4, g6 = col3("gaer400", nk), col3("gaer600", nk)
    tau = np.zeros((ncol, 14, nlay)); ssa = np.ones((ncol, 14, nlay)); asm = np.zeros((ncol, 14, nlay))
    for nb in range(14):
        tau[:, nb, :nk] = t4 * (0.4 / wavemid[nb]) ** ang
        ssa[:, nb, :nk] = np.clip((w6 - w4) / 0.2 * (wavemid[nb] - 0.6) + w6, 0.4, 1.0)
        asm[:, nb, :nk] = np.clip((g6 - g4) / 0.2 * (wavemid[nb] - 0.6) + g6, 0.5, 1.0)
    if variant:  # make a few layers aerosol-free and a few columns aerosol-free
        tau[:, :, nk // 2:] = 0.0
        tau[::5] = 0.0
    cols.update(tauaer=tau, ssaaer=ssa, asmaer=asm)
    return cols


def write_sw_columns(path, cols, clean=1, nrep=1):
    ncol, nlay = cols["play"].shape
    with open(path, "wb") as f:
        f.write(struct.pack("<4i", ncol, nlay, clean, nrep))
        for ic in range(ncol):
            for k in ("play", "plev", "tlay", "tlev"):
                f.write(cols[k][ic].astype("<f8").tobytes())
            f.write(struct.pack("<d", cols["tsfc"][ic]))
            for k in ("h2ovmr", "o3vmr", "co2vmr", "ch4vmr", "n2ovmr", "o2vmr"):
                f.write(cols[k][ic].astype("<f8").tobytes())
            f.write(cols["scal"][ic].astype("<f8").tobytes())
            for k in ("tauaer", "ssaaer", "asmaer"):
                # Fortran reshape(buf,(nlay,nbndsw)): layer fastest == [band][layer] C order
                f.write(np.ascontiguousarray(cols[k][ic]).astype("<f8").tobytes())


def read_sw_out(path, ncol, nlay):
    raw = np.fromfile(path, dtype="<f8")
    per = sum(nlay + e for _, e in SW_COL_OUT)
    raw = raw.reshape(ncol, per)
    out, o = {}, 0
    for name, e in SW_COL_OUT:
        n = nlay + e
        out[name] = raw[:, o:o + n].copy()
        o += n
    return out


def run_ref_sw(tables, cols, clean=1, dump=None, nrep=1):
    exe = os.path.join(REF, "ref_sw")
    with tempfile.TemporaryDirectory() as td:
        fc, fo = os.path.join(td, "c.bin"), os.path.join(td, "o.bin")
        write_sw_columns(fc, cols, clean, nrep)
        cmd = [exe, tables, fc, fo] + ([dump] if dump else [])
        r = subprocess.run(cmd, check=True, capture_output=True, text=True)
        return read_sw_out(fo, *cols["play"].shape), r.stdout


def main():
    os.makedirs(GOLD, exist_ok=True)
    with tempfile.TemporaryDirectory() as td:
        tab = os.path.join(td, "RRTMG_SW_DATA")
        synth.write_sw_data(tab, seed=20260001, real_bytes=8)
        dump = os.path.join(td, "dump.bin")
        for name, (ni, nj, nk, seed, variant) in {
            "sw_cols_k40": (8, 4, 40, 20260011, 0),
            "sw_cols_k50_variant": (4, 2, 50, 20260012, 1),
        }.items():
            st = synth.make_wrf_state(ni, nj, nk, seed=seed)
            cols = sw_columns_from_state(st, variant)
            ref, _ = run_ref_sw(tab, cols, clean=1, dump=dump)
            np.savez_compressed(os.path.join(GOLD, name + ".npz"), table_seed=20260001,
                                **{"in_" + k: v for k, v in cols.items()},
                                **{"ref_" + k: v for k, v in ref.items()})
            print(name, "columns", cols["play"].shape, "swuflx TOA", ref["swuflx"][:3, -1])
        # reduced-table dump of the reference init: keep a strided sample + sums
        d = np.fromfile(dump, dtype="<f8")
        np.savez_compressed(os.path.join(GOLD, "sw_reduced_tables.npz"), table_seed=20260001,
                            n=d.size, stride=np.int64(29), sample=d[::29].copy(),
                            total=np.float64(d.sum()), exp_tbl=d[:10001].copy())
        print("reduced tables:", d.size, "values")


if __name__ == "__main__":
    main()
