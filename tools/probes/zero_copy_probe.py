"""What would the kernels cost if they read a host model's arrays over PCIe themselves (pinned host memory handed to the
mem = 1 entry points, no staging at all)?  500 x 40 x 50 tile, SW + LW, against the packed mem = 0 path.
Run on the GPU box: python tools/probes/zero_copy_probe.py"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import wrfchem_arc_amd as wa  # noqa: E402


class AsDevice:
    """a pinned host tensor passed as if it were device memory (the address is valid on the device too)"""
    def __init__(self, t):
        self.t = t
        self.is_cuda = True

    def is_contiguous(self):
        return True

    def data_ptr(self):
        return self.t.data_ptr()

    def element_size(self):
        return self.t.element_size()


ni, nj, nk = 500, 40, 50
tmp = tempfile.mkdtemp()
tab, tabl = os.path.join(tmp, "SW"), os.path.join(tmp, "LW")
wa.synth.write_sw_data(tab, seed=20260001, real_bytes=8)
wa.synth.write_lw_data(tabl, seed=20260003, real_bytes=8)
wa.engine.rrtmg_swinit(tab, 8, device=0)
wa.engine_lw.rrtmg_lwinit(bench.P_TOP, tabl, kme=nk + 1, device=0)
st = wa.synth.make_wrf_rows(ni, nj, nk, 0, nj, seed=20260002)
d = bench.dims_of(ni, nj, nk)
pin = lambda a: torch.from_numpy(np.ascontiguousarray(a)).pin_memory()
hin = {n: pin(st.a2[n]) for n in bench.SW_IN2}
hin.update({n: pin(st.a3[n]) for n in bench.ST_IN3 + bench.SW_AER})
hout = {"rthratensw": pin(np.zeros((nj, nk + 1, ni)))}
hout.update({n: pin(np.zeros((nj, ni))) for n in bench.SW_OUT2})
lin = {n: pin(st.a2[n]) for n in bench.LW_IN2}
lin.update({n: hin[n] for n in bench.ST_IN3})
lin.update({n: pin(st.a3[n]) for n in bench.LW_AER})
lout = {"rthratenlw": pin(np.zeros((nj, nk + 1, ni)))}
lout.update({n: pin(np.zeros((nj, ni))) for n in bench.LW_OUT2})
W = lambda dd: {n: AsDevice(t) for n, t in dd.items()}
N = lambda dd: {n: t.numpy() for n, t in dd.items()}


def run(sw_in, sw_out, lw_in, lw_out, sync):
    def sw():
        wa.engine.rrtmg_swrad(solcon=1370.0, aer_ra_feedback=1, clean_atm_diag=1, icloud=0, **d, **sw_in, **sw_out)
        if sync:
            torch.cuda.synchronize()

    def lw():
        wa.engine_lw.rrtmg_lwrad(aer_ra_feedback=1, clean_atm_diag=1, icloud=0, **d, **lw_in, **lw_out)
        if sync:
            torch.cuda.synchronize()
    res = {}
    for f in (sw, lw):
        f()
    for name, f in (("sw", sw), ("lw", lw)):
        t = time.perf_counter()
        for _ in range(3):
            f()
        res[name] = (time.perf_counter() - t) / 3
    return res


packed = run(N(hin), N(hout), N(lin), N(lout), False)
ref = {n: t.numpy().copy() for n, t in list(hout.items()) + list(lout.items())}
zero = run(W(hin), W(hout), W(lin), W(lout), True)
same = all(np.array_equal(ref[n], t.numpy()) for n, t in list(hout.items()) + list(lout.items()))
for tag, r in (("packed mem=0", packed), ("zero-copy", zero)):
    print("%-14s sw %.2f ms  lw %.2f ms  pair %.2f ms = %.2f M columns/s" % (
        tag, 1e3 * r["sw"], 1e3 * r["lw"], 1e3 * (r["sw"] + r["lw"]), ni * nj / (r["sw"] + r["lw"]) / 1e6))
print("same bits:", same)
