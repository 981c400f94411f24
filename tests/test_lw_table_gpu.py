"""The longwave transfer kernel with the exponential / tfn pair table in LDS against the one that re-evaluates every
entry from its index (csrc/lw_kernel.hip: lw_rt_fast_tbl_kernel, lw_rt_fast_kernel): same bits.  The switch is read once
per process, so each setting runs in a child (tests/lw_table_child.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_pair_table_in_lds_gives_the_bits_of_the_reevaluated_entries(hip_lib, tmp_path):
    here = os.path.dirname(os.path.abspath(__file__))
    outs = {}
    for mode in ("1", "0"):
        env = dict(os.environ, RRTMG_HIP_LW_TABLE=mode)
        out = str(tmp_path / f"lw{mode}.npz")
        r = subprocess.run([sys.executable, os.path.join(here, "lw_table_child.py"), str(tmp_path), out], env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "LWTABLE-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
        outs[mode] = np.load(out)
    assert set(outs["0"].files) == set(outs["1"].files) and len(outs["0"].files) > 70
    assert any(n.startswith("k40_cloud_") for n in outs["0"].files)
    for n in outs["0"].files:
        np.testing.assert_array_equal(outs["0"][n], outs["1"][n], err_msg=n)
    assert np.isfinite(outs["1"]["k40_aer_lwupt"]).all() and (outs["1"]["k40_aer_lwupt"] > 0).all()
