"""The chunked host path (mem = 0: j-chunks pipelined over the copy and compute streams, csrc/tile_staging.h) with a
chunk size small enough that test-sized tiles are cut into the maximum of 16 chunks.  The chunk size is read once per
process, so the calls run in a child (tests/chunk_child.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_many_chunks_equal_the_device_resident_call(hip_lib, tmp_path):
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, RRTMG_HIP_CHUNK_COLS="12")
    r = subprocess.run([sys.executable, os.path.join(here, "chunk_child.py"), str(tmp_path)], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "CHUNKS-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_registered_host_arrays_equal_the_packed_path(hip_lib, tmp_path):
    """RRTMG_HIP_REGISTER_HOST=1 (the caller's arrays pinned in place, pitched copies instead of packing): the same
    calls -- whole tile and a partial one, cloudy, both radiation entry points and the optical averaging -- give the same
    bits as the packed path, inside the tile and outside it."""
    import numpy as np
    here = os.path.dirname(os.path.abspath(__file__))
    outs = {}
    for mode in ("0", "1"):
        env = dict(os.environ, RRTMG_HIP_REGISTER_HOST=mode, RRTMG_HIP_CHUNK_COLS="96")
        out = str(tmp_path / f"res{mode}.npz")
        r = subprocess.run([sys.executable, os.path.join(here, "register_child.py"), str(tmp_path), out], env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "REGISTER-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
        outs[mode] = np.load(out)
    assert set(outs["0"].files) == set(outs["1"].files) and len(outs["0"].files) > 60
    for n in outs["0"].files:
        np.testing.assert_array_equal(outs["0"][n], outs["1"][n], err_msg=n)
