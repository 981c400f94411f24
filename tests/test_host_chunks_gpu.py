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
