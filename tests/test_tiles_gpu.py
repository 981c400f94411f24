"""GPU: the native per-tile flux sums (rrtmg_hip_tile_sums) against numpy, and determinism."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_tile_sums_device_matches_numpy(hip_lib, wa):
    import torch
    rng = np.random.default_rng(5)
    ni, nj = 37, 23
    names = wa.tiles.SW12
    host = {n: rng.uniform(0, 400, size=(nj, ni)) for n in names}
    devt = {n: torch.from_numpy(v).cuda() for n, v in host.items()}
    dims = dict(ims=1, ime=ni, jms=1, jme=nj, its=4, ite=30, jts=2, jte=20)
    out = torch.zeros(16, dtype=torch.float64, device="cuda")
    wa.tiles.tile_sums_device(devt, names, dims, out)
    torch.cuda.synchronize()
    ref = wa.tiles.tile_sums_host(host, names, 4, 30, 2, 20)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-13)
    out2 = torch.zeros(16, dtype=torch.float64, device="cuda")
    wa.tiles.tile_sums_device(devt, names, dims, out2)
    torch.cuda.synchronize()
    assert torch.equal(out, out2)                      # fixed summation order: bitwise reproducible
    dev32 = {n: v.float() for n, v in devt.items()}
    wa.tiles.tile_sums_device(dev32, names, dims, out2)
    torch.cuda.synchronize()
    np.testing.assert_allclose(out2.cpu().numpy(), ref, rtol=1e-6)
