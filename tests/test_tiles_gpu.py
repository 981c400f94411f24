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


def test_bench_two_ranks_rehearsal_on_one_gpu():
    """`python3 bench.py --gpus 2` end to end on this one-GPU box: the parent starts the two ranks, each builds only
    its j-slab of the domain and runs the three stages on it, the per-tile sums are all-reduced (over gloo here: RCCL
    refuses two ranks on one device) and rank 0 prints one line whose reduced column count is the whole domain's."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--config", "2",
                        "--steps", "1", "--warmup", "1", "--night", "0.2"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["allreduce"]["columns_reduced"] == d["allreduce"]["columns_expected"] == 40000.0
    assert 0 < d["config"]["columns_rank0"] < 40000
    assert 50.0 < d["allreduce"]["mean_lwupt_Wm2"] < 600.0
