"""CPU: the N>1 path -- j-slab sharding and the one collective -- with world_size-2 gloo.
Each rank runs the ORACLE on its own slab (the oracle is allowed in tests), reduces its 12 TOA/surface
fields to the 16-vector, all-reduces it over gloo, and the result must equal the single-process
sums over the whole domain; every column must be computed by exactly one rank."""
import os

import numpy as np
import pytest


def test_slab_bounds_cover_and_balance(wa):
    t = wa.tiles
    for n, world in ((200, 8), (7, 3), (5, 8), (1, 2), (500, 4)):
        seen = []
        for r in range(world):
            js, je = t.slab_bounds(1, n, world, r)
            seen.extend(range(js, je + 1))
            assert je - js + 1 in (n // world, n // world + 1)
        assert seen == list(range(1, n + 1))
    # weighted: balance by daylit columns; slabs stay contiguous, ordered, covering
    w = np.array([0, 0, 0, 10, 10, 10, 10, 0, 0, 10, 10, 10])
    cuts = [t.slab_bounds(1, 12, 3, r, weights=w) for r in range(3)]
    cover = [j for js, je in cuts for j in range(js, je + 1)]
    assert cover == list(range(1, 13))
    loads = [w[js - 1:je].sum() for js, je in cuts]
    assert max(loads) - min(loads) <= 10
    with pytest.raises(ValueError):
        t.slab_bounds(1, 12, 3, 0, weights=w[:-1])


def _worker(rank, world, port, tab, q):
    import torch
    import torch.distributed as dist
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import wrfchem_arc_amd as wa
    from oracle import oracle as orc
    from helpers import run_oracle_swrad
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    st = wa.synth.make_wrf_state(5, 6, 30, seed=20260300, night_fraction=0.3)
    day_rows = (st.a2["xcoszen"] > 0).sum(axis=1)
    js, je = wa.tiles.slab_bounds(1, st.nj, world, rank, weights=day_rows)
    out = run_oracle_swrad(orc.SwOracle(tab), st, tile=dict(jts=js, jte=je), profiles=False)
    v = torch.from_numpy(wa.tiles.tile_sums_host(out, wa.tiles.SW12, 1, st.ni, js, je))
    wa.tiles.allreduce_sums(v)
    q.put((rank, js, je, v.numpy().copy()))
    dist.destroy_process_group()


def test_two_rank_gloo_sums_equal_single_process(wa, sw_table_file, sw_oracle):
    import torch.multiprocessing as mp
    from helpers import run_oracle_swrad
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, sw_table_file, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, js0, je0, v0), (r1, js1, je1, v1) = res
    st = wa.synth.make_wrf_state(5, 6, 30, seed=20260300, night_fraction=0.3)
    assert js0 == 1 and je1 == st.nj and js1 == je0 + 1          # disjoint, ordered, covering
    np.testing.assert_array_equal(v0, v1)                        # both ranks hold the reduced vector
    full = run_oracle_swrad(sw_oracle, st, profiles=False)
    ref = wa.tiles.tile_sums_host(full, wa.tiles.SW12, 1, st.ni, 1, st.nj)
    np.testing.assert_allclose(v0[:13], ref[:13], rtol=1e-13, atol=1e-9)
    assert v0[13] == st.ni * st.nj
    means = wa.tiles.domain_means(v0)
    np.testing.assert_allclose(means[0], full["swupt"].mean(), rtol=1e-12)


def test_weighted_slabs_keep_one_row_per_rank(wa):
    """One very heavy row must not collapse the cuts around it: with at least as many rows as ranks every rank gets a
    row (an empty slab would make that rank skip the kernels but still has to take part in the all-reduce)."""
    t = wa.tiles
    w = np.array([1, 1, 1000, 1, 1, 1, 1, 1])
    cuts = [t.slab_bounds(1, 8, 4, r, weights=w) for r in range(4)]
    assert all(je >= js for js, je in cuts)
    assert [j for js, je in cuts for j in range(js, je + 1)] == list(range(1, 9))
    # fewer rows than ranks: the surplus ranks get an empty slab
    cuts = [t.slab_bounds(1, 2, 4, r, weights=np.array([5, 5])) for r in range(4)]
    assert sum(je - js + 1 for js, je in cuts) == 2


def test_row_generators_build_one_domain(wa):
    """synth.make_wrf_rows / synth_mosaic.make_mosaic_rows: any partition into j-slabs is the same domain, bit for bit
    (what bench.py's strong-scaling mode builds: each rank only its slab)."""
    ni, nj, nk = 6, 7, 12
    full = wa.synth.make_wrf_rows(ni, nj, nk, 0, nj, seed=20260310)
    parts = [wa.synth.make_wrf_rows(ni, nj, nk, a, b, seed=20260310) for a, b in ((0, 3), (3, 4), (4, 7))]
    for n in full.a3:
        np.testing.assert_array_equal(np.concatenate([p.a3[n] for p in parts], axis=0), full.a3[n], err_msg=n)
    for n in full.a2:
        np.testing.assert_array_equal(np.concatenate([p.a2[n] for p in parts], axis=0), full.a2[n], err_msg=n)
    stf, msf = wa.synth_mosaic.make_mosaic_rows(ni, nj, nk, 0, nj, nbin=4, seed=20260311, wrf_seed=20260310)
    np.testing.assert_array_equal(stf.a3["t3d"], full.a3["t3d"])
    _, ms1 = wa.synth_mosaic.make_mosaic_rows(ni, nj, nk, 2, 5, nbin=4, seed=20260311, wrf_seed=20260310)
    np.testing.assert_array_equal(ms1.chem, msf.chem[:, 2:5])
    np.testing.assert_array_equal(ms1.alt, msf.alt[2:5])


def _strong_worker(rank, world, port, tab, q):
    """bench.py's strong-scaling path with the oracle in place of the device: night mask of the whole domain from the
    per-row generator, daylight-weighted slab, only the slab is built, per-tile sums, all-reduce."""
    import torch
    import torch.distributed as dist
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import wrfchem_arc_amd as wa
    import bench
    from oracle import oracle as orc
    from helpers import run_oracle_swrad
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ni, nj, nk = 5, 9, 24
    night = bench.night_rows(ni, nj, 20260320, 0.3)
    js, je = wa.tiles.slab_bounds(1, nj, world, rank, weights=(~night).sum(axis=1))
    st = wa.synth.make_wrf_rows(ni, nj, nk, js - 1, je, seed=20260321)
    st.a2["xcoszen"] = np.where(night[js - 1:je], -0.3, st.a2["xcoszen"])
    out = run_oracle_swrad(orc.SwOracle(tab), st, profiles=False)          # the slab is the rank's whole memory
    v = torch.from_numpy(wa.tiles.tile_sums_host(out, wa.tiles.SW12, 1, ni, 1, st.nj))
    wa.tiles.allreduce_sums(v)
    q.put((rank, js, je, v.numpy().copy()))
    dist.destroy_process_group()


def test_strong_scaling_slabs_reduce_to_the_single_process_sums(wa, sw_table_file, sw_oracle):
    import torch.multiprocessing as mp
    import bench
    from helpers import run_oracle_swrad
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_strong_worker, args=(r, 2, port, sw_table_file, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, js0, je0, v0), (r1, js1, je1, v1) = res
    ni, nj, nk = 5, 9, 24
    assert js0 == 1 and je1 == nj and js1 == je0 + 1
    np.testing.assert_array_equal(v0, v1)
    night = bench.night_rows(ni, nj, 20260320, 0.3)
    st = wa.synth.make_wrf_rows(ni, nj, nk, 0, nj, seed=20260321)          # N = 1: the whole domain in one process
    st.a2["xcoszen"] = np.where(night, -0.3, st.a2["xcoszen"])
    full = run_oracle_swrad(sw_oracle, st, profiles=False)
    ref = wa.tiles.tile_sums_host(full, wa.tiles.SW12, 1, ni, 1, nj)
    np.testing.assert_allclose(v0[:13], ref[:13], rtol=1e-13, atol=1e-9)
    assert v0[13] == ni * nj


def test_bench_starts_its_own_ranks_dry():
    """`python3 bench.py --gpus 2` is how the driver starts a scaling run: with no launcher environment the parent
    must start the ranks itself (child process, torch.distributed.run, rendezvous on 127.0.0.1) and relay rank 0's
    line and the exit code.  --dry-launch walks that path without a GPU: gloo, slab of each rank, one all-reduce."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-launch", "--night", "0.1"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                  # rank 0 only
    out = json.loads(lines[0])
    assert out["dry_launch"] and out["n_gpus"] == 2 and out["scaling"] == "strong"
    s0, s1 = out["slabs"]
    assert (s0["rank"], s1["rank"]) == (0, 1) and (s0["local_rank"], s1["local_rank"]) == (0, 1)
    assert s0["js"] == 1 and s1["js"] == s0["je"] + 1 and s1["je"] == 500      # disjoint, ordered, covering
    assert out["columns_all_ranks"] == 250000.0 == s0["columns"] + s1["columns"]
    # one rank: no launcher, same line shape
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--dry-launch", "--config", "2"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["slabs"][0]["columns"] == 40000
    # a failing rank's exit code comes back through the parent (no GPU here: the ranks refuse to run)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--no-cpu"],
                       capture_output=True, text=True, env=dict(env, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""), timeout=300)
    assert r.returncode != 0
