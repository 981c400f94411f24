"""Sharding of the (i,j) plane over GPUs and the one collective of the path.

Columns are independent (no horizontal coupling anywhere in RRTMG; the host
model already treats radiation as tile-local, module_radiation_driver.F:975-982),
so each rank owns a contiguous slab of j rows -- j is the slowest WRF index, a
slab is one contiguous block of every 2-D array and one strided block per k of
every 3-D array -- and no halo or data-path collective is needed.  The only
exchange is the sum of the per-tile flux / forcing sums (16 doubles per call
family), from which the domain means of the reference's post-processing follow
(NCL_extraction_package/misc_stats_library.ncl:433).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .engine import ERRLEN, _check, load_library

SW12 = ["swupt", "swuptc", "swuptcln", "swdnt", "swdntc", "swdntcln",
        "swupb", "swupbc", "swupbcln", "swdnb", "swdnbc", "swdnbcln"]
LW12 = ["lwupt", "lwuptc", "lwuptcln", "lwdnt", "lwdntc", "lwdntcln",
        "lwupb", "lwupbc", "lwupbcln", "lwdnb", "lwdnbc", "lwdnbcln"]


def slab_bounds(jts: int, jte: int, world: int, rank: int, weights=None):
    """Contiguous j-slab [js, je] (inclusive, WRF indices) of rank `rank`.

    Without weights the rows are split as evenly as possible (first `rem` ranks
    get one more row).  With `weights` (one non-negative number per j row, e.g.
    the daylit-column count: night columns cost nothing in the shortwave) the
    cut points are placed where the running weight crosses k/world of the total.
    Slabs are disjoint, ordered, cover [jts, jte].  With at least as many rows as
    ranks every rank gets at least one row (a heavy row cannot collapse the cuts
    around it); with fewer rows than ranks the surplus ranks get an empty slab
    (js > je), which every entry point of the engine treats as a no-op."""
    n = jte - jts + 1
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    if weights is None:
        q, r = divmod(n, world)
        start = rank * q + min(rank, r)
        size = q + (1 if rank < r else 0)
        return jts + start, jts + start + size - 1
    w = np.asarray(weights, dtype=np.float64)
    if w.shape != (n,) or (w < 0).any():
        raise ValueError("weights must be one non-negative number per j row")
    cum = np.concatenate([[0.0], np.cumsum(w)])
    total = cum[-1]
    if total <= 0:
        return slab_bounds(jts, jte, world, rank)
    cuts = [0]
    for k in range(1, world):
        cuts.append(int(np.searchsorted(cum, total * k / world, side="left")))
    cuts.append(n)
    if n >= world:
        for k in range(1, world):          # at least one row per rank: push collapsed cuts apart ...
            cuts[k] = max(cuts[k], cuts[k - 1] + 1)
        for k in range(world - 1, 0, -1):  # ... and keep room for the ranks behind
            cuts[k] = min(cuts[k], cuts[k + 1] - 1)
    else:
        for k in range(1, world + 1):
            cuts[k] = max(cuts[k], cuts[k - 1])
    return jts + cuts[rank], jts + cuts[rank + 1] - 1


def tile_sums_host(outputs: dict, names, its, ite, jts, jte, ims=1, jms=1):
    """Reference (numpy) definition of the 16-vector: 12 sums, (uptcln - upt), column count, 0, 0."""
    v = np.zeros(16)
    for i, n in enumerate(names):
        v[i] = np.asarray(outputs[n], dtype=np.float64)[jts - jms:jte - jms + 1, its - ims:ite - ims + 1].sum()
    v[12] = v[2] - v[0]
    v[13] = float((ite - its + 1) * (jte - jts + 1))
    return v


def tile_sums_device(outputs: dict, names, dims: dict, out16):
    """Native device reduction (rrtmg_hip_tile_sums) of 12 torch CUDA tensors into out16 (CUDA float64[16])."""
    import torch
    lib = load_library()
    lib.rrtmg_hip_tile_sums.argtypes = [C.POINTER(C.c_void_p), C.c_int] + [C.c_int] * 8 + [C.c_void_p, C.c_void_p,
                                                                                        C.c_char_p]
    ptrs = (C.c_void_p * 12)(*[outputs[n].data_ptr() for n in names])
    rb = outputs[names[0]].element_size()
    err = C.create_string_buffer(ERRLEN)
    _check(lib.rrtmg_hip_tile_sums(ptrs, rb, dims["ims"], dims["ime"], dims["jms"], dims["jme"], dims["its"],
                                   dims["ite"], dims["jts"], dims["jte"], out16.data_ptr(),
                                   torch.cuda.current_stream().cuda_stream, err), err)
    return out16


def allreduce_sums(vec, group=None):
    """Sum the per-rank vectors (torch tensor, any backend).  No-op without an initialised process group."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(vec, group=group)
    return vec


def domain_means(total16):
    """Domain means of the 12 fields and of the clean-minus-aerosol TOA term from the all-reduced vector."""
    t = np.asarray(total16, dtype=np.float64)
    n = t[13]
    return t[:13] / n if n > 0 else np.zeros(13)
