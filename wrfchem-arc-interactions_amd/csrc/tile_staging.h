// Staging of a mem = 0 call (host arrays, what the Fortran shims pass): only the tile its:ite x jts:jte of every array
// travels.  The host side packs the tile of each array into ONE pinned buffer ([block][j][k][i] with the tile's own
// extents), one H2D copy brings it to the device, the kernels run on the packed arrays (memory dimensions = tile
// dimensions), one D2H copy brings the output region back and the host side scatters it into the tile of the
// caller's arrays.  Nothing outside the tile is read or written, so OpenMP tile loops that share their WRF arrays
// (DRV:975-982) can call concurrently: tiles are disjoint, and every host thread owns its pinned and device buffers
// (grow-only, reused from call to call -- no hipMalloc / hipFree per call) and its status word.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace rrtmg {

// default cap (MiB) of a radiation entry point's layer scratch: a modest fixed size, and never more than a quarter of
// what the device has FREE at the first call (other tenants -- a framework's allocator, other ranks, other streams --
// keep their room).  One launch per kernel for a very large tile is an explicit opt-in through
// RRTMG_HIP_SW_SCRATCH_MB / RRTMG_HIP_LW_SCRATCH_MB (bench.py sets them); larger tiles otherwise run in column batches.
inline size_t default_scratch_mb(size_t modest_mb)
{
    static std::mutex m;
    static size_t free_at_first_call = 0;   // asked once per process
    {
        std::lock_guard<std::mutex> lock(m);
        if (!free_at_first_call) {
            size_t fr = 0, tot = 0;
            if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); return modest_mb; }
            free_at_first_call = fr ? fr : 1;
        }
    }
    const size_t quarter = (free_at_first_call / 4) >> 20;
    return quarter < modest_mb ? (quarter > 64 ? quarter : 64) : modest_mb;
}

struct TileGeom {
    int ni_m, nk_m, nj_m;   // memory extents of the caller's arrays (ims:ime, kms:kme, jms:jme)
    int i0, j0;             // tile origin inside memory (its - ims, jts - jms)
    int ni_t, nj_t;         // tile extents
    size_t rb;              // bytes per element (4 or 8)
};

struct StagePool {
    void *pin = nullptr, *dev = nullptr;
    size_t pin_cap = 0, dev_cap = 0;
};

// all pools ever created, so that the finalize entry points can release them
inline std::vector<StagePool *> &stage_pools()
{
    static std::vector<StagePool *> v;
    return v;
}
inline std::mutex &stage_pools_mutex()
{
    static std::mutex m;
    return m;
}
inline StagePool &thread_stage_pool()
{
    static thread_local StagePool *p = nullptr;
    if (!p) {
        p = new StagePool();   // lives until stage_pools_release(): a thread may end long after the HIP runtime
        std::lock_guard<std::mutex> lock(stage_pools_mutex());
        stage_pools().push_back(p);
    }
    return *p;
}
inline void stage_pools_release()
{
    std::lock_guard<std::mutex> lock(stage_pools_mutex());
    for (StagePool *p : stage_pools()) {
        if (p->pin) (void)hipHostFree(p->pin);
        if (p->dev) (void)hipFree(p->dev);
        p->pin = p->dev = nullptr;
        p->pin_cap = p->dev_cap = 0;
    }
}

class TileStager {
public:
    explicit TileStager(const TileGeom &g) : g_(g) { used_ = 64; }   // [0, 64): the call's status word (+ padding)

    // kext: extent of the k dimension in memory (1 for a 2-D array, nk_m, nk_m + 2 for the flux profiles);
    // nblk: leading blocks of a 4-D array (i,k,j,block).  target receives the device address in upload().
    void in(const void *host, int kext, int nblk, const void **target)
    {
        *target = nullptr;
        if (!host) return;
        reqs_.push_back({host, nullptr, kext, nblk, used_, target, nullptr});
        used_ += bytes(kext, nblk);
    }
    void out(void *host, int kext, int nblk, void **target)
    {
        *target = nullptr;
        if (!host) return;
        if (out_begin_ == SIZE_MAX) out_begin_ = used_;
        reqs_.push_back({host, host, kext, nblk, used_, nullptr, target});
        used_ += bytes(kext, nblk);
    }

    // pack inputs (and the current contents of the outputs: INOUT semantics, the routines leave night columns and
    // unwritten fields as they were), copy to the device, hand out the device addresses
    hipError_t upload(hipStream_t s)
    {
        StagePool &p = thread_stage_pool();
        if (p.pin_cap < used_) {
            if (p.pin) (void)hipHostFree(p.pin);
            p.pin = nullptr; p.pin_cap = 0;
            hipError_t e = hipHostMalloc(&p.pin, used_ + used_ / 4, hipHostMallocDefault);
            if (e != hipSuccess) return e;
            p.pin_cap = used_ + used_ / 4;
        }
        if (p.dev_cap < used_) {
            if (p.dev) { hipError_t e = hipStreamSynchronize(s); if (e != hipSuccess) return e; (void)hipFree(p.dev); }
            p.dev = nullptr; p.dev_cap = 0;
            hipError_t e = hipMalloc(&p.dev, used_ + used_ / 4);
            if (e != hipSuccess) return e;
            p.dev_cap = used_ + used_ / 4;
        }
        pool_ = &p;
        std::memset(p.pin, 0, 64);
        for (const Req &r : reqs_) {
            copy_tile(r, true);
            char *d = static_cast<char *>(p.dev) + r.off;
            if (r.in_target) *r.in_target = d;
            if (r.out_target) *r.out_target = d;
        }
        return hipMemcpyAsync(p.dev, p.pin, used_, hipMemcpyHostToDevice, s);
    }
    int *status_word() const { return static_cast<int *>(pool_->dev); }

    // bring the output region and the status word back, wait, scatter into the caller's tiles
    hipError_t download(hipStream_t s, int *status)
    {
        StagePool &p = *pool_;
        hipError_t e = hipMemcpyAsync(p.pin, p.dev, 64, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess && out_begin_ != SIZE_MAX)
            e = hipMemcpyAsync(static_cast<char *>(p.pin) + out_begin_, static_cast<char *>(p.dev) + out_begin_,
                               used_ - out_begin_, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return e;
        *status = *static_cast<int *>(p.pin);
        for (const Req &r : reqs_)
            if (r.hout) copy_tile(r, false);
        return hipSuccess;
    }
    size_t staged_bytes() const { return used_; }

private:
    struct Req {
        const void *hin;
        void *hout;
        int kext, nblk;
        size_t off;
        const void **in_target;
        void **out_target;
    };
    size_t bytes(int kext, int nblk) const
    {
        const size_t b = (size_t)nblk * g_.nj_t * kext * g_.ni_t * g_.rb;
        return (b + 63) & ~size_t(63);
    }
    void copy_rows(const Req &r, bool to_pinned, int j0, int j1) const
    {
        const size_t row = (size_t)g_.ni_t * g_.rb;
        char *pk = static_cast<char *>(pool_->pin) + r.off;
        for (int b = 0; b < r.nblk; ++b)
            for (int j = j0; j < j1; ++j)
                for (int k = 0; k < r.kext; ++k) {
                    const size_t h = ((((size_t)b * g_.nj_m + (g_.j0 + j)) * r.kext + k) * g_.ni_m + g_.i0) * g_.rb;
                    const size_t q = (((size_t)b * g_.nj_t + j) * r.kext + k) * row;
                    if (to_pinned) std::memcpy(pk + q, static_cast<const char *>(r.hin) + h, row);
                    else std::memcpy(static_cast<char *>(r.hout) + h, pk + q, row);
                }
    }
    // the packing is plain memory copying; large arrays are split over a few helper threads by rows (a patch of a
    // production domain is hundreds of MB per call, one core moves about 5 GB/s)
    void copy_tile(const Req &r, bool to_pinned) const
    {
        const size_t total = (size_t)r.nblk * g_.nj_t * r.kext * g_.ni_t * g_.rb;
        unsigned nt = total >= (size_t(4) << 20) ? std::thread::hardware_concurrency() : 1;
        if (nt > 8) nt = 8;
        if (nt > (unsigned)g_.nj_t) nt = (unsigned)g_.nj_t;
        if (nt <= 1) { copy_rows(r, to_pinned, 0, g_.nj_t); return; }
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; ++t)
            th.emplace_back([=, &r] { copy_rows(r, to_pinned, (int)((size_t)g_.nj_t * t / nt), (int)((size_t)g_.nj_t * (t + 1) / nt)); });
        copy_rows(r, to_pinned, 0, (int)((size_t)g_.nj_t / nt));
        for (auto &x : th) x.join();
    }
    TileGeom g_;
    std::vector<Req> reqs_;
    size_t used_ = 0, out_begin_ = SIZE_MAX;
    StagePool *pool_ = nullptr;
};

}  // namespace rrtmg
