// Staging of a mem = 0 call (host arrays, what the Fortran shims pass): only the tile its:ite x jts:jte of every array
// travels, and it travels in j-CHUNKS so that the stages of neighbouring chunks overlap:
//
//     host thread     pack 0 | pack 1 | pack 2 | ...                             | scatter outputs
//     copy-in stream      H2D 0 | H2D 1 | H2D 2 | ...
//     compute stream            kernels 0 | kernels 1 | kernels 2 | ...
//     copy-out stream                       D2H 0 |     D2H 1 |     D2H 2 | ...
//
// The host side packs the rows of a chunk of each array into the chunk's block of ONE pinned buffer ([block][j][k][i]
// with the chunk's own extents), one asynchronous H2D copy brings the block to the device, the kernels run on the
// packed arrays (memory dimensions = chunk dimensions), one asynchronous D2H copy brings the chunk's output region
// back; while the device works on chunk c the host packs chunk c + 1 and the link carries chunk c + 1 in and chunk
// c - 1 out (three non-blocking streams per calling thread, chained by events: on ONE stream the copies and kernels
// of successive chunks would run one after the other -- measured: 19.4 ms per call pair for 20 000 columns against
// 5.8 ms of link time and 8.4 ms of kernel time).  Concurrent OpenMP tiles (DRV:975-982) do not serialise on the null
// stream, and the call returns after one wait on the copy-out stream and the scatter of the outputs into the
// caller's tile.
// Nothing outside the tile is read or written; tiles of concurrent callers are disjoint, and every host thread owns
// its pinned and device buffers (grow-only, reused from call to call -- no hipMalloc / hipFree per call; a thread
// that ends hands its buffers to the next new thread) and its status words (one per chunk).
// The packing itself is plain memory copying (a patch of a production domain is hundreds of MB per call, one core
// moves about 5 GB/s): a small persistent pool of helper threads shares the (array, row) items of a chunk with the
// caller; when several tiles call at once only one of them gets the helpers, the others pack alone.
#pragma once
#include <hip/hip_runtime.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

namespace rrtmg {

// default cap (MiB) of a radiation entry point's layer scratch: a modest fixed size, and never more than a quarter of
// what the device has free WHEN THE BUFFER HAS TO GROW (other tenants -- a framework's allocator, other ranks, the other
// engine, other threads' streams -- keep their room; ADVICE r2, r3: the figure is asked for every time, it is cheap
// next to the hipMalloc that follows).  One launch per kernel for a very large tile is an explicit opt-in through
// RRTMG_HIP_SW_SCRATCH_MB / RRTMG_HIP_LW_SCRATCH_MB (bench.py sets them); larger tiles otherwise run in column batches.
inline size_t default_scratch_mb(size_t modest_mb)
{
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); return modest_mb; }
    const size_t quarter = (fr / 4) >> 20;
    return quarter < modest_mb ? (quarter > 64 ? quarter : 64) : modest_mb;
}

struct TileGeom {
    int ni_m, nk_m, nj_m;   // memory extents of the caller's arrays (ims:ime, kms:kme, jms:jme)
    int i0, j0;             // tile origin inside memory (its - ims, jts - jms)
    int ni_t, nj_t;         // tile extents
    size_t rb;              // bytes per element (4 or 8)
};

// ---- helper threads for the packing (persistent; RRTMG_HIP_PACK_THREADS = helpers besides the caller, default up
// to 7, 0 = none) ----
class PackPool {
public:
    static PackPool &instance()
    {
        static PackPool p;
        return p;
    }
    // fn(i) for i in [0, n): shared with the helpers when they are free, else done by the caller alone
    void run(int n, const std::function<void(int)> &fn)
    {
        if (n <= 0) return;
        if (workers_.empty() || n < 4 || !busy_.try_lock()) {
            for (int i = 0; i < n; ++i) fn(i);
            return;
        }
        {
            std::lock_guard<std::mutex> lock(m_);
            fn_ = &fn; n_ = n; next_.store(0); pending_ = (int)workers_.size(); ++gen_;
        }
        cv_.notify_all();
        work(fn, n);
        {
            std::unique_lock<std::mutex> lock(m_);
            done_.wait(lock, [&] { return pending_ == 0; });
            fn_ = nullptr;
        }
        busy_.unlock();
    }

private:
    PackPool()
    {
        // half of the cores this PROCESS may run on (its affinity mask: with one MPI rank per few cores the machine's
        // core count would oversubscribe the node), less the caller
        int cores = (int)std::thread::hardware_concurrency();
        {
            cpu_set_t set;
            CPU_ZERO(&set);
            if (sched_getaffinity(0, sizeof set, &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0 && c < cores) cores = c; }
        }
        int n = cores / 2 - 1;
        if (n > 7) n = 7;
        if (const char *e = std::getenv("RRTMG_HIP_PACK_THREADS")) n = std::atoi(e);
        if (n < 0) n = 0;
        if (n > 31) n = 31;
        for (int i = 0; i < n; ++i) workers_.emplace_back([this] { loop(); });
    }
    ~PackPool()
    {
        {
            std::lock_guard<std::mutex> lock(m_);
            stop_ = true; ++gen_;
        }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
    void work(const std::function<void(int)> &fn, int n)
    {
        for (int i = next_.fetch_add(1); i < n; i = next_.fetch_add(1)) fn(i);
    }
    void loop()
    {
        unsigned long seen = 0;
        for (;;) {
            const std::function<void(int)> *fn;
            int n;
            {
                std::unique_lock<std::mutex> lock(m_);
                cv_.wait(lock, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                fn = fn_; n = n_;
            }
            if (fn) work(*fn, n);
            {
                std::lock_guard<std::mutex> lock(m_);
                if (--pending_ == 0) done_.notify_all();
            }
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_, busy_;
    std::condition_variable cv_, done_;
    const std::function<void(int)> *fn_ = nullptr;
    int n_ = 0, pending_ = 0;
    std::atomic<int> next_{0};
    unsigned long gen_ = 0;
    bool stop_ = false;
};

// ---- RRTMG_HIP_REGISTER_HOST=1: the caller's arrays pinned in place, no packing ----
// The packing above is a second pass over every byte on the host (and the scatter a third); on a box whose cores are
// busy it, not the link or the kernels, sets the pace of a mem = 0 call.  A host model's arrays live as long as the
// run, so they can be pinned where they are (hipHostRegister, once per array, remembered by base address) and the
// chunks travel by pitched copies straight between the caller's tile and the device buffers -- the host thread only
// enqueues.  OPT-IN, because the engine cannot see an array die: memory that is unmapped while registered and mapped
// again at the same address (a large temporary of a scripting host) would be read through the stale pinning.  Use it
// where the arrays passed to the engine are allocated once (WRF's grid fields); rrtmg_hip_finalize & co. unpin.
// An array that cannot be registered still travels correctly (the runtime stages pageable memory itself), only slower.
inline bool host_registration_enabled()   // (read at every call: a host may switch it per phase)
{
    const char *e = std::getenv("RRTMG_HIP_REGISTER_HOST");
    return e && e[0] == '1';
}
struct HostRegistry {
    std::mutex m;
    std::unordered_map<const void *, size_t> bytes;   // base address -> registered bytes (0: registration refused)
    static HostRegistry &instance() { static HostRegistry r; return r; }
    void ensure(const void *p, size_t n)
    {
        std::lock_guard<std::mutex> lock(m);
        auto it = bytes.find(p);
        if (it != bytes.end() && it->second == 0) return;   // refused before: the runtime stages it itself
        if (it != bytes.end() && it->second >= n) {
            // remembered: still pinned?  (memory unmapped and mapped again at this address is not; ask the runtime
            // instead of trusting the address)
            void *d = nullptr;
            if (hipHostGetDevicePointer(&d, const_cast<void *>(p), 0) == hipSuccess) return;
            (void)hipGetLastError();
            (void)hipHostUnregister(const_cast<void *>(p));
            (void)hipGetLastError();
        } else if (it != bytes.end()) {
            // grew: pin the larger range -- after every copy that may still read the old one has ended (other
            // threads' streams included)
            (void)hipDeviceSynchronize();
            (void)hipHostUnregister(const_cast<void *>(p));
        }
        const hipError_t e = hipHostRegister(const_cast<void *>(p), n, hipHostRegisterDefault);
        if (e != hipSuccess) (void)hipGetLastError();
        bytes[p] = (e == hipSuccess || e == hipErrorHostMemoryAlreadyRegistered) ? n : 0;
    }
    // one array: before the host frees it (rrtmg_hip_unregister_host)
    void release(const void *p)
    {
        std::lock_guard<std::mutex> lock(m);
        auto it = bytes.find(p);
        if (it == bytes.end()) return;
        if (it->second) {
            (void)hipDeviceSynchronize();
            if (hipHostUnregister(const_cast<void *>(p)) != hipSuccess) (void)hipGetLastError();
        }
        bytes.erase(it);
    }
    void release_all()
    {
        std::lock_guard<std::mutex> lock(m);
        for (auto &kv : bytes)
            if (kv.second) { if (hipHostUnregister(const_cast<void *>(kv.first)) != hipSuccess) (void)hipGetLastError(); }
        bytes.clear();
    }
};

// ---- per-thread staging buffers and stream ----
struct StagePool {
    void *pin = nullptr, *dev = nullptr;
    size_t pin_cap = 0, dev_cap = 0;
    // non-blocking streams, created at the pool's first use on the engine's device: kernels; copies in; copies out
    hipStream_t stream = nullptr, h2d = nullptr, d2h = nullptr;
    hipEvent_t ev_in[16] = {}, ev_k[16] = {};   // chunk c is on the device / chunk c's kernels are done
    bool in_use = false;            // owned by a live thread (guarded by stage_pools_mutex)
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
struct StagePoolHolder {
    StagePool *p = nullptr;
    ~StagePoolHolder()
    {
        // the thread ends (possibly long after the HIP runtime): no HIP call here, the pool -- buffers and stream --
        // goes back to the list and the next new thread takes it over
        if (!p) return;
        std::lock_guard<std::mutex> lock(stage_pools_mutex());
        p->in_use = false;
    }
};
inline StagePool &thread_stage_pool()
{
    static thread_local StagePoolHolder h;
    if (!h.p) {
        std::lock_guard<std::mutex> lock(stage_pools_mutex());
        for (StagePool *q : stage_pools())
            if (!q->in_use) { h.p = q; break; }
        if (!h.p) {
            h.p = new StagePool();   // lives until the process ends; its buffers until stage_pools_release()
            stage_pools().push_back(h.p);
        }
        h.p->in_use = true;
    }
    return *h.p;
}
// Frees the buffers and streams of every pool.  Called by the finalize entry points once no engine is initialised any
// more (c_abi.hip: stage_pools_release_if_idle); a host must not finalize while other threads are inside a call.
inline void stage_pools_release()
{
    HostRegistry::instance().release_all();
    std::lock_guard<std::mutex> lock(stage_pools_mutex());
    for (StagePool *p : stage_pools()) {
        for (hipStream_t *q : {&p->stream, &p->h2d, &p->d2h})
            if (*q) { (void)hipStreamSynchronize(*q); (void)hipStreamDestroy(*q); *q = nullptr; }
        for (int i = 0; i < 16; ++i) {
            if (p->ev_in[i]) { (void)hipEventDestroy(p->ev_in[i]); p->ev_in[i] = nullptr; }
            if (p->ev_k[i]) { (void)hipEventDestroy(p->ev_k[i]); p->ev_k[i] = nullptr; }
        }
        if (p->pin) (void)hipHostFree(p->pin);
        if (p->dev) (void)hipFree(p->dev);
        p->pin = p->dev = nullptr;
        p->pin_cap = p->dev_cap = 0;
    }
}
void stage_pools_release_if_idle();   // c_abi.hip: when the SW, LW and optics engines are all finalized

// columns a chunk of a mem = 0 call should hold: enough to fill the device a few times over, small enough that four
// or more chunks overlap on a production patch (RRTMG_HIP_CHUNK_COLS overrides)
inline long stage_chunk_columns()
{
    static const long v = [] {
        long n = 4096;
        if (const char *e = std::getenv("RRTMG_HIP_CHUNK_COLS")) { const long q = std::atol(e); if (q > 0) n = q; }
        return n;
    }();
    return v;
}

class TileStager {
public:
    static constexpr int MAX_CHUNKS = 16;   // one status word each in the first 64 bytes of the buffers
    explicit TileStager(const TileGeom &g) : g_(g) {}
    // A call that leaves early (an error from stage(), a launch, fetch(), an allocation) would otherwise leave copies and
    // kernels in flight on the thread's buffers -- with registered host arrays even copies into the caller's memory --
    // while the next call on this thread re-packs and may free them: whatever was enqueued ends here.
    ~TileStager()
    {
        if (!pool_ || !begun_ || finished_) return;
        for (hipStream_t q : {pool_->h2d, pool_->stream, pool_->d2h})
            if (q && hipStreamSynchronize(q) != hipSuccess) (void)hipGetLastError();
    }
    TileStager(const TileStager &) = delete;
    TileStager &operator=(const TileStager &) = delete;

    // kext: extent of the k dimension in memory (1 for a 2-D array, nk_m, nk_m + 2 for the flux profiles);
    // nblk: leading blocks of a 4-D array (i,k,j,block).  target receives the chunk's device address in stage().
    void in(const void *host, int kext, int nblk, const void **target)
    {
        *target = nullptr;
        if (!host) return;
        reqs_.push_back({host, nullptr, kext, nblk, target, nullptr});
    }
    void out(void *host, int kext, int nblk, void **target)
    {
        *target = nullptr;
        if (!host) return;
        reqs_.push_back({host, host, kext, nblk, nullptr, target});
    }

    // The calling thread's streams: the kernels of the call go to stream(), the copies to its two companions.
    hipError_t open()
    {
        pool_ = &thread_stage_pool();
        StagePool &p = *pool_;
        if (p.stream) return hipSuccess;
        hipError_t e = hipStreamCreateWithFlags(&p.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&p.h2d, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&p.d2h, hipStreamNonBlocking);
        for (int i = 0; i < MAX_CHUNKS && e == hipSuccess; ++i) {
            e = hipEventCreateWithFlags(&p.ev_in[i], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&p.ev_k[i], hipEventDisableTiming);
        }
        return e;
    }
    hipStream_t stream() const { return pool_->stream; }

    // Cuts the tile into j-chunks of about `cols_per_chunk` columns (at most MAX_CHUNKS, at least one row each) and
    // sizes the buffers for the whole tile; the inputs of a chunk precede its outputs in the chunk's block.
    hipError_t begin(long cols_per_chunk)
    {
        if (!pool_) { hipError_t e = open(); if (e != hipSuccess) return e; }
        begun_ = true;
        direct_ = host_registration_enabled();
        if (direct_) {
            for (const Req &r : reqs_)
                HostRegistry::instance().ensure(r.hin, (size_t)r.nblk * g_.nj_m * r.kext * g_.ni_m * g_.rb);
            cols_per_chunk += cols_per_chunk / 2;   // every (array, chunk) is a copy of its own: fewer, larger chunks
                                                    // (measured: 6 000 columns per chunk best, 10 000 too few chunks to overlap)
        }
        std::stable_partition(reqs_.begin(), reqs_.end(), [](const Req &r) { return r.hout == nullptr; });
        long rows = cols_per_chunk / (g_.ni_t > 0 ? g_.ni_t : 1);
        if (rows < 1) rows = 1;
        int nc = (int)((g_.nj_t + rows - 1) / rows);
        if (nc > MAX_CHUNKS) nc = MAX_CHUNKS;
        if (nc < 1) nc = 1;
        chunks_.clear();
        size_t off = 64;   // [0, 64): the status words
        for (int c = 0; c < nc; ++c) {
            Chunk ch;
            ch.j0 = (int)((long)g_.nj_t * c / nc);
            ch.nj = (int)((long)g_.nj_t * (c + 1) / nc) - ch.j0;
            ch.off = off;
            ch.out_off = SIZE_MAX;
            for (const Req &r : reqs_) {
                if (r.hout && ch.out_off == SIZE_MAX) ch.out_off = off;
                ch.req_off.push_back(off);
                off += bytes(r.kext, r.nblk, ch.nj);
            }
            ch.end = off;
            chunks_.push_back(ch);
        }
        used_ = off;
        StagePool &p = *pool_;
        const size_t pin_need = direct_ ? 4096 : used_;   // (direct: the status words alone)
        if (p.pin_cap < pin_need) {
            if (p.pin) (void)hipHostFree(p.pin);
            p.pin = nullptr; p.pin_cap = 0;
            hipError_t e = hipHostMalloc(&p.pin, pin_need + pin_need / 4, hipHostMallocDefault);
            if (e != hipSuccess) return e;
            p.pin_cap = pin_need + pin_need / 4;
        }
        if (p.dev_cap < used_) {
            if (p.dev) (void)hipFree(p.dev);   // the previous call on these buffers ended with a wait for all of its work
            p.dev = nullptr; p.dev_cap = 0;
            hipError_t e = hipMalloc(&p.dev, used_ + used_ / 4);
            if (e != hipSuccess) return e;
            p.dev_cap = used_ + used_ / 4;
        }
        std::memset(p.pin, 0, 64);
        return hipMemcpyAsync(p.dev, p.pin, 64, hipMemcpyHostToDevice, p.h2d);   // in front of chunk 0 on that stream
    }
    int nchunks() const { return (int)chunks_.size(); }
    int chunk_j0(int c) const { return chunks_[c].j0; }
    int chunk_nj(int c) const { return chunks_[c].nj; }
    int *status_word(int c) const { return static_cast<int *>(pool_->dev) + c; }

    // pack the rows of chunk c (inputs, and the current contents of the outputs: INOUT semantics, the routines leave
    // night columns and unwritten fields as they were), enqueue its H2D copy, hand out its device addresses
    hipError_t stage(int c)
    {
        const Chunk &ch = chunks_[c];
        StagePool &p = *pool_;
        if (!direct_) copy_chunk(ch, true, false);
        for (size_t r = 0; r < reqs_.size(); ++r) {
            char *d = static_cast<char *>(p.dev) + ch.req_off[r];
            if (reqs_[r].in_target) *reqs_[r].in_target = d;
            if (reqs_[r].out_target) *reqs_[r].out_target = d;
        }
        hipError_t e = hipSuccess;
        if (direct_) {
            for (size_t r = 0; r < reqs_.size() && e == hipSuccess; ++r) e = direct_copy(ch, r, true, p.h2d);
        } else {
            e = hipMemcpyAsync(static_cast<char *>(p.dev) + ch.off, static_cast<char *>(p.pin) + ch.off,
                               ch.end - ch.off, hipMemcpyHostToDevice, p.h2d);
        }
        if (e == hipSuccess) e = hipEventRecord(p.ev_in[c], p.h2d);
        if (e == hipSuccess) e = hipStreamWaitEvent(p.stream, p.ev_in[c], 0);   // the chunk's kernels wait for it
        return e;
    }
    // behind the chunk's kernels: the D2H copy of its output region, on the copy-out stream
    hipError_t fetch(int c)
    {
        const Chunk &ch = chunks_[c];
        StagePool &p = *pool_;
        hipError_t e = hipEventRecord(p.ev_k[c], p.stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(p.d2h, p.ev_k[c], 0);
        if (e != hipSuccess || ch.out_off == SIZE_MAX) return e;
        if (direct_) {   // straight into the caller's tile
            for (size_t r = 0; r < reqs_.size() && e == hipSuccess; ++r)
                if (reqs_[r].hout) e = direct_copy(ch, r, false, p.d2h);
            return e;
        }
        return hipMemcpyAsync(static_cast<char *>(p.pin) + ch.out_off, static_cast<char *>(p.dev) + ch.out_off,
                              ch.end - ch.out_off, hipMemcpyDeviceToHost, p.d2h);
    }
    // status words back, one wait on the stream, outputs scattered into the caller's tile.  *status = the first
    // non-zero status word, *chunk the chunk that raised it.
    hipError_t finish(int *status, int *chunk)
    {
        StagePool &p = *pool_;
        // the copy-out stream has waited for every chunk's kernels (fetch), so its end is the call's end
        hipError_t e = hipMemcpyAsync(p.pin, p.dev, 64, hipMemcpyDeviceToHost, p.d2h);
        if (e == hipSuccess) e = hipStreamSynchronize(p.d2h);
        if (e != hipSuccess) return e;
        finished_ = true;
        *status = 0;
        if (chunk) *chunk = 0;
        for (int c = 0; c < nchunks(); ++c) {
            const int s = static_cast<const int *>(p.pin)[c];
            if (s != 0) { *status = s; if (chunk) *chunk = c; break; }
        }
        if (!direct_)
            for (const Chunk &ch : chunks_) copy_chunk(ch, false, true);
        return hipSuccess;
    }
    size_t staged_bytes() const { return used_; }

private:
    struct Req {
        const void *hin;
        void *hout;
        int kext, nblk;
        const void **in_target;
        void **out_target;
    };
    struct Chunk {
        int j0, nj;
        size_t off, out_off, end;
        std::vector<size_t> req_off;
    };
    size_t bytes(int kext, int nblk, int nj) const
    {
        const size_t b = (size_t)nblk * nj * kext * g_.ni_t * g_.rb;
        return (b + 63) & ~size_t(63);
    }
    // direct mode: the chunk's rows of request ri between the caller's array (rows of ni_t elements at a pitch of ni_m;
    // the (j, k) rows of a block follow one another) and its packed place on the device, one pitched copy per block
    hipError_t direct_copy(const Chunk &ch, size_t ri, bool to_device, hipStream_t s) const
    {
        const Req &r = reqs_[ri];
        const size_t row = (size_t)g_.ni_t * g_.rb, pitch = (size_t)g_.ni_m * g_.rb;
        const size_t height = (size_t)ch.nj * r.kext;
        char *d = static_cast<char *>(pool_->dev) + ch.req_off[ri];
        for (int b = 0; b < r.nblk; ++b) {
            const size_t h = ((((size_t)b * g_.nj_m + (g_.j0 + ch.j0)) * r.kext) * g_.ni_m + g_.i0) * g_.rb;
            char *dd = d + (size_t)b * height * row;
            hipError_t e;
            if (row == pitch) {   // the tile spans the memory's i range: one contiguous block
                if (to_device) e = hipMemcpyAsync(dd, static_cast<const char *>(r.hin) + h, height * row, hipMemcpyHostToDevice, s);
                else e = hipMemcpyAsync(static_cast<char *>(r.hout) + h, dd, height * row, hipMemcpyDeviceToHost, s);
            } else if (to_device) {
                e = hipMemcpy2DAsync(dd, row, static_cast<const char *>(r.hin) + h, pitch, row, height, hipMemcpyHostToDevice, s);
            } else {
                e = hipMemcpy2DAsync(static_cast<char *>(r.hout) + h, pitch, dd, row, row, height, hipMemcpyDeviceToHost, s);
            }
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    // one (request, row) item: the row's kext x nblk segments of ni_t elements
    void copy_row(const Chunk &ch, size_t ri, int j, bool to_pinned) const
    {
        const Req &r = reqs_[ri];
        const size_t row = (size_t)g_.ni_t * g_.rb;
        char *pk = static_cast<char *>(pool_->pin) + ch.req_off[ri];
        for (int b = 0; b < r.nblk; ++b)
            for (int k = 0; k < r.kext; ++k) {
                const size_t h = ((((size_t)b * g_.nj_m + (g_.j0 + ch.j0 + j)) * r.kext + k) * g_.ni_m + g_.i0) * g_.rb;
                const size_t q = (((size_t)b * ch.nj + j) * r.kext + k) * row;
                if (to_pinned) std::memcpy(pk + q, static_cast<const char *>(r.hin) + h, row);
                else std::memcpy(static_cast<char *>(r.hout) + h, pk + q, row);
            }
    }
    void copy_chunk(const Chunk &ch, bool to_pinned, bool outputs_only) const
    {
        size_t first = 0;
        if (outputs_only)
            while (first < reqs_.size() && !reqs_[first].hout) ++first;
        const int nreq = (int)(reqs_.size() - first), nj = ch.nj;
        if (nreq <= 0) return;
        size_t total = 0;
        for (size_t r = first; r < reqs_.size(); ++r) total += (size_t)reqs_[r].nblk * nj * reqs_[r].kext * g_.ni_t * g_.rb;
        const std::function<void(int)> item = [&](int it) { copy_row(ch, first + (size_t)(it / nj), it % nj, to_pinned); };
        if (total < (size_t(2) << 20)) { for (int it = 0; it < nreq * nj; ++it) item(it); }
        else PackPool::instance().run(nreq * nj, item);
    }
    TileGeom g_;
    std::vector<Req> reqs_;
    std::vector<Chunk> chunks_;
    size_t used_ = 0;
    StagePool *pool_ = nullptr;
    bool direct_ = false;   // RRTMG_HIP_REGISTER_HOST: pitched copies between the caller's arrays and the device
    bool begun_ = false, finished_ = false;
};

}  // namespace rrtmg
