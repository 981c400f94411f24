// Longwave entry points of the C ABI (include/rrtmg_hip.h).  No CPU fallback.
#include <hip/hip_runtime.h>

#include <algorithm>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/rrtmg_hip.h"
#include "lw_common.h"
#include "lw_kernel.h"
#include "tile_staging.h"
#include "cloud_lists.h"
#include "sw_kernel.h"   // kiss_jump_table
#include "lw_ref_tables.h"
#include "lw_state.h"

namespace rrtmg {

LwState &lw_state()
{
    static LwState s;
    return s;
}

namespace {

std::mutex g_lw_mutex;

int failw(char *err, const std::string &msg)
{
    if (err) {
        std::strncpy(err, msg.c_str(), RRTMG_HIP_ERRLEN - 1);
        err[RRTMG_HIP_ERRLEN - 1] = 0;
    }
    return 1;
}

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return failw(err, std::string("rrtmg_hip: ") + #expr + ": " + hipGetErrorString(e_)); \
    } while (0)

template <class T>
hipError_t upload(void **dptr, const std::vector<T> &v)
{
    if (*dptr) { (void)hipFree(*dptr); *dptr = nullptr; }
    hipError_t e = hipMalloc(dptr, v.size() * sizeof(T));
    if (e != hipSuccess) return e;
    return hipMemcpy(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
}

}  // namespace
}  // namespace rrtmg

using namespace rrtmg;

// NLAYERS = kme + nint(p_top*0.01/deltap) - 1 with deltap = 4 mb, LW:12861.  The reference evaluates the quotient in
// default REAL: single precision in a stock build, where a quotient next to .5 can round to the other side than in
// double precision -- so the width of the caller's REAL decides the arithmetic (nint rounds halves away from zero).
static int lw_nlayers_of(int kme, double p_top_pa, int real_bytes)
{
    if (real_bytes == 4) {
        const float q = (float)p_top_pa * 0.01f / 4.0f;
        return kme + (int)std::lroundf(q) - 1;
    }
    return kme + (int)std::lround(p_top_pa * 0.01 / 4.) - 1;
}
extern "C" int rrtmg_hip_lw_nlayers(int kme, double p_top_pa) { return lw_nlayers_of(kme, p_top_pa, 8); }
extern "C" int rrtmg_hip_lw_nlayers_real(int kme, double p_top_pa, int real_bytes)
{
    return lw_nlayers_of(kme, p_top_pa, real_bytes);
}

extern "C" int rrtmg_hip_lw_init(const char *data_path, int data_real_bytes, double cpdair, double p_top_pa, int kme,
                                 int device, char *err)
{
    std::lock_guard<std::mutex> lock(g_lw_mutex);
    LwState &st = lw_state();
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return failw(err, "rrtmg_hip_lw_init: no HIP device visible (the engine has no CPU path)");
    HIPCHK(hipSetDevice(device));
    std::string msg;
    LwHostTables host;
    if (lw_build_tables(data_path, data_real_bytes, cpdair, host, msg)) return failw(err, msg);
    HIPCHK(upload(&st.dK, host.K));
    HIPCHK(upload(&st.dExp, host.exp_tbl));
    HIPCHK(upload(&st.dTfn, host.tfn_tbl));
    HIPCHK(upload(&st.dPlk, host.totplnk));
    HIPCHK(upload(&st.dChi, host.chi_mls));
    HIPCHK(upload(&st.dG, host.gdesc));
    HIPCHK(lw_upload_constants(LWREF_PREFLOG, LWREF_TREF));
    HIPCHK(lw_upload_segments(host.gdesc.data()));
    HIPCHK(lw_upload_retab());
    {
        std::vector<double> cld((size_t)2 * LW_NBND * 58);
        lw_cloud_tables(cld.data());
        if (!st.dCld) HIPCHK(hipMalloc(&st.dCld, cld.size() * sizeof(double)));
        HIPCHK(hipMemcpy(st.dCld, cld.data(), cld.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    if (!st.dStatus) HIPCHK(hipMalloc(&st.dStatus, sizeof(int)));
    HIPCHK(hipMemset(st.dStatus, 0, sizeof(int)));
    if (!st.dPairs) HIPCHK(hipMalloc(&st.dPairs, (size_t)(LW_NTBL + 1) * 2 * sizeof(double)));
    HIPCHK(lw_fill_pairs(static_cast<double *>(st.dPairs), host.bpade));
    st.dev.K = static_cast<const double *>(st.dK);
    st.dev.exp_tbl = static_cast<const double *>(st.dExp);
    st.dev.tfn_tbl = static_cast<const double *>(st.dTfn);
    st.dev.tbl_pairs = static_cast<const double *>(st.dPairs);
    st.dev.totplnk = static_cast<const double *>(st.dPlk);
    st.dev.chi_mls = static_cast<const double *>(st.dChi);
    st.dev.gdesc = static_cast<const LwGDesc *>(st.dG);
    st.dev.bpade = host.bpade; st.dev.heatfac = host.heatfac; st.dev.fluxfac = host.fluxfac;
    st.dev.oneminus = host.oneminus;
    st.host = std::move(host);
    st.device = device;
    st.nlayers = lw_nlayers_of(kme, p_top_pa, data_real_bytes);
    st.kme = kme; st.p_top_pa = p_top_pa;
    st.ready = true;
    return 0;
}

extern "C" int rrtmg_hip_lw_reduce_tables(const char *data_path, int data_real_bytes, double cpdair, double *k_out,
                                          long long k_cap, long long *k_len, double *exp_out, double *tfn_out, char *err)
{
    std::string msg;
    LwHostTables host;
    if (lw_build_tables(data_path, data_real_bytes, cpdair, host, msg)) return failw(err, msg);
    if (k_len) *k_len = (long long)host.K.size();
    if (k_out) {
        if ((long long)host.K.size() > k_cap) return failw(err, "rrtmg_hip_lw_reduce_tables: buffer too small");
        std::memcpy(k_out, host.K.data(), host.K.size() * sizeof(double));
    }
    if (exp_out) std::memcpy(exp_out, host.exp_tbl.data(), host.exp_tbl.size() * sizeof(double));
    if (tfn_out) std::memcpy(tfn_out, host.tfn_tbl.data(), host.tfn_tbl.size() * sizeof(double));
    return 0;
}

extern "C" int rrtmg_hip_lw_table_info(const char *data_path, int data_real_bytes, int g, int *fields /*[24]*/,
                                       double *dfields /*[8]*/, char *err)
{
    std::string msg;
    static LwHostTables host;
    static std::string cached;
    if (cached != data_path) {
        host = LwHostTables{};
        if (lw_build_tables(data_path, data_real_bytes, 1004.5, host, msg)) return failw(err, msg);
        cached = data_path;
    }
    if (g < 0 || g >= LW_NGPT) return failw(err, "g out of range");
    const LwGDesc &d = host.gdesc[g];
    int f[24] = {d.band, d.ig, d.ng, d.off_self, d.off_for, d.lo.off_major, d.up.off_major, d.lo.off_frac, d.up.off_frac,
                 (int)d.lo.cfg, (int)d.up.cfg,
                 d.lo.minor[0].off, d.lo.minor[1].off, d.lo.minor[2].off, d.up.minor[0].off, d.up.minor[1].off,
                 d.lo.minor[0].nsp, d.lo.minor[1].nsp, d.lo.minor[2].nsp, d.up.minor[0].nsp, d.up.minor[1].nsp, d.stride, 0, 0};
    std::memcpy(fields, f, sizeof f);
    double df[8] = {d.lo.fracconst, d.up.fracconst, d.lo.halo_coef[0], d.lo.halo_coef[1], d.up.halo_coef[0],
                    d.up.halo_coef[1], d.up.postscale, d.delwave};
    std::memcpy(dfields, df, sizeof df);
    return 0;
}

static const char *lw_status_text(int kind)
{
    switch (kind) {
    case 1: return "ERROR: ICE GENERALIZED EFFECTIVE SIZE OUT OF BOUNDS";
    case 2: return "ERROR: SNOW GENERALIZED EFFECTIVE SIZE OUT OF BOUNDS";
    case 3: return "LIQUID EFFECTIVE RADIUS OUT OF BOUNDS";
    default: return "ERROR: Negative total lw optical depth";
    }
}

extern "C" int rrtmg_hip_lwrad(const rrtmg_hip_lwrad_args *a, char *err)
{
    LwState &st = lw_state();
    if (!st.ready) return failw(err, "rrtmg_hip_lwrad: rrtmg_hip_lw_init has not been called");
    if (a->real_bytes != 4 && a->real_bytes != 8) return failw(err, "rrtmg_hip_lwrad: real_bytes must be 4 or 8");
    if (a->icloud != 0) {
        if (a->is_cammgmp_used && !(a->lradius && a->iradius))
            return failw(err, "rrtmg_hip_lwrad: is_cammgmp_used needs lradius and iradius");
        if ((a->mp_physics == 5 || a->mp_physics == 15) && !a->qi3d)
            return failw(err, "rrtmg_hip_lwrad: FER_MP_HIRES needs qi3d");
        if (a->f_ice_phy && !a->qs3d) return failw(err, "rrtmg_hip_lwrad: the F_ICE_PHY split needs qs3d");
        if (!a->cldfra3d) return failw(err, "rrtmg_hip_lwrad: icloud /= 0 needs cldfra3d");
        if ((a->has_reqc && !a->re_cloud) || (a->has_reqi && !a->re_ice) || (a->has_reqs && !a->re_snow))
            return failw(err, "rrtmg_hip_lwrad: has_reqc/has_reqi/has_reqs set without re_cloud/re_ice/re_snow");
        if (!(a->g > 0.)) return failw(err, "rrtmg_hip_lwrad: gravity g must be positive when icloud /= 0");
    }
    const int ni_m = a->ime - a->ims + 1, nk_m = a->kme - a->kms + 1, nj_m = a->jme - a->jms + 1;
    const int ni_t = a->ite - a->its + 1, nj_t = a->jte - a->jts + 1, nk = a->kte - a->kts + 1;
    if (ni_t <= 0 || nj_t <= 0) return 0;   // an empty tile: the reference's i/j loops do not execute
    if (nk < 2 || a->its < a->ims || a->ite > a->ime || a->jts < a->jms ||
        a->jte > a->jme || a->kts < a->kms || a->kte + 1 > a->kme)
        return failw(err, "rrtmg_hip_lwrad: inconsistent tile / memory dimensions");
    // NLAYERS in the precision of THIS call's REAL (LW:12861 evaluates it in default REAL): the same number as at init
    // unless the quotient sits next to .5 and the table file's width differs from the arrays'
    const int nl = lw_nlayers_of(st.kme, st.p_top_pa, a->real_bytes);
    if (nl < nk + 1 || nl > 200) return failw(err, "rrtmg_hip_lwrad: NLAYERS from rrtmg_hip_lw_init does not fit this tile");
    if (lw_gas_lds_bytes(nl) > 160 * 1024 - 64 || lw_rt_lds_bytes(nl) > 160 * 1024 - 64)
        return failw(err, "rrtmg_hip_lwrad: column does not fit LDS");
    if (a->aer_ra_feedback == 1)
        for (int b = 0; b < 16; ++b)
            if (!a->tauaerlw[b]) return failw(err, "Warning: missing fields required for aerosol radiation");   // LW:11823-11845
    HIPCHK(hipSetDevice(st.device));
    hipStream_t stream = a->mem == 1 ? static_cast<hipStream_t>(a->stream) : nullptr;
    const size_t rb = a->real_bytes;

    LwKernelArgs k{};
    k.ims = a->ims; k.jms = a->jms; k.kms = a->kms; k.ni_m = ni_m; k.nk_m = nk_m; k.nj_m = nj_m;
    k.its = a->its; k.jts = a->jts; k.kts = a->kts; k.ni_t = ni_t; k.nj_t = nj_t; k.nk = nk;
    k.nlayers = nl;
    k.real4 = a->real_bytes == 4;
    k.aer_ra_feedback = a->aer_ra_feedback; k.clean = a->clean_atm_diag; k.o3input = a->o3input;
    k.tab = st.dev;
    k.status = static_cast<int *>(st.dStatus);
    k.icloud = a->icloud;
    if (a->icloud != 0) {
        if (nk > 128) return failw(err, "rrtmg_hip_lwrad: icloud /= 0 supports at most 128 model layers");
        if (nk < 4) return failw(err, "rrtmg_hip_lwrad: the McICA seeds need at least four layers");
        k.has_reqc = a->has_reqc; k.has_reqi = a->has_reqi; k.has_reqs = a->has_reqs;
        k.warm_rain = a->warm_rain; k.progn = a->progn; k.grav = a->g;
        k.f_ice_phy = a->f_ice_phy; k.fer_hires = (a->mp_physics == 5 || a->mp_physics == 15) ? 1 : 0;
        k.cldtab = static_cast<const double *>(st.dCld);
        // KISS jump table: NLAYERS draws per sub-column, permuteseed = 150 (LW:12555)
        std::lock_guard<std::mutex> lock(st.scratch_mutex);
        void *dj = st.dJump.count(nl) ? st.dJump[nl] : nullptr;
        if (!dj) {   // enters the map only once its contents are on the device
            std::vector<uint32_t> h((size_t)LW_NGPT * KISS_JUMP_WORDS);
            kiss_jump_table(nl, 150, LW_NGPT, h.data());
            HIPCHK(hipMalloc(&dj, h.size() * sizeof(uint32_t)));
            hipError_t e = hipMemcpy(dj, h.data(), h.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
            if (e != hipSuccess) { (void)hipFree(dj); HIPCHK(e); }
            st.dJump[nl] = dj;
        }
        k.jump = static_cast<const uint32_t *>(dj);
    }
    // mem = 0: tile-only staging through this thread's buffers, own status word (tile_staging.h)
    TileStager stg(TileGeom{ni_m, nk_m, nj_m, a->its - a->ims, a->jts - a->jms, ni_t, nj_t, rb});
    if (a->mem == 1) {
        k.emiss = a->emiss; k.tsk = a->tsk;
        k.t3d = a->t3d; k.t8w = a->t8w; k.p3d = a->p3d; k.p8w = a->p8w; k.pi3d = a->pi3d; k.qv3d = a->qv3d; k.o33d = a->o33d;
        for (int b = 0; b < 16; ++b) k.tauaerlw[b] = a->tauaerlw[b];
        k.rthratenlw = a->rthratenlw; k.glw = a->glw; k.olr = a->olr; k.lwcf = a->lwcf;
        k.lwupt = a->lwupt; k.lwuptc = a->lwuptc; k.lwuptcln = a->lwuptcln; k.lwdnt = a->lwdnt; k.lwdntc = a->lwdntc;
        k.lwdntcln = a->lwdntcln; k.lwupb = a->lwupb; k.lwupbc = a->lwupbc; k.lwupbcln = a->lwupbcln;
        k.lwdnb = a->lwdnb; k.lwdnbc = a->lwdnbc; k.lwdnbcln = a->lwdnbcln;
        k.lwupflx = a->lwupflx; k.lwupflxc = a->lwupflxc; k.lwupflxcln = a->lwupflxcln;
        k.lwdnflx = a->lwdnflx; k.lwdnflxc = a->lwdnflxc; k.lwdnflxcln = a->lwdnflxcln;
        if (a->icloud != 0) {
            k.cldfra3d = a->cldfra3d; k.qc3d = a->qc3d; k.qr3d = a->qr3d; k.qi3d = a->qi3d; k.qs3d = a->qs3d;
            k.re_cloud = a->has_reqc ? a->re_cloud : nullptr; k.re_ice = a->has_reqi ? a->re_ice : nullptr;
            k.re_snow = a->has_reqs ? a->re_snow : nullptr;
            k.qndrop3d = a->progn == 1 ? a->qndrop3d : nullptr;
            k.xland = a->xland; k.xice = a->xice; k.snow = a->snow;
            if (a->is_cammgmp_used) { k.iradius = a->iradius; k.lradius = a->lradius; }
        }
    } else {
        k.ims = a->its; k.jms = a->jts; k.ni_m = ni_t; k.nj_m = nj_t;   // packed arrays: memory = tile (k unchanged)
#define IN2(f) stg.in(a->f, 1, 1, &k.f)
#define IN3(f) stg.in(a->f, nk_m, 1, &k.f)
#define OUT2(f) stg.out(a->f, 1, 1, &k.f)
#define OUTF(f) stg.out(a->f, nk_m + 2, 1, &k.f)
        IN2(emiss); IN2(tsk);
        IN3(t3d); IN3(t8w); IN3(p3d); IN3(p8w); IN3(pi3d); IN3(qv3d); IN3(o33d);
        if (a->aer_ra_feedback == 1)
            for (int b = 0; b < 16; ++b) stg.in(a->tauaerlw[b], nk_m, 1, &k.tauaerlw[b]);
        if (a->icloud != 0) {
            IN3(cldfra3d); IN3(qc3d); IN3(qr3d); IN3(qi3d); IN3(qs3d);
            if (a->has_reqc) IN3(re_cloud);
            if (a->has_reqi) IN3(re_ice);
            if (a->has_reqs) IN3(re_snow);
            if (a->progn == 1) IN3(qndrop3d);
            IN2(xland); IN2(xice); IN2(snow);
            if (a->is_cammgmp_used) { IN3(iradius); IN3(lradius); }
        }
        stg.out(a->rthratenlw, nk_m, 1, &k.rthratenlw); OUT2(glw); OUT2(olr); OUT2(lwcf);
        OUT2(lwupt); OUT2(lwuptc); OUT2(lwuptcln); OUT2(lwdnt); OUT2(lwdntc); OUT2(lwdntcln);
        OUT2(lwupb); OUT2(lwupbc); OUT2(lwupbcln); OUT2(lwdnb); OUT2(lwdnbc); OUT2(lwdnbcln);
        OUTF(lwupflx); OUTF(lwupflxc); OUTF(lwupflxcln); OUTF(lwdnflx); OUTF(lwdnflxc); OUTF(lwdnflxcln);
        // chunks of the tile, all on this thread's own stream (tile_staging.h)
        HIPCHK(stg.begin(stage_chunk_columns()));
        stream = stg.stream();
#undef OUT2
#undef OUTF
#undef IN2
#undef IN3
    }
    if (!(a->emiss && a->tsk && a->t3d && a->t8w && a->p3d && a->p8w && a->pi3d && a->qv3d))
        return failw(err, "rrtmg_hip_lwrad: a required input array is NULL");
    if (a->lwupt && !(a->lwuptc && a->lwuptcln && a->lwdnt && a->lwdntc && a->lwdntcln && a->lwupb && a->lwupbc &&
                      a->lwupbcln && a->lwdnb && a->lwdnbc && a->lwdnbcln))
        return failw(err, "rrtmg_hip_lwrad: the twelve TOA/surface flux arrays must be passed together");
    if (a->lwupflx && !(a->lwupflxc && a->lwupflxcln && a->lwdnflx && a->lwdnflxc && a->lwdnflxcln))
        return failw(err, "rrtmg_hip_lwrad: the six flux profile arrays must be passed together");

    // scratch of this stream: as many columns per batch as RRTMG_HIP_LW_SCRATCH_MB holds (default: 16 GiB, at most a
    // quarter of the device's free memory; grow-only, smaller batches when the device cannot spare it)
    const int nchunk = a->mem == 1 ? 1 : stg.nchunks();
    int ncol = ni_t * nj_t;     // columns of the largest launch unit: the tile, or its largest chunk
    if (a->mem != 1) {
        ncol = 0;
        for (int c = 0; c < nchunk; ++c) ncol = std::max(ncol, ni_t * stg.chunk_nj(c));
    }
    double *scratch = nullptr;
    int cap_cols = 0;
    {
        size_t mb = 0;
        if (const char *e = std::getenv("RRTMG_HIP_LW_SCRATCH_MB")) {
            const long v = std::atol(e);
            if (v > 0) mb = (size_t)v;
        }
        const size_t percol = lw_scratch_doubles(nl);
        cap_cols = ncol;
        std::unique_lock<std::mutex> lock(st.scratch_mutex);
        LwState::Scratch &w = st.scratch[stream];
        if (mb || w.doubles < percol * (size_t)ncol) {
            // the buffer may have to grow: the cap from what the device has free now (asked only here)
            if (!mb) mb = default_scratch_mb(16384);
            size_t cols = (mb << 20) / (percol * sizeof(double));
            if (cols < 8) cols = 8;
            cap_cols = (size_t)ncol < cols ? ncol : (int)cols;
        }
        if (w.limit && percol * (size_t)cap_cols > w.limit) cap_cols = w.limit / percol < 8 ? 8 : (int)(w.limit / percol);
        if (w.doubles < percol * (size_t)cap_cols) {
            if (w.p) { HIPCHK(hipStreamSynchronize(stream)); HIPCHK(hipFree(w.p)); w.p = nullptr; w.doubles = 0; }
            void *p = nullptr;
            hipError_t e = hipMalloc(&p, percol * (size_t)cap_cols * sizeof(double));
            while (e == hipErrorOutOfMemory && cap_cols > 8) {   // halve the batch until it fits (down to 8 columns)
                (void)hipGetLastError();
                cap_cols = cap_cols / 2 < 8 ? 8 : cap_cols / 2;
                w.limit = percol * (size_t)cap_cols;
                e = hipMalloc(&p, percol * (size_t)cap_cols * sizeof(double));
            }
            HIPCHK(e);
            w.p = static_cast<double *>(p);
            w.doubles = percol * (size_t)cap_cols;
        }
        scratch = w.p;
        // cloud-free and cloudy columns to their own kernel instantiations (cloud_lists.h)
        const bool lists_on = k.icloud != 0 && cloud_lists_enabled();
        if (lists_on && w.lists_ints < 2 * (size_t)ncol + 8) {
            if (w.lists) { HIPCHK(hipStreamSynchronize(stream)); HIPCHK(hipFree(w.lists)); w.lists = nullptr; w.lists_ints = 0; }
            void *p = nullptr;
            HIPCHK(hipMalloc(&p, (2 * (size_t)ncol + 8) * sizeof(int)));
            w.lists = static_cast<int *>(p);
            w.lists_ints = 2 * (size_t)ncol + 8;
        }
        int *const lists = w.lists;
        // the gas and transfer kernels of one call hand their batch over through this buffer.  mem = 1: host threads
        // that share the caller's stream enqueue their kernel sequences one call at a time, under the lock.  mem = 0:
        // stream and scratch belong to this thread alone; the lock is dropped so that concurrent tiles pack and
        // enqueue side by side.
        if (a->mem != 1) lock.unlock();
        for (int c = 0; c < nchunk; ++c) {
            int ncol_c = ni_t * nj_t;
            if (a->mem != 1) {
                HIPCHK(stg.stage(c));          // pack + H2D of this chunk; sets the array pointers of k
                k.jms = k.jts = a->jts + stg.chunk_j0(c);
                k.nj_m = k.nj_t = stg.chunk_nj(c);
                k.status = stg.status_word(c);
                ncol_c = ni_t * k.nj_t;
            }
            if (lists_on) {
                int *const counts = lists + 2 * (size_t)ncol;
                HIPCHK(cloud_lists_launch(k.cldfra3d, k.real4, k.ni_m, k.nk_m, k.its - k.ims, k.jts - k.jms, k.kts - k.kms,
                                          k.ni_t, k.nj_t, k.nk, lists, counts, stream));
                for (int pass = 0; pass < 2; ++pass) {   // 0: the cloud-free columns without McICA, 1: the cloudy ones
                    LwKernelArgs kp = k;
                    if (pass == 0) kp.icloud = 0;
                    kp.colmap = lists + (size_t)pass * ncol_c;
                    kp.colcount = counts + pass;
                    HIPCHK(lw_launch(kp, ncol_c, scratch, cap_cols, stream));
                }
            } else
            HIPCHK(lw_launch(k, ncol_c, scratch, cap_cols, stream));
            if (a->mem != 1) HIPCHK(stg.fetch(c));
        }
    }

    if (a->mem != 1) {
        int status = 0, chunk = 0;
        HIPCHK(stg.finish(&status, &chunk));
        if (status != 0) {
            const int col = (status & 0x0fffffff) - 1;   // column of the chunk that raised it
            char m[RRTMG_HIP_ERRLEN];
            std::snprintf(m, sizeof m, "%s at point i,j=%5d%5d", lw_status_text(status >> 28),
                          a->its + col % ni_t, a->jts + stg.chunk_j0(chunk) + col / ni_t);   // LW:12613-12615, LW:2831-2836
            return failw(err, m);
        }
    }
    return 0;
}

extern "C" int rrtmg_hip_lw_status(char *err)
{
    LwState &st = lw_state();
    if (!st.ready) return failw(err, "rrtmg_hip_lw_status: rrtmg_hip_lw_init has not been called");
    HIPCHK(hipSetDevice(st.device));
    HIPCHK(hipDeviceSynchronize());
    int status = 0;
    HIPCHK(hipMemcpy(&status, st.dStatus, sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(st.dStatus, 0, sizeof(int)));
    if (status != 0) {
        char m[RRTMG_HIP_ERRLEN];
        std::snprintf(m, sizeof m, "%s at tile column %d", lw_status_text(status >> 28), (status & 0x0fffffff) - 1);
        return failw(err, m);
    }
    return 0;
}

extern "C" int rrtmg_hip_lw_finalize(void)
{
    LwState &st = lw_state();
    void **ps[] = {&st.dK, &st.dExp, &st.dTfn, &st.dPlk, &st.dChi, &st.dG, &st.dStatus, &st.dCld, &st.dPairs};
    for (auto &kv : st.dJump) if (kv.second) (void)hipFree(kv.second);
    st.dJump.clear();
    for (void **p : ps) { if (*p) (void)hipFree(*p); *p = nullptr; }
    {
        std::lock_guard<std::mutex> lock(st.scratch_mutex);
        for (auto &kv : st.scratch) {
            if (kv.second.p) (void)hipFree(kv.second.p);
            if (kv.second.lists) (void)hipFree(kv.second.lists);
        }
        st.scratch.clear();
    }
    st.ready = false;
    stage_pools_release_if_idle();
    return 0;
}
