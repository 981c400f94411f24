// C ABI of the engine (include/rrtmg_hip.h).  No CPU fallback: every entry
// point fails loudly if the device or the HIP path is unavailable.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/rrtmg_hip.h"
#include "rrtmg_common.h"
#include "rrtmg_state.h"
#include "sw_kernel.h"
#include "tile_staging.h"
#include "cloud_lists.h"
#include "lw_state.h"

namespace rrtmg {

SwState &sw_state()
{
    static SwState s;
    return s;
}

// The staging buffers and streams of the host threads (tile_staging.h) serve all three entry-point families: they are
// released by whichever finalize comes last.
bool optics_ready();   // c_abi_optics.hip
void stage_pools_release_if_idle()
{
    if (!sw_state().ready && !lw_state().ready && !optics_ready()) stage_pools_release();
}

namespace {

std::mutex g_init_mutex;

int fail(char *err, const std::string &msg)
{
    if (err) {
        std::strncpy(err, msg.c_str(), RRTMG_HIP_ERRLEN - 1);
        err[RRTMG_HIP_ERRLEN - 1] = 0;
    }
    return 1;
}

#define HIPCHK(expr)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(err, std::string("rrtmg_hip: ") + #expr + ": " + hipGetErrorString(e_)); \
    } while (0)

}  // namespace
}  // namespace rrtmg

using namespace rrtmg;

extern "C" const char *rrtmg_hip_version(void) { return "rrtmg_hip 0.1 gfx950"; }

// rank -> device map for an MPI job with one rank per GPU (the reference runs one RRTMG per MPI patch)
extern "C" int rrtmg_hip_pick_device(void)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return 0;
    const char *vars[] = {"RRTMG_HIP_DEVICE", "OMPI_COMM_WORLD_LOCAL_RANK", "MV2_COMM_WORLD_LOCAL_RANK",
                          "MPI_LOCALRANKID", "SLURM_LOCALID", "LOCAL_RANK"};
    for (const char *v : vars) {
        const char *e = std::getenv(v);
        if (e && *e) return std::atoi(e) % ndev;
    }
    return 0;
}

extern "C" int rrtmg_hip_sw_init(const char *data_path, int data_real_bytes, double cpdair, int device, char *err)
{
    std::lock_guard<std::mutex> lock(g_init_mutex);
    SwState &st = sw_state();
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(err, "rrtmg_hip_sw_init: no HIP device visible (the engine has no CPU path)");
    HIPCHK(hipSetDevice(device));
    std::string msg;
    SwHostTables host;
    if (sw_build_tables(data_path, data_real_bytes, cpdair, host, msg)) return fail(err, msg);
    if (st.dK) { (void)hipFree(st.dK); (void)hipFree(st.dExp); (void)hipFree(st.dG); st.dK = st.dExp = st.dG = nullptr; }
    HIPCHK(hipMalloc(&st.dK, host.K.size() * sizeof(double)));
    HIPCHK(hipMalloc(&st.dExp, host.exp_tbl.size() * sizeof(double)));
    HIPCHK(hipMalloc(&st.dG, host.gdesc.size() * sizeof(GDesc)));
    HIPCHK(hipMemcpy(st.dK, host.K.data(), host.K.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(st.dExp, host.exp_tbl.data(), host.exp_tbl.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(st.dG, host.gdesc.data(), host.gdesc.size() * sizeof(GDesc), hipMemcpyHostToDevice));
    HIPCHK(sw_upload_constants(nullptr));
    HIPCHK(sw_upload_retab());
    {
        std::vector<double> cld((size_t)CLD_NTAB * SW_NBND * CLD_STRIDE);
        sw_cloud_tables(cld.data());
        if (!st.dCld) HIPCHK(hipMalloc(&st.dCld, cld.size() * sizeof(double)));
        HIPCHK(hipMemcpy(st.dCld, cld.data(), cld.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    if (!st.dStatus) HIPCHK(hipMalloc(&st.dStatus, sizeof(int)));
    HIPCHK(hipMemset(st.dStatus, 0, sizeof(int)));
    st.dev.K = static_cast<const double *>(st.dK);
    st.dev.exp_tbl = static_cast<const double *>(st.dExp);
    st.dev.gdesc = static_cast<const GDesc *>(st.dG);
    st.dev.bpade = host.bpade;
    st.dev.heatfac = host.heatfac;
    st.dev.oneminus = host.oneminus;
    st.host = std::move(host);
    st.device = device;
    st.ready = true;
    return 0;
}

// host-side reduced tables for parity tests of the init path (no device needed)
extern "C" int rrtmg_hip_sw_reduce_tables(const char *data_path, int data_real_bytes, double cpdair,
                                          double *k_out, long long k_cap, long long *k_len,
                                          double *exp_out, char *err)
{
    std::string msg;
    SwHostTables host;
    if (sw_build_tables(data_path, data_real_bytes, cpdair, host, msg)) return fail(err, msg);
    if (k_len) *k_len = (long long)host.K.size();
    if (k_out) {
        if ((long long)host.K.size() > k_cap) return fail(err, "rrtmg_hip_sw_reduce_tables: buffer too small");
        std::memcpy(k_out, host.K.data(), host.K.size() * sizeof(double));
    }
    if (exp_out) std::memcpy(exp_out, host.exp_tbl.data(), host.exp_tbl.size() * sizeof(double));
    return 0;
}

// table lookups by (band, name) into the packed layout, for the same tests
extern "C" int rrtmg_hip_sw_table_info(const char *data_path, int data_real_bytes, int g, int *fields /*[16]*/,
                                       double *dfields /*[10]*/, char *err)
{
    std::string msg;
    static SwHostTables host;
    static std::string cached;
    if (cached != data_path) {
        host = SwHostTables{};
        if (sw_build_tables(data_path, data_real_bytes, 1004.5, host, msg)) return fail(err, msg);
        cached = data_path;
    }
    if (g < 0 || g >= SW_NGPT) return fail(err, "g out of range");
    const GDesc &d = host.gdesc[g];
    int f[16] = {d.band, d.ig, d.ng, d.lo.off_major, d.up.off_major, d.off_self, d.off_for, d.off_sflux,
                 d.off_rayla, d.nsf, d.layreffr, d.solar_kind, d.uvband, (int)d.lo.cfg, (int)d.up.cfg, 0};
    std::memcpy(fields, f, sizeof f);
    double df[10] = {d.lo.strrat, d.up.strrat, d.lo.mfac, d.up.mfac, d.lo.minorabs, d.up.minorabs,
                     d.lo.rayl, d.up.rayl, d.scalekur, 0};
    std::memcpy(dfields, df, sizeof df);
    return 0;
}

static const char *sw_status_text(int kind)
{
    switch (kind) {
    case 1: return "ERROR: ICE GENERALIZED EFFECTIVE SIZE OUT OF BOUNDS";
    case 2: return "ERROR: SNOW GENERALIZED EFFECTIVE SIZE OUT OF BOUNDS";
    case 3: return "liquid effective radius out of bounds";
    default: return "ERROR: Negative total optical depth";
    }
}

extern "C" long long rrtmg_hip_swrad_staging_bytes(const rrtmg_hip_swrad_args *a)
{
    // what one mem = 0 call moves over PCIe: the tile of every array (tile_staging.h), inputs one way, outputs both
    const long long ni = a->ite - a->its + 1, nk = a->kme - a->kms + 1, nj = a->jte - a->jts + 1;
    if (ni <= 0 || nj <= 0) return 0;
    const long long n2 = ni * nj * a->real_bytes, n3 = ni * nk * nj * a->real_bytes;
    return 3 * n2 + 6 * n3 + (a->aer_ra_feedback ? 8 * n3 : 0) + 2 * (n3 + 22 * n2);
}

extern "C" int rrtmg_hip_swrad(const rrtmg_hip_swrad_args *a, char *err)
{
    SwState &st = sw_state();
    if (!st.ready) return fail(err, "rrtmg_hip_swrad: rrtmg_hip_sw_init has not been called");
    if (a->real_bytes != 4 && a->real_bytes != 8) return fail(err, "rrtmg_hip_swrad: real_bytes must be 4 or 8");
    if (a->icloud != 0) {
        if (a->is_cammgmp_used && !(a->lradius && a->iradius))
            return fail(err, "rrtmg_hip_swrad: is_cammgmp_used needs lradius and iradius");
        if ((a->mp_physics == 5 || a->mp_physics == 15) && !a->qi3d)
            return fail(err, "rrtmg_hip_swrad: FER_MP_HIRES needs qi3d");
        if (a->f_ice_phy && !a->qs3d) return fail(err, "rrtmg_hip_swrad: the F_ICE_PHY split needs qs3d");
        if (!a->cldfra3d) return fail(err, "rrtmg_hip_swrad: icloud /= 0 needs cldfra3d");
        if ((a->has_reqc && !a->re_cloud) || (a->has_reqi && !a->re_ice) || (a->has_reqs && !a->re_snow))
            return fail(err, "rrtmg_hip_swrad: has_reqc/has_reqi/has_reqs set without re_cloud/re_ice/re_snow");
        if (!(a->g > 0.)) return fail(err, "rrtmg_hip_swrad: gravity g must be positive when icloud /= 0");
    }
    const int ni_m = a->ime - a->ims + 1, nk_m = a->kme - a->kms + 1, nj_m = a->jme - a->jms + 1;
    const int ni_t = a->ite - a->its + 1, nj_t = a->jte - a->jts + 1, nk = a->kte - a->kts + 1;
    if (ni_t <= 0 || nj_t <= 0) return 0;   // an empty tile: the reference's i/j loops do not execute
    if (nk < 2 || a->its < a->ims || a->ite > a->ime || a->jts < a->jms ||
        a->jte > a->jme || a->kts < a->kms || a->kte + 1 > a->kme)
        return fail(err, "rrtmg_hip_swrad: inconsistent tile / memory dimensions");
    if (nk + 1 > 128) return fail(err, "rrtmg_hip_swrad: more than 127 model layers are not supported");
    if (a->aer_ra_feedback == 1 &&
        !(a->tauaer300 && a->tauaer400 && a->tauaer600 && a->tauaer999 && a->gaer300 && a->gaer400 && a->gaer600 &&
          a->gaer999 && a->waer300 && a->waer400 && a->waer600 && a->waer999))
        return fail(err, "Warning: missing fields required for aerosol radiation");   // SW:10287-10305
    if ((a->alswvisdir || a->alswvisdif || a->alswnirdir || a->alswnirdif) &&
        !(a->alswvisdir && a->alswvisdif && a->alswnirdir && a->alswnirdif && a->xland))
        return fail(err, "rrtmg_hip_swrad: the four SSiB albedos and xland must be passed together");
    if (a->tauaer3d_sw && !(a->ssaaer3d_sw && a->asyaer3d_sw))
        return fail(err, "rrtmg_hip_swrad: tauaer3d_sw needs ssaaer3d_sw and asyaer3d_sw");
    if (sw_lds_bytes(nk, a->icloud != 0, a->swupflx != nullptr, !a->tauaer3d_sw && !a->aerod) > 160 * 1024) return fail(err, "rrtmg_hip_swrad: column does not fit LDS");
    HIPCHK(hipSetDevice(st.device));

    hipStream_t stream = a->mem == 1 ? static_cast<hipStream_t>(a->stream) : nullptr;
    const size_t rb = a->real_bytes;

    SwKernelArgs k{};
    k.ims = a->ims; k.jms = a->jms; k.kms = a->kms; k.ni_m = ni_m; k.nk_m = nk_m; k.nj_m = nj_m;
    k.its = a->its; k.jts = a->jts; k.kts = a->kts; k.ni_t = ni_t; k.nj_t = nj_t; k.nk = nk;
    k.real4 = a->real_bytes == 4;
    k.aer_ra_feedback = a->aer_ra_feedback; k.clean = a->clean_atm_diag; k.o3input = a->o3input;
    k.solcon = a->solcon;
    k.tab = st.dev;
    k.icloud = a->icloud;
    if (a->icloud != 0) {
        if (nk + 1 < 4) return fail(err, "rrtmg_hip_swrad: the McICA seeds need at least four layers");
        k.has_reqc = a->has_reqc; k.has_reqi = a->has_reqi; k.has_reqs = a->has_reqs;
        k.warm_rain = a->warm_rain; k.progn = a->progn; k.grav = a->g;
        k.f_ice_phy = a->f_ice_phy; k.fer_hires = (a->mp_physics == 5 || a->mp_physics == 15) ? 1 : 0;
        k.cldtab = static_cast<const double *>(st.dCld);
        // KISS jump table of this layer count (permuteseed = 1, SW:10952), built once per nlay
        std::lock_guard<std::mutex> lock(st.jump_mutex);
        void *dj = st.dJump.count(nk + 1) ? st.dJump[nk + 1] : nullptr;
        if (!dj) {   // enters the map only once its contents are on the device
            std::vector<uint32_t> h((size_t)SW_NGPT * KISS_JUMP_WORDS);
            kiss_jump_table(nk + 1, 1, SW_NGPT, h.data());
            HIPCHK(hipMalloc(&dj, h.size() * sizeof(uint32_t)));
            hipError_t e = hipMemcpy(dj, h.data(), h.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
            if (e != hipSuccess) { (void)hipFree(dj); HIPCHK(e); }
            st.dJump[nk + 1] = dj;
        }
        k.jump = static_cast<const uint32_t *>(dj);
    }

    // mem = 0: the tile of every array is packed, staged and unpacked by this thread's stager (tile_staging.h) and the
    // kernels see arrays whose memory dimensions are the tile's; the call has its own status word
    TileStager stg(TileGeom{ni_m, nk_m, nj_m, a->its - a->ims, a->jts - a->jms, ni_t, nj_t, rb});
    k.status = static_cast<int *>(st.dStatus);
    if (a->mem == 1) {
        k.xcoszen = a->xcoszen; k.albedo = a->albedo; k.tsk = a->tsk;
        k.t3d = a->t3d; k.t8w = a->t8w; k.p3d = a->p3d; k.p8w = a->p8w; k.pi3d = a->pi3d; k.qv3d = a->qv3d;
        k.o33d = a->o33d;
        k.tauaer300 = a->tauaer300; k.tauaer400 = a->tauaer400; k.tauaer600 = a->tauaer600; k.tauaer999 = a->tauaer999;
        k.gaer400 = a->gaer400; k.gaer600 = a->gaer600; k.waer400 = a->waer400; k.waer600 = a->waer600;
        k.rthratensw = a->rthratensw; k.coszr = a->coszr; k.gsw = a->gsw; k.swcf = a->swcf;
        k.swupt = a->swupt; k.swuptc = a->swuptc; k.swuptcln = a->swuptcln; k.swdnt = a->swdnt;
        k.swdntc = a->swdntc; k.swdntcln = a->swdntcln; k.swupb = a->swupb; k.swupbc = a->swupbc;
        k.swupbcln = a->swupbcln; k.swdnb = a->swdnb; k.swdnbc = a->swdnbc; k.swdnbcln = a->swdnbcln;
        k.swvisdir = a->swvisdir; k.swvisdif = a->swvisdif; k.swnirdir = a->swnirdir; k.swnirdif = a->swnirdif;
        k.swddir = a->swddir; k.swddni = a->swddni; k.swddif = a->swddif;
        k.swupflx = a->swupflx; k.swupflxc = a->swupflxc; k.swupflxcln = a->swupflxcln;
        k.swdnflx = a->swdnflx; k.swdnflxc = a->swdnflxc; k.swdnflxcln = a->swdnflxcln;
        k.tauaer3d_sw = a->tauaer3d_sw; k.ssaaer3d_sw = a->ssaaer3d_sw; k.asyaer3d_sw = a->asyaer3d_sw;
        k.aerod = a->aerod;
        if (a->alswvisdir) {
            k.alswvisdir = a->alswvisdir; k.alswvisdif = a->alswvisdif; k.alswnirdir = a->alswnirdir;
            k.alswnirdif = a->alswnirdif; k.xland = a->xland;
        }
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
        IN2(xcoszen); IN2(albedo); IN2(tsk);
        IN3(t3d); IN3(t8w); IN3(p3d); IN3(p8w); IN3(pi3d); IN3(qv3d); IN3(o33d);
        if (a->aer_ra_feedback == 1) {
            IN3(tauaer300); IN3(tauaer400); IN3(tauaer600); IN3(tauaer999);
            IN3(gaer400); IN3(gaer600); IN3(waer400); IN3(waer600);
        }
        stg.in(a->tauaer3d_sw, nk_m, SW_NBND, &k.tauaer3d_sw);
        stg.in(a->ssaaer3d_sw, nk_m, SW_NBND, &k.ssaaer3d_sw);
        stg.in(a->asyaer3d_sw, nk_m, SW_NBND, &k.asyaer3d_sw);
        stg.in(a->aerod, nk_m, 6, &k.aerod);
        if (a->alswvisdir) {
            IN2(alswvisdir); IN2(alswvisdif); IN2(alswnirdir); IN2(alswnirdif);
            if (a->icloud == 0) IN2(xland);
        }
        if (a->icloud != 0) {
            IN3(cldfra3d); IN3(qc3d); IN3(qr3d); IN3(qi3d); IN3(qs3d);
            if (a->has_reqc) IN3(re_cloud);
            if (a->has_reqi) IN3(re_ice);
            if (a->has_reqs) IN3(re_snow);
            if (a->progn == 1) IN3(qndrop3d);
            IN2(xland); IN2(xice); IN2(snow);
            if (a->is_cammgmp_used) { IN3(iradius); IN3(lradius); }
        }
        stg.out(a->rthratensw, nk_m, 1, &k.rthratensw); OUT2(coszr); OUT2(gsw); OUT2(swcf);
        OUT2(swupt); OUT2(swuptc); OUT2(swuptcln); OUT2(swdnt); OUT2(swdntc); OUT2(swdntcln);
        OUT2(swupb); OUT2(swupbc); OUT2(swupbcln); OUT2(swdnb); OUT2(swdnbc); OUT2(swdnbcln);
        OUT2(swvisdir); OUT2(swvisdif); OUT2(swnirdir); OUT2(swnirdif);
        OUT2(swddir); OUT2(swddni); OUT2(swddif);
        stg.out(a->swupflx, nk_m + 2, 1, &k.swupflx); stg.out(a->swupflxc, nk_m + 2, 1, &k.swupflxc);
        stg.out(a->swupflxcln, nk_m + 2, 1, &k.swupflxcln); stg.out(a->swdnflx, nk_m + 2, 1, &k.swdnflx);
        stg.out(a->swdnflxc, nk_m + 2, 1, &k.swdnflxc); stg.out(a->swdnflxcln, nk_m + 2, 1, &k.swdnflxcln);
        // chunks of the tile, all on this thread's own stream (tile_staging.h)
        HIPCHK(stg.begin(stage_chunk_columns()));
        stream = stg.stream();
#undef OUT2
#undef IN2
#undef IN3
    }
    if (!(a->xcoszen && a->albedo && a->t3d && a->t8w && a->p3d && a->p8w && a->pi3d && a->qv3d))
        return fail(err, "rrtmg_hip_swrad: a required input array is NULL");
    if (a->swupt && !(a->swuptc && a->swuptcln && a->swdnt && a->swdntc && a->swdntcln && a->swupb && a->swupbc &&
                      a->swupbcln && a->swdnb && a->swdnbc && a->swdnbcln))
        return fail(err, "rrtmg_hip_swrad: the twelve TOA/surface flux arrays must be passed together");

    // scratch of this stream: as many columns per launch as RRTMG_HIP_SW_SCRATCH_MB holds (default: 32 GiB, at most a
    // quarter of the device's free memory; the buffer only grows to what the largest tile so far needed, and a
    // device that cannot spare it gets smaller batches)
    {
        const int nchunk = a->mem == 1 ? 1 : stg.nchunks();
        int ncol_all = ni_t * nj_t;     // columns of the largest launch unit: the tile, or its largest chunk
        if (a->mem != 1) {
            ncol_all = 0;
            for (int c = 0; c < nchunk; ++c) ncol_all = std::max(ncol_all, ni_t * stg.chunk_nj(c));
        }
        size_t mb = 0;
        if (const char *e = std::getenv("RRTMG_HIP_SW_SCRATCH_MB")) {
            const long v = std::atol(e);
            if (v > 0) mb = (size_t)v;
        }
        const size_t percol = sw_scratch_doubles_per_column(k.nk);
        int cap_cols = ncol_all;
        {
            std::unique_lock<std::mutex> lock(st.jump_mutex);
            SwState::Scratch &w = st.scratch[stream];
            if (mb || w.doubles < percol * (size_t)ncol_all) {
                // the buffer may have to grow: the cap from what the device has free now (asked only here)
                if (!mb) mb = default_scratch_mb(32768);
                size_t cols = (mb << 20) / (percol * sizeof(double));
                if (cols < 8) cols = 8;
                cap_cols = (size_t)ncol_all < cols ? ncol_all : (int)cols;
            }
            if (w.limit && percol * (size_t)cap_cols > w.limit) cap_cols = w.limit / percol < 8 ? 8 : (int)(w.limit / percol);
            if (w.doubles < percol * (size_t)cap_cols) {
                if (w.p) { HIPCHK(hipStreamSynchronize(stream)); HIPCHK(hipFree(w.p)); w.p = nullptr; w.doubles = 0; }
                // when the device cannot spare that much, halve the batch until it fits (down to 8 columns)
                void *p = nullptr;
                hipError_t e = hipMalloc(&p, percol * (size_t)cap_cols * sizeof(double));
                while (e == hipErrorOutOfMemory && cap_cols > 8) {
                    (void)hipGetLastError();
                    cap_cols = cap_cols / 2 < 8 ? 8 : cap_cols / 2;
                    w.limit = percol * (size_t)cap_cols;
                    e = hipMalloc(&p, percol * (size_t)cap_cols * sizeof(double));
                }
                HIPCHK(e);
                w.p = static_cast<double *>(p);
                w.doubles = percol * (size_t)cap_cols;
            } else if (w.doubles / percol < (size_t)cap_cols) {
                cap_cols = (int)(w.doubles / percol);
            }
            k.scratch = w.p;
            k.scratch_stride = percol;
            // cloud-free and cloudy columns to their own kernel instantiations (cloud_lists.h): two device lists per
            // launch unit, grow-only like the scratch
            const bool lists_on = k.icloud != 0 && cloud_lists_enabled();
            if (lists_on && w.lists_ints < 2 * (size_t)ncol_all + 8) {
                if (w.lists) { HIPCHK(hipStreamSynchronize(stream)); HIPCHK(hipFree(w.lists)); w.lists = nullptr; w.lists_ints = 0; }
                void *p = nullptr;
                HIPCHK(hipMalloc(&p, (2 * (size_t)ncol_all + 8) * sizeof(int)));
                w.lists = static_cast<int *>(p);
                w.lists_ints = 2 * (size_t)ncol_all + 8;
            }
            int *const lists = w.lists;
            // mem = 1: launched under the lock -- another host thread on the caller's stream may not resize the buffer
            // in between.  mem = 0: the stream and its scratch belong to this thread alone; the lock is dropped so
            // that concurrent tiles pack and enqueue side by side.
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
                    int *const counts = lists + 2 * (size_t)ncol_all;
                    HIPCHK(cloud_lists_launch(k.cldfra3d, k.real4, k.ni_m, k.nk_m, k.its - k.ims, k.jts - k.jms, k.kts - k.kms,
                                              k.ni_t, k.nj_t, k.nk, lists, counts, stream));
                    for (int pass = 0; pass < 2; ++pass) {   // 0: the cloud-free columns without McICA, 1: the cloudy ones
                        SwKernelArgs kp = k;
                        if (pass == 0) kp.icloud = 0;
                        kp.colmap = lists + (size_t)pass * ncol_c;
                        kp.colcount = counts + pass;
                        for (int c0 = 0; c0 < ncol_c; c0 += cap_cols) {
                            kp.col0 = c0;
                            kp.ncol = ncol_c - c0 < cap_cols ? ncol_c - c0 : cap_cols;
                            HIPCHK(sw_launch(kp, stream));
                        }
                    }
                } else
                for (int c0 = 0; c0 < ncol_c; c0 += cap_cols) {
                    k.col0 = c0;
                    k.ncol = ncol_c - c0 < cap_cols ? ncol_c - c0 : cap_cols;
                    HIPCHK(sw_launch(k, stream));
                }
                if (a->mem != 1) HIPCHK(stg.fetch(c));
            }
        }
    }

    if (a->mem != 1) {
        int status = 0, chunk = 0;
        HIPCHK(stg.finish(&status, &chunk));
        if (status != 0) {
            const int col = (status & 0x0fffffff) - 1;   // column of the chunk that raised it
            char m[RRTMG_HIP_ERRLEN];
            std::snprintf(m, sizeof m, "%s at point i,j=%5d%5d", sw_status_text(status >> 28),
                          a->its + col % ni_t, a->jts + stg.chunk_j0(chunk) + col / ni_t);   // SW:11031-11033, SW:2169-2173
            return fail(err, m);
        }
    }
    // mem = 1: nothing synchronises here; rrtmg_hip_sw_status() reports a fatal raised on the device
    return 0;
}

extern "C" int rrtmg_hip_sw_status(char *err)
{
    SwState &st = sw_state();
    if (!st.ready) return fail(err, "rrtmg_hip_sw_status: rrtmg_hip_sw_init has not been called");
    HIPCHK(hipSetDevice(st.device));
    HIPCHK(hipDeviceSynchronize());
    int status = 0;
    HIPCHK(hipMemcpy(&status, st.dStatus, sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(st.dStatus, 0, sizeof(int)));
    if (status != 0) {
        char m[RRTMG_HIP_ERRLEN];
        std::snprintf(m, sizeof m, "%s at tile column %d", sw_status_text(status >> 28), (status & 0x0fffffff) - 1);
        return fail(err, m);
    }
    return 0;
}


// RRTMG_HIP_REGISTER_HOST=1 pins the caller's arrays where they are and remembers them by base address; an array
// that the host is about to free is dropped from that registry here (no-op for an address that is not in it).
extern "C" int rrtmg_hip_unregister_host(const void *base)
{
    if (base) HostRegistry::instance().release(base);
    return 0;
}

extern "C" int rrtmg_hip_finalize(void)
{
    SwState &st = sw_state();
    if (st.dK) { (void)hipFree(st.dK); (void)hipFree(st.dExp); (void)hipFree(st.dG); }
    if (st.dStatus) (void)hipFree(st.dStatus);
    if (st.dCld) (void)hipFree(st.dCld);
    for (auto &kv : st.dJump) if (kv.second) (void)hipFree(kv.second);
    st.dJump.clear();
    for (auto &kv : st.scratch) {
        if (kv.second.p) (void)hipFree(kv.second.p);
        if (kv.second.lists) (void)hipFree(kv.second.lists);
    }
    st.scratch.clear();
    st.dK = st.dExp = st.dG = st.dStatus = st.dCld = nullptr;
    st.ready = false;
    stage_pools_release_if_idle();
    return 0;
}
