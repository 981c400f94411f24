// C ABI of the aerosol optical-averaging stage (include/rrtmg_hip.h).
#include <hip/hip_runtime.h>

#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/rrtmg_hip.h"
#include "optics_common.h"
#include "optics_kernel.h"
#include "tile_staging.h"

namespace rrtmg {
namespace {

struct OpticsState {
    bool ready = false;
    int device = 0;
    int precision = 0;       // 0 fp64, 1 mixed (rrtmg_hip_optics_precision)
    OpticsTables tab{};
};
OpticsState g_opt;
std::mutex g_opt_mutex;

int failo(char *err, const std::string &msg)
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
            return failo(err, std::string("rrtmg_hip: ") + #expr + ": " + hipGetErrorString(e_)); \
    } while (0)

int fill_tables(OpticsTables &t, int nbin, double dlo_um, double dhi_um, const double *refindex, const double *density,
                std::string &msg)
{
    if (nbin < 1 || nbin > OPT_MAXBIN) { msg = "rrtmg_hip_optics_init: nbin must be 1..8"; return 1; }
    if (!(dlo_um > 0. && dhi_um > dlo_um)) { msg = "rrtmg_hip_optics_init: need 0 < dlo_um < dhi_um"; return 1; }
    optics_default_tables(t, nbin, dlo_um, dhi_um);
    if (refindex)
        for (int s = 0; s < OPT_NSPEC; ++s)
            for (int w = 0; w < OPT_NWAVE; ++w) {
                const double re = refindex[(s * OPT_NWAVE + w) * 2], im = refindex[(s * OPT_NWAVE + w) * 2 + 1];
                if (!(re > 0.) || im < 0.) { msg = "rrtmg_hip_optics_init: refractive index needs real > 0, imaginary >= 0"; return 1; }
                t.refr[s][w] = re; t.refi[s][w] = im;
            }
    if (density)
        for (int s = 0; s < OPT_NSPEC; ++s) {
            if (!(density[s] > 0.)) { msg = "rrtmg_hip_optics_init: density must be positive"; return 1; }
            t.dens[s] = density[s];
        }
    for (int s = 0; s < OPT_NSPEC; ++s) t.rdens[s] = 1.0 / t.dens[s];
    return 0;
}

}  // namespace

// for modal_sections.hip
bool optics_ready() { return g_opt.ready; }
int optics_device() { return g_opt.device; }
const OpticsTables &optics_tables() { return g_opt.tab; }
}  // namespace rrtmg

using namespace rrtmg;

extern "C" int rrtmg_hip_optics_init(int nbin, double dlo_um, double dhi_um, const double *refindex,
                                     const double *density, int device, char *err)
{
    std::lock_guard<std::mutex> lock(g_opt_mutex);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return failo(err, "rrtmg_hip_optics_init: no HIP device visible (the engine has no CPU path)");
    std::string msg;
    OpticsTables t{};
    if (fill_tables(t, nbin, dlo_um, dhi_um, refindex, density, msg)) return failo(err, msg);
    HIPCHK(hipSetDevice(device));
    HIPCHK(optics_upload_tables(t));
    g_opt.tab = t;
    g_opt.device = device;
    g_opt.ready = true;
    return 0;
}

extern "C" int rrtmg_hip_optics_table_info(double *wavelength_um20, double *refindex_9x20x2, double *density9)
{
    OpticsTables t{};
    if (g_opt.ready) t = g_opt.tab;
    else optics_default_tables(t, 4, 0.0390625, 10.0);
    if (wavelength_um20) std::memcpy(wavelength_um20, t.wavelength_um, sizeof t.wavelength_um);
    if (refindex_9x20x2)
        for (int s = 0; s < OPT_NSPEC; ++s)
            for (int w = 0; w < OPT_NWAVE; ++w) {
                refindex_9x20x2[(s * OPT_NWAVE + w) * 2] = t.refr[s][w];
                refindex_9x20x2[(s * OPT_NWAVE + w) * 2 + 1] = t.refi[s][w];
            }
    if (density9) std::memcpy(density9, t.dens, sizeof t.dens);
    return 0;
}

extern "C" int rrtmg_hip_optical_averaging(const rrtmg_hip_optics_args *a, char *err)
{
    if (!g_opt.ready) return failo(err, "rrtmg_hip_optical_averaging: rrtmg_hip_optics_init has not been called");
    if (a->real_bytes != 4 && a->real_bytes != 8) return failo(err, "rrtmg_hip_optical_averaging: real_bytes must be 4 or 8");
    if (a->aer_op_opt < 1 || a->aer_op_opt > 5)
        return failo(err, "rrtmg_hip_optical_averaging: aer_op_opt must be 1..5 (registry.chem:4014-4018)");
    const int ni_m = a->ime - a->ims + 1, nk_m = a->kme - a->kms + 1, nj_m = a->jme - a->jms + 1;
    const int ni_t = a->ite - a->its + 1, nj_t = a->jte - a->jts + 1, nk_t = a->kte - a->kts + 1;
    if (ni_t <= 0 || nj_t <= 0) return 0;   // an empty tile: the reference's i/j loops do not execute
    if (nk_t <= 0 || a->its < a->ims || a->ite > a->ime || a->jts < a->jms ||
        a->jte > a->jme || a->kts < a->kms || a->kte > a->kme)
        return failo(err, "rrtmg_hip_optical_averaging: inconsistent tile / memory dimensions");
    if (a->num_chem < 1) return failo(err, "rrtmg_hip_optical_averaging: num_chem must be positive");
    if (!a->chem || !a->alt || !a->dz8w) return failo(err, "rrtmg_hip_optical_averaging: a required input array is NULL");
    const int nbin = g_opt.tab.nbin;
    OpticsKernelArgs k{};
    for (int b = 0; b < OPT_MAXBIN; ++b)
        for (int s = 0; s < OPT_NSLOT; ++s) {
            const int id = b < nbin ? a->idx[b][s] : 0;
            if (id < 0 || id > a->num_chem) return failo(err, "rrtmg_hip_optical_averaging: species index outside 1..num_chem");
            k.idx[b][s] = id - 1;
        }
    // cloud-borne counterparts (optional): combined with the interstitial tracers by a pre-pass, so that the Mie kernels
    // see one set of tracers and stay as they are
    int cw[OPT_MAXBIN][OPT_NSLOT];
    bool any_cw = false;
    for (int b = 0; b < OPT_MAXBIN; ++b)
        for (int s = 0; s < OPT_NSLOT; ++s) {
            const int id = (b < nbin && s != OS_WATER) ? a->idx_cw[b][s] : 0;
            if (id < 0 || id > a->num_chem) return failo(err, "rrtmg_hip_optical_averaging: cloud-borne species index outside 1..num_chem");
            cw[b][s] = id - 1;
            any_cw = any_cw || id > 0;
        }
    auto group = [](void *const *p, int n) {
        int cnt = 0;
        for (int i = 0; i < n; ++i) cnt += p[i] != nullptr;
        return cnt == 0 ? 0 : (cnt == n ? 1 : -1);
    };
    void *const sw3[12] = {a->tauaer[0], a->tauaer[1], a->tauaer[2], a->tauaer[3], a->waer[0], a->waer[1],
                           a->waer[2], a->waer[3], a->gaer[0], a->gaer[1], a->gaer[2], a->gaer[3]};
    if (group(sw3, 12) < 0 || group(a->extaer, 4) < 0 || group(a->bscoef, 4) < 0 || group(a->tauaerlw, 16) < 0 ||
        group(a->extaerlw, 16) < 0 || group(a->laer, 6) < 0)
        return failo(err, "rrtmg_hip_optical_averaging: an output group (tauaer/waer/gaer, extaer, bscoef, tauaerlw, "
                          "extaerlw, l2aer..l7aer) must be passed complete or not at all");
    HIPCHK(hipSetDevice(g_opt.device));
    hipStream_t stream = a->mem == 1 ? static_cast<hipStream_t>(a->stream) : nullptr;
    k.ni_m = ni_m; k.nk_m = nk_m; k.nj_m = nj_m;
    k.i0 = a->its - a->ims; k.k0 = a->kts - a->kms; k.j0 = a->jts - a->jms;
    k.ni_t = ni_t; k.nk_t = nk_t; k.nj_t = nj_t;
    k.real4 = a->real_bytes == 4;
    k.mix = a->aer_op_opt == 5 ? 2 : ((a->aer_op_opt == 2 || a->aer_op_opt == 4) ? 1 : 0);   // volume, Maxwell-Garnett, core + shell
    k.nbin = nbin;
    // mem = 0: tile-only staging through this thread's buffers (tile_staging.h); of the tracer array only the span of
    // species the sections use travels (MOSAIC's aerosol tracers are contiguous in the Registry)
    TileStager stg(TileGeom{ni_m, nk_m, nj_m, a->its - a->ims, a->jts - a->jms, ni_t, nj_t, (size_t)a->real_bytes});
    if (a->mem == 1) {
        k.chem = a->chem; k.alt = a->alt; k.dz8w = a->dz8w;
        for (int w = 0; w < OPT_NSW; ++w) {
            k.tauaer[w] = a->tauaer[w]; k.waer[w] = a->waer[w]; k.gaer[w] = a->gaer[w];
            k.extaer[w] = a->extaer[w]; k.bscoef[w] = a->bscoef[w];
        }
        for (int b = 0; b < OPT_NLW; ++b) { k.tauaerlw[b] = a->tauaerlw[b]; k.extaerlw[b] = a->extaerlw[b]; }
        for (int l = 0; l < 6; ++l) k.laer[l] = a->laer[l];
    } else {
        int lo = a->num_chem, hi = -1;
        for (int b = 0; b < nbin; ++b)
            for (int sp = 0; sp < OPT_NSLOT; ++sp)
                for (const int id : {k.idx[b][sp], cw[b][sp]})
                    if (id >= 0) { lo = id < lo ? id : lo; hi = id > hi ? id : hi; }
        if (hi < lo) { lo = 0; hi = 0; }
        for (int b = 0; b < nbin; ++b)
            for (int sp = 0; sp < OPT_NSLOT; ++sp) {
                if (k.idx[b][sp] >= 0) k.idx[b][sp] -= lo;
                if (cw[b][sp] >= 0) cw[b][sp] -= lo;
            }
        k.ni_m = ni_t; k.nj_m = nj_t; k.i0 = 0; k.j0 = 0;   // packed arrays: memory = tile (k unchanged)
        const size_t sspec = (size_t)ni_m * nk_m * nj_m * a->real_bytes;
        stg.in(static_cast<const char *>(a->chem) + (size_t)lo * sspec, nk_m, hi - lo + 1, &k.chem);
        stg.in(a->alt, nk_m, 1, &k.alt);
        stg.in(a->dz8w, nk_m, 1, &k.dz8w);
        for (int w = 0; w < OPT_NSW; ++w) {
            stg.out(a->tauaer[w], nk_m, 1, &k.tauaer[w]); stg.out(a->waer[w], nk_m, 1, &k.waer[w]);
            stg.out(a->gaer[w], nk_m, 1, &k.gaer[w]); stg.out(a->extaer[w], nk_m, 1, &k.extaer[w]);
            stg.out(a->bscoef[w], nk_m, 1, &k.bscoef[w]);
        }
        for (int b = 0; b < OPT_NLW; ++b) {
            stg.out(a->tauaerlw[b], nk_m, 1, &k.tauaerlw[b]); stg.out(a->extaerlw[b], nk_m, 1, &k.extaerlw[b]);
        }
        for (int l = 0; l < 6; ++l) stg.out(a->laer[l], nk_m, OPT_NSW, &k.laer[l]);   // (i, k, j, 4)
        // chunks of the tile, all on this thread's own stream (tile_staging.h): this stage moves the most bytes per
        // column of the three (80 tracers in, 28 fields out per cell), so the overlap of packing, copies and kernels
        // matters most here
        HIPCHK(stg.begin(stage_chunk_columns()));
        stream = stg.stream();
    }
    const int nchunk = a->mem == 1 ? 1 : stg.nchunks();
    for (int c = 0; c < nchunk; ++c) {
        if (a->mem != 1) {
            HIPCHK(stg.stage(c));
            k.nj_m = k.nj_t = stg.chunk_nj(c);
        }
        if (any_cw) {
            // stream-ordered temporary of nbin * 10 fields in the memory shape of this launch
            const size_t ncell = (size_t)k.ni_m * k.nk_m * k.nj_m;
            void *tmp = nullptr;
            HIPCHK(hipMallocAsync(&tmp, (size_t)nbin * OPT_NSLOT * ncell * a->real_bytes, stream));
            {
                const hipError_t ec = optics_combine_cw(k.chem, tmp, k.idx, cw, nbin, ncell, k.real4, stream);
                if (ec != hipSuccess) { (void)hipFreeAsync(tmp, stream); HIPCHK(ec); }
            }
            OpticsKernelArgs kc = k;
            kc.chem = tmp;
            for (int b = 0; b < OPT_MAXBIN; ++b)
                for (int s = 0; s < OPT_NSLOT; ++s) kc.idx[b][s] = b < nbin ? b * OPT_NSLOT + s : -1;
            const hipError_t e = optics_launch(kc, stream, g_opt.precision);
            (void)hipFreeAsync(tmp, stream);
            HIPCHK(e);
        } else {
            HIPCHK(optics_launch(k, stream, g_opt.precision));
        }
        if (a->mem != 1) HIPCHK(stg.fetch(c));
    }
    if (a->mem != 1) {
        int status = 0;
        HIPCHK(stg.finish(&status, nullptr));
    }
    return 0;
}

extern "C" int rrtmg_hip_optics_precision(int mode, char *err)
{
    if (mode != 0 && mode != 1) return failo(err, "rrtmg_hip_optics_precision: mode must be 0 (fp64) or 1 (mixed)");
    std::lock_guard<std::mutex> lock(g_opt_mutex);
    g_opt.precision = mode;
    return 0;
}

extern "C" int rrtmg_hip_optics_finalize(void)
{
    {
        std::lock_guard<std::mutex> lock(g_opt_mutex);
        g_opt.ready = false;
    }
    stage_pools_release_if_idle();
    return 0;
}
