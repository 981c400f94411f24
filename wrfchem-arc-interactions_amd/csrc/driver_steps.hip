// Driver-side neighbours of the radiation call, elementwise over the tile's (i,j) points, so that the
// flux fields the column kernels left in HBM are consumed there:
//   * accumulation of the AC* energy buckets every model step   (module_radiation_driver.F:2308-2381)
//   * swint_opt = 1: update of the interpolation parameters after a radiation step (DRV:2668-2749) and the
//     solar-zenith-angle interpolation of the surface fluxes between radiation steps (DRV:2751-2795)
// Arithmetic in fp64 whatever the storage type (a REAL*4 WRF build evaluates the same formulas in fp32); the
// thresholds and the stored exponents are rounded to the storage type so that the branches agree with such a build.
// Bandwidth-trivial: 13-16 values per point.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rrtmg_hip.h"

namespace {

__device__ __forceinline__ double ldv(const void *p, size_t q, int r4)
{
    return r4 ? (double)reinterpret_cast<const float *>(p)[q] : reinterpret_cast<const double *>(p)[q];
}
__device__ __forceinline__ void stv(void *p, size_t q, int r4, double v)
{
    if (r4) reinterpret_cast<float *>(p)[q] = (float)v;
    else reinterpret_cast<double *>(p)[q] = v;
}
// a value as the reference would hold it in its own REAL variable
__device__ __forceinline__ double rnd(double v, int r4) { return r4 ? (double)(float)v : v; }

struct Tile {
    int ni_m, i0, j0, ni_t, nj_t, real4;
};
__device__ __forceinline__ bool point(const Tile &t, size_t &q)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= t.ni_t * t.nj_t) return false;
    const int it = c % t.ni_t, jt = c / t.ni_t;
    q = (size_t)(t.j0 + jt) * t.ni_m + (t.i0 + it);
    return true;
}

struct AccArgs {
    Tile t;
    double dt;
    const void *flux[16];
    void *acc[16];
};
__global__ void __launch_bounds__(256) accumulate_kernel(const AccArgs a)
{
    size_t q;
    if (!point(a.t, q)) return;
#pragma unroll
    for (int f = 0; f < 16; ++f)
        if (a.flux[f] && a.acc[f]) stv(a.acc[f], q, a.t.real4, ldv(a.acc[f], q, a.t.real4) + ldv(a.flux[f], q, a.t.real4) * a.dt);
}

struct UpdArgs {
    Tile t;
    const void *coszen, *coszen_loc, *swddir, *swdown;
    void *swddir_ref, *bb, *Bx, *swdown_ref, *gg, *Gx, *coszen_ref;
};
// exponent and scale of flux = X * coszen**e through the last two radiation steps, DRV:2690-2707 / 2714-2727
__device__ __forceinline__ void fit(double flux, double flux_ref, double cz, double cz_loc, double cz_ref, double X_old,
                                    int r4, double &e, double &X)
{
    double f0, c0;
    if (X_old <= 0.) { f0 = (cz_loc / cz) * flux; c0 = cz_loc; }   // linear first guess
    else { f0 = flux_ref; c0 = cz_ref; }
    const double ratio = cz / c0;
    const double num = log(fmax(1., flux) / fmax(1., f0));
    if (ratio < 1.) e = num / log(fmin(rnd(1. - 1e-4, r4), ratio));
    else if (ratio > 1.) e = num / log(fmax(rnd(1. + 1e-4, r4), ratio));
    else e = 0.;
    e = rnd(fmax(-.5, fmin(2.5, e)), r4);   // as stored: interp_sw_radiation tests bb against -0.5 / 2.5 / 0
    X = flux / pow(cz, e);
}
__global__ void __launch_bounds__(256) swinterp_update_kernel(const UpdArgs a)
{
    size_t q;
    if (!point(a.t, q)) return;
    const int r4 = a.t.real4;
    const double coszen_min = r4 ? (double)1e-4f : 1e-4;
    const double cz = ldv(a.coszen, q, r4), czl = ldv(a.coszen_loc, q, r4);
    const double dir = ldv(a.swddir, q, r4), glo = ldv(a.swdown, q, r4);
    if (cz > coszen_min && czl > coszen_min) {
        double e, X;
        fit(dir, ldv(a.swddir_ref, q, r4), cz, czl, ldv(a.coszen_ref, q, r4), ldv(a.Bx, q, r4), r4, e, X);
        stv(a.bb, q, r4, e); stv(a.Bx, q, r4, X);
        fit(glo, ldv(a.swdown_ref, q, r4), cz, czl, ldv(a.coszen_ref, q, r4), ldv(a.Gx, q, r4), r4, e, X);
        stv(a.gg, q, r4, e); stv(a.Gx, q, r4, X);
    } else {
        stv(a.Bx, q, r4, 0.); stv(a.bb, q, r4, 0.); stv(a.Gx, q, r4, 0.); stv(a.gg, q, r4, 0.);
    }
    // the last radiation step becomes the reference, DRV:2734-2737
    stv(a.coszen_ref, q, r4, cz); stv(a.swdown_ref, q, r4, glo); stv(a.swddir_ref, q, r4, dir);
}

struct IntArgs {
    Tile t;
    const void *coszen_ref, *coszen_loc, *swddir_ref, *bb, *Bx, *swdown_ref, *gg, *Gx, *albedo;
    void *swdown, *swddir, *swddni, *swddif, *gsw;
};
__global__ void __launch_bounds__(256) swinterp_apply_kernel(const IntArgs a)
{
    size_t q;
    if (!point(a.t, q)) return;
    const int r4 = a.t.real4;
    const double coszen_min = r4 ? (double)1e-4f : 1e-4;
    const double czr = ldv(a.coszen_ref, q, r4), czl = ldv(a.coszen_loc, q, r4);
    double dir = 0., glo = 0., dif = 0., dni = 0., g = 0.;
    if (czr > coszen_min && czl > coszen_min) {   // DRV:2772-2786
        const double bb = ldv(a.bb, q, r4), gg = ldv(a.gg, q, r4);
        if (bb == -0.5 || bb == 2.5 || bb == 0.0) dir = (czl / czr) * ldv(a.swddir_ref, q, r4);
        else dir = ldv(a.Bx, q, r4) * pow(czl, bb);
        if (gg == -0.5 || gg == 2.5 || gg == 0.0) glo = (czl / czr) * ldv(a.swdown_ref, q, r4);
        else glo = ldv(a.Gx, q, r4) * pow(czl, gg);
        dir = rnd(dir, r4); glo = rnd(glo, r4);   // as stored, before they are combined
        dif = glo - dir;
        dni = dir / czl;
        g = glo * (1. - ldv(a.albedo, q, r4));
    }
    stv(a.swddir, q, r4, dir); stv(a.swdown, q, r4, glo); stv(a.swddif, q, r4, dif); stv(a.swddni, q, r4, dni);
    stv(a.gsw, q, r4, g);
}

int fail(char *err, const std::string &msg)
{
    if (err) {
        std::strncpy(err, msg.c_str(), RRTMG_HIP_ERRLEN - 1);
        err[RRTMG_HIP_ERRLEN - 1] = 0;
    }
    return 1;
}

#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            st.release();                                                                         \
            return fail(err, std::string(who) + ": " + #expr + ": " + hipGetErrorString(e_));     \
        }                                                                                         \
    } while (0)

// mem = 0: every array is staged through HBM (copied in, outputs copied back), the call is synchronous
struct Stage {
    struct Item { void *host, *dev; bool out; };
    std::vector<Item> items;
    size_t bytes = 0;
    void release()
    {
        for (auto &i : items) if (i.dev) (void)hipFree(i.dev);
        items.clear();
    }
};

bool tile_of(const rrtmg_hip_tile2d *d, Tile &t, size_t &npts_m)
{
    if (d->real_bytes != 4 && d->real_bytes != 8) return false;
    if (d->ite < d->its || d->jte < d->jts || d->its < d->ims || d->ite > d->ime || d->jts < d->jms || d->jte > d->jme)
        return false;
    t.ni_m = d->ime - d->ims + 1;
    t.i0 = d->its - d->ims; t.j0 = d->jts - d->jms;
    t.ni_t = d->ite - d->its + 1; t.nj_t = d->jte - d->jts + 1;
    t.real4 = d->real_bytes == 4;
    npts_m = (size_t)t.ni_m * (d->jme - d->jms + 1);
    return true;
}

}  // namespace

// One helper for the three entry points: maps host arrays to device copies when mem = 0.
#define MAP_IN(field)                                                                                     \
    if (d->mem != 1 && k.field) {                                                                         \
        void *p_ = nullptr;                                                                               \
        HIPCHK(hipMalloc(&p_, nbytes));                                                                   \
        st.items.push_back({const_cast<void *>(static_cast<const void *>(k.field)), p_, false});          \
        HIPCHK(hipMemcpyAsync(p_, k.field, nbytes, hipMemcpyHostToDevice, s));                            \
        k.field = static_cast<const void *>(p_);                                                          \
    }
#define MAP_IO(field)                                                                                     \
    if (d->mem != 1 && k.field) {                                                                         \
        void *p_ = nullptr;                                                                               \
        HIPCHK(hipMalloc(&p_, nbytes));                                                                   \
        st.items.push_back({k.field, p_, true});                                                          \
        HIPCHK(hipMemcpyAsync(p_, k.field, nbytes, hipMemcpyHostToDevice, s));                            \
        k.field = p_;                                                                                     \
    }
#define FINISH()                                                                                          \
    HIPCHK(hipGetLastError());                                                                            \
    if (d->mem != 1) {                                                                                    \
        for (auto &i : st.items)                                                                          \
            if (i.out) HIPCHK(hipMemcpyAsync(i.host, i.dev, nbytes, hipMemcpyDeviceToHost, s));           \
        HIPCHK(hipStreamSynchronize(s));                                                                  \
        st.release();                                                                                     \
    }                                                                                                     \
    return 0;

extern "C" int rrtmg_hip_accumulate_fluxes(const rrtmg_hip_tile2d *d, double dtaccum, const void *const *flux16,
                                           void *const *acc16, char *err)
{
    static const char *who = "rrtmg_hip_accumulate_fluxes";
    Stage st;
    AccArgs k{};
    size_t npts = 0;
    if (!d || !flux16 || !acc16) return fail(err, std::string(who) + ": NULL argument");
    if (!tile_of(d, k.t, npts)) return fail(err, std::string(who) + ": inconsistent tile / memory dimensions or REAL size");
    const size_t nbytes = npts * d->real_bytes;
    hipStream_t s = d->mem == 1 ? static_cast<hipStream_t>(d->stream) : nullptr;
    k.dt = dtaccum;
    for (int f = 0; f < 16; ++f) { k.flux[f] = flux16[f]; k.acc[f] = acc16[f]; }
    for (int f = 0; f < 16; ++f) {
        if (!(k.flux[f] && k.acc[f])) { k.flux[f] = nullptr; k.acc[f] = nullptr; continue; }
        MAP_IN(flux[f]);
        MAP_IO(acc[f]);
    }
    const int n = k.t.ni_t * k.t.nj_t;
    hipLaunchKernelGGL(accumulate_kernel, dim3((n + 255) / 256), dim3(256), 0, s, k);
    FINISH();
}

extern "C" int rrtmg_hip_update_swinterp_parameters(const rrtmg_hip_tile2d *d, const rrtmg_hip_swinterp_update_args *a,
                                                    char *err)
{
    static const char *who = "rrtmg_hip_update_swinterp_parameters";
    Stage st;
    UpdArgs k{};
    size_t npts = 0;
    if (!d || !a) return fail(err, std::string(who) + ": NULL argument");
    if (!tile_of(d, k.t, npts)) return fail(err, std::string(who) + ": inconsistent tile / memory dimensions or REAL size");
    if (!(a->coszen && a->coszen_loc && a->swddir && a->swdown && a->swddir_ref && a->bb && a->Bx && a->swdown_ref &&
          a->gg && a->Gx && a->coszen_ref))
        return fail(err, std::string(who) + ": an array is NULL");
    const size_t nbytes = npts * d->real_bytes;
    hipStream_t s = d->mem == 1 ? static_cast<hipStream_t>(d->stream) : nullptr;
    k.coszen = a->coszen; k.coszen_loc = a->coszen_loc; k.swddir = a->swddir; k.swdown = a->swdown;
    k.swddir_ref = a->swddir_ref; k.bb = a->bb; k.Bx = a->Bx; k.swdown_ref = a->swdown_ref; k.gg = a->gg; k.Gx = a->Gx;
    k.coszen_ref = a->coszen_ref;
    MAP_IN(coszen); MAP_IN(coszen_loc); MAP_IN(swddir); MAP_IN(swdown);
    MAP_IO(swddir_ref); MAP_IO(bb); MAP_IO(Bx); MAP_IO(swdown_ref); MAP_IO(gg); MAP_IO(Gx); MAP_IO(coszen_ref);
    const int n = k.t.ni_t * k.t.nj_t;
    hipLaunchKernelGGL(swinterp_update_kernel, dim3((n + 255) / 256), dim3(256), 0, s, k);
    FINISH();
}

extern "C" int rrtmg_hip_interp_sw_radiation(const rrtmg_hip_tile2d *d, const rrtmg_hip_swinterp_apply_args *a, char *err)
{
    static const char *who = "rrtmg_hip_interp_sw_radiation";
    Stage st;
    IntArgs k{};
    size_t npts = 0;
    if (!d || !a) return fail(err, std::string(who) + ": NULL argument");
    if (!tile_of(d, k.t, npts)) return fail(err, std::string(who) + ": inconsistent tile / memory dimensions or REAL size");
    if (!(a->coszen_ref && a->coszen_loc && a->swddir_ref && a->bb && a->Bx && a->swdown_ref && a->gg && a->Gx &&
          a->albedo && a->swdown && a->swddir && a->swddni && a->swddif && a->gsw))
        return fail(err, std::string(who) + ": an array is NULL");
    const size_t nbytes = npts * d->real_bytes;
    hipStream_t s = d->mem == 1 ? static_cast<hipStream_t>(d->stream) : nullptr;
    k.coszen_ref = a->coszen_ref; k.coszen_loc = a->coszen_loc; k.swddir_ref = a->swddir_ref; k.bb = a->bb; k.Bx = a->Bx;
    k.swdown_ref = a->swdown_ref; k.gg = a->gg; k.Gx = a->Gx; k.albedo = a->albedo;
    k.swdown = a->swdown; k.swddir = a->swddir; k.swddni = a->swddni; k.swddif = a->swddif; k.gsw = a->gsw;
    MAP_IN(coszen_ref); MAP_IN(coszen_loc); MAP_IN(swddir_ref); MAP_IN(bb); MAP_IN(Bx); MAP_IN(swdown_ref); MAP_IN(gg);
    MAP_IN(Gx); MAP_IN(albedo);
    MAP_IO(swdown); MAP_IO(swddir); MAP_IO(swddni); MAP_IO(swddif); MAP_IO(gsw);
    const int n = k.t.ni_t * k.t.nj_t;
    hipLaunchKernelGGL(swinterp_apply_kernel, dim3((n + 255) / 256), dim3(256), 0, s, k);
    FINISH();
}
