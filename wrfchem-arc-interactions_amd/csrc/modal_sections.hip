// Modal (MADE/SORGAM-type) aerosol -> MOSAIC-type sections, the form the optical-averaging kernels take.
//
// north_star names "MOSAIC/MADE size bins".  The reference carries neither tracer family's optics arithmetic
// (optics_common.h), so this step is the engine's own specification: every mode is a log-normal distribution with a
// fixed geometric standard deviation sigma; its number N and dry volume V fix the number-median diameter
//     dgn = (6 V / (pi N))^(1/3) exp(-1.5 ln^2 sigma),       volume-median  dgv = dgn exp(3 ln^2 sigma),
// and bin b receives the fractions
//     fn_b = Phi((ln dhi_b - ln dgn)/ln sigma) - Phi((ln dlo_b - ln dgn)/ln sigma)        of the number,
//     fv_b = the same with dgv                                                            of every species mass
// (internal mixture inside a mode, water included).  The first bin extends down to zero and the last up to infinity,
// so number and mass are conserved.  Phi(z) = erfc(-z/sqrt 2)/2.  Bandwidth-trivial: one thread per cell.
//
// Output layout: chem_sec(ims:ime, kms:kme, jms:jme, nbin*10), species s (order of enum RRTMG_HIP_AER_*) of bin b at
// 1-based index b*10 + s + 1 -- the idx table rrtmg_hip_modal_section_idx() returns for rrtmg_hip_optical_averaging.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rrtmg_hip.h"
#include "optics_common.h"

namespace rrtmg {
// c_abi_optics.hip
bool optics_ready();
int optics_device();
const OpticsTables &optics_tables();
}  // namespace rrtmg

namespace {

using namespace rrtmg;

struct ModalArgs {
    int ni_m, nk_m, nj_m, i0, k0, j0, ni_t, nk_t, nj_t, real4;
    int nmode, nbin;
    int idx[RRTMG_HIP_MAXMODE][OPT_NSLOT];   // 0-based into chem, -1 = absent
    double lnsig[RRTMG_HIP_MAXMODE];
    double dens[OPT_NSPEC], lndlo[OPT_MAXBIN], lndhi[OPT_MAXBIN];
    const void *chem;
    void *sec;
};

__device__ __forceinline__ double ldv(const void *p, size_t q, int r4)
{
    return r4 ? (double)reinterpret_cast<const float *>(p)[q] : reinterpret_cast<const double *>(p)[q];
}
__device__ __forceinline__ void stv(void *p, size_t q, int r4, double v)
{
    if (r4) reinterpret_cast<float *>(p)[q] = (float)v;
    else reinterpret_cast<double *>(p)[q] = v;
}
__device__ __forceinline__ double phi(double z) { return 0.5 * erfc(-z * 0.70710678118654752440); }

__global__ void __launch_bounds__(256) modal_sections_kernel(const ModalArgs a)
{
    const long cell = (long)blockIdx.x * 256 + threadIdx.x;
    const long ncell = (long)a.ni_t * a.nk_t * a.nj_t;
    if (cell >= ncell) return;
    const int it = (int)(cell % a.ni_t);
    const long r = cell / a.ni_t;
    const int k = (int)(r % a.nk_t), jt = (int)(r / a.nk_t);
    const size_t q = ((size_t)(a.j0 + jt) * a.nk_m + (a.k0 + k)) * a.ni_m + (a.i0 + it);
    const size_t sspec = (size_t)a.ni_m * a.nk_m * a.nj_m;
    const int r4 = a.real4;
    const double pi = 3.14159265358979323846;
    double out[OPT_MAXBIN][OPT_NSLOT];
#pragma unroll
    for (int b = 0; b < OPT_MAXBIN; ++b)
#pragma unroll
        for (int s = 0; s < OPT_NSLOT; ++s) out[b][s] = 0.;
    for (int m = 0; m < a.nmode; ++m) {
        double mass[OPT_NSPEC], vdry = 0.;
#pragma unroll
        for (int s = 0; s < OPT_NSPEC; ++s) {
            const int id = a.idx[m][s];
            mass[s] = id >= 0 ? fmax(ldv(a.chem, (size_t)id * sspec + q, r4), 0.) : 0.;
            if (s != OS_WATER) vdry += mass[s] / a.dens[s];          // 1e-6 cm3 per kg of dry air
        }
        const int idn = a.idx[m][OS_NUM];
        const double num = idn >= 0 ? fmax(ldv(a.chem, (size_t)idn * sspec + q, r4), 0.) : 0.;
        if (!(vdry > 1.e-25) || !(num > 0.)) continue;               // empty mode
        const double ls = a.lnsig[m];
        const double lndgn = log(1.e4 * cbrt(vdry * 1.e-6 / num * (6. / pi))) - 1.5 * ls * ls;   // ln(um)
        const double lndgv = lndgn + 3. * ls * ls;
        const double rls = 1. / ls;
        double cn_lo = 0., cv_lo = 0.;                               // cumulative fractions at the lower edge
        for (int b = 0; b < a.nbin; ++b) {
            const bool last = b == a.nbin - 1;
            const double cn_hi = last ? 1. : phi((a.lndhi[b] - lndgn) * rls);
            const double cv_hi = last ? 1. : phi((a.lndhi[b] - lndgv) * rls);
            const double fn = cn_hi - cn_lo, fv = cv_hi - cv_lo;
#pragma unroll
            for (int s = 0; s < OPT_NSPEC; ++s) out[b][s] += mass[s] * fv;
            out[b][OS_NUM] += num * fn;
            cn_lo = cn_hi; cv_lo = cv_hi;
        }
    }
    for (int b = 0; b < a.nbin; ++b)
#pragma unroll
        for (int s = 0; s < OPT_NSLOT; ++s) stv(a.sec, (size_t)(b * OPT_NSLOT + s) * sspec + q, r4, out[b][s]);
}

int failm(char *err, const std::string &msg)
{
    if (err) {
        std::strncpy(err, msg.c_str(), RRTMG_HIP_ERRLEN - 1);
        err[RRTMG_HIP_ERRLEN - 1] = 0;
    }
    return 1;
}

}  // namespace

#define HIPCHK(expr)                                                                                        \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) {                                                                             \
            for (void *p_ : bufs) (void)hipFree(p_);                                                        \
            return failm(err, std::string("rrtmg_hip_modal_to_sections: ") + #expr + ": " + hipGetErrorString(e_)); \
        }                                                                                                   \
    } while (0)

extern "C" int rrtmg_hip_modal_section_idx(int nbin, int *idx /* [RRTMG_HIP_MAXBIN][RRTMG_HIP_AER_NSPEC] */)
{
    if (nbin < 1 || nbin > OPT_MAXBIN || !idx) return 1;
    for (int b = 0; b < OPT_MAXBIN; ++b)
        for (int s = 0; s < OPT_NSLOT; ++s) idx[b * OPT_NSLOT + s] = b < nbin ? b * OPT_NSLOT + s + 1 : 0;
    return 0;
}

extern "C" int rrtmg_hip_modal_to_sections(const rrtmg_hip_modal_args *a, char *err)
{
    std::vector<void *> bufs;
    if (!a) return failm(err, "rrtmg_hip_modal_to_sections: NULL argument");
    if (!optics_ready()) return failm(err, "rrtmg_hip_modal_to_sections: rrtmg_hip_optics_init has not been called");
    if (a->real_bytes != 4 && a->real_bytes != 8) return failm(err, "rrtmg_hip_modal_to_sections: real_bytes must be 4 or 8");
    if (a->nmode < 1 || a->nmode > RRTMG_HIP_MAXMODE) return failm(err, "rrtmg_hip_modal_to_sections: nmode must be 1..4");
    const int ni_m = a->ime - a->ims + 1, nk_m = a->kme - a->kms + 1, nj_m = a->jme - a->jms + 1;
    const int ni_t = a->ite - a->its + 1, nj_t = a->jte - a->jts + 1, nk_t = a->kte - a->kts + 1;
    if (ni_t <= 0 || nj_t <= 0 || nk_t <= 0 || a->its < a->ims || a->ite > a->ime || a->jts < a->jms ||
        a->jte > a->jme || a->kts < a->kms || a->kte > a->kme)
        return failm(err, "rrtmg_hip_modal_to_sections: inconsistent tile / memory dimensions");
    if (a->num_chem < 1 || !a->chem || !a->chem_sec) return failm(err, "rrtmg_hip_modal_to_sections: chem / chem_sec / num_chem");
    const OpticsTables &t = optics_tables();
    ModalArgs k{};
    k.nmode = a->nmode; k.nbin = t.nbin;
    for (int m = 0; m < a->nmode; ++m) {
        if (!(a->sigma[m] > 1.0)) return failm(err, "rrtmg_hip_modal_to_sections: geometric standard deviations must exceed 1");
        k.lnsig[m] = std::log(a->sigma[m]);
        for (int s = 0; s < OPT_NSLOT; ++s) {
            const int id = a->idx[m][s];
            if (id < 0 || id > a->num_chem) return failm(err, "rrtmg_hip_modal_to_sections: species index outside 1..num_chem");
            k.idx[m][s] = id - 1;
        }
    }
    for (int s = 0; s < OPT_NSPEC; ++s) k.dens[s] = t.dens[s];
    for (int b = 0; b < t.nbin; ++b) { k.lndlo[b] = std::log(t.dlo_um[b]); k.lndhi[b] = std::log(t.dhi_um[b]); }
    k.ni_m = ni_m; k.nk_m = nk_m; k.nj_m = nj_m;
    k.i0 = a->its - a->ims; k.k0 = a->kts - a->kms; k.j0 = a->jts - a->jms;
    k.ni_t = ni_t; k.nk_t = nk_t; k.nj_t = nj_t;
    k.real4 = a->real_bytes == 4;
    HIPCHK(hipSetDevice(optics_device()));
    hipStream_t stream = a->mem == 1 ? static_cast<hipStream_t>(a->stream) : nullptr;
    const size_t n3 = (size_t)ni_m * nk_m * nj_m * a->real_bytes;
    const size_t nout = n3 * (size_t)t.nbin * OPT_NSLOT;
    if (a->mem == 1) {
        k.chem = a->chem; k.sec = a->chem_sec;
    } else {
        void *din = nullptr, *dout = nullptr;
        HIPCHK(hipMalloc(&din, n3 * (size_t)a->num_chem)); bufs.push_back(din);
        HIPCHK(hipMalloc(&dout, nout)); bufs.push_back(dout);
        HIPCHK(hipMemcpyAsync(din, a->chem, n3 * (size_t)a->num_chem, hipMemcpyHostToDevice, stream));
        HIPCHK(hipMemcpyAsync(dout, a->chem_sec, nout, hipMemcpyHostToDevice, stream));   // only the tile is rewritten
        k.chem = din; k.sec = dout;
    }
    const long ncell = (long)ni_t * nk_t * nj_t;
    hipLaunchKernelGGL(modal_sections_kernel, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, stream, k);
    HIPCHK(hipGetLastError());
    if (a->mem != 1) {
        HIPCHK(hipMemcpyAsync(a->chem_sec, k.sec, nout, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        for (void *p_ : bufs) (void)hipFree(p_);
    }
    return 0;
}
