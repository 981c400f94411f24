// Launch interface of the longwave column kernel (lw_kernel.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "lw_common.h"

namespace rrtmg {

struct LwKernelArgs {
    int ims, jms, kms, ni_m, nk_m, nj_m;
    int its, jts, kts, ni_t, nj_t, nk;
    int nlayers;                        // model layers + buffer layers to 0 mb (LW:12861)
    int real4;
    int aer_ra_feedback, clean, o3input;
    const void *emiss, *tsk;
    const void *t3d, *t8w, *p3d, *p8w, *pi3d, *qv3d, *o33d;
    const void *tauaerlw[LW_NBND];
    void *rthratenlw, *glw, *olr, *lwcf;
    void *lwupt, *lwuptc, *lwuptcln, *lwdnt, *lwdntc, *lwdntcln;
    void *lwupb, *lwupbc, *lwupbcln, *lwdnb, *lwdnbc, *lwdnbcln;
    void *lwupflx, *lwupflxc, *lwupflxcln, *lwdnflx, *lwdnflxc, *lwdnflxcln;
    int *status;
    // scratch between the gas-optics kernel and the transfer kernel, per column slot of the batch (frac is written and
    // read only where the regime interpolates it -- elsewhere it is a constant of the g-point, LwRegime::fracconst):
    // taug / frac [slot][band][layer][ig] (band-major: the band's block starts at nlayers * g0(band), a layer's row
    // of the band's ng g-points is contiguous, consecutive layers follow each other); rec [slot][lw_rec_doubles(nlayers)]
    double *taug, *frac, *rec;
    double *flx;                        // [slot][lw_flux_doubles(nlayers)]: flux slots of lw_rt_fast_tbl_kernel
    int col0;                           // first tile column of this batch
    int nbatch;                         // columns of this batch
    int stage_cont;                     // gas kernel: continuum rows of a segment's band through LDS (lw_launch decides)
    const int *colmap, *colcount;       // column list of this call (cloud_lists.h) or null: columns col0 .. col0+nbatch-1
    // clouds (icloud != 0), as in SwKernelArgs
    int icloud, has_reqc, has_reqi, has_reqs, warm_rain, progn;
    int f_ice_phy, fer_hires;            // "MP option 5" split of snow; Ferrier-Aligo (qi3d = all frozen water)
    const void *iradius, *lradius;       // CAM-MG effective radii (um), both or none
    double grav;
    const void *cldfra3d, *qc3d, *qr3d, *qi3d, *qs3d, *re_cloud, *re_ice, *re_snow, *qndrop3d, *xland, *xice, *snow;
    const uint32_t *jump;               // KISS jump table for NLAYERS draws per sub-column: [LW_NGPT][36]
    const double *cldtab;               // [2][LW_NBND][58]: absliq1, absice3 (lwcldpr, LW:9858-10496)
    LwDeviceTables tab;
};

void lw_cloud_tables(double *out /*[2][LW_NBND][58]*/);
hipError_t lw_upload_retab();

// per-column record handed from lw_gas_kernel to the transfer kernels:
//   secdiff[16] plankbnd[16] pz[nl+1] laytrop                                              (lw_rec_base_doubles)
//   the in-cloud state of the column (written when icloud != 0): cldf clwp ciwp cswp fliq fice fsno [nl] each,
//   iliq iice isno [nl] each (integers stored as doubles), seeds[4]
//   q4 [16][nl4][4] = planklay, planklev(k) - planklay, planklev(k+1) - planklay, taua of every (band, layer), nl4 = nl
//   rounded up to whole blocks of four layers, on a 128-byte boundary: a band's block of four layers is one line, read
//   with two 16-byte loads per layer.  (Three separate arrays cost three line fills per band and layer -- the
//   streaming gas optics turn the L1 over between two layers -- and were 10 of the transfer kernel's 24 ms.)
__host__ __device__ inline size_t lw_rec_base_doubles(int nl) { return (size_t)nl + 34; }
__host__ __device__ inline size_t lw_rec_q4_offset(int nl) { return (lw_rec_base_doubles(nl) + (size_t)10 * nl + 4 + 15) & ~(size_t)15; }
__host__ __device__ inline size_t lw_rec_q4_layers(int nl) { return (size_t)4 * ((nl + 3) / 4); }
//   tc [16][nl4] (McICA calls): the cloud absorption optical depth of every (band, layer) (cldprmc, LW:2800-2908) -- it
//   does not depend on the sub-column, so the gas kernel forms it once and the cloudy lanes of the transfer kernel take
//   it like the quadruples; a size out of the tables' bounds is stored as minus its kind (fatal only if a cloudy
//   sub-column uses it, as in the reference)
__host__ __device__ inline size_t lw_rec_tc_offset(int nl) { return lw_rec_q4_offset(nl) + (size_t)LW_NBND * lw_rec_q4_layers(nl) * 4; }
__host__ __device__ inline size_t lw_rec_doubles(int nl) { return lw_rec_tc_offset(nl) + (size_t)LW_NBND * lw_rec_q4_layers(nl); }
// flux slots of a column in the transfer kernel without profile outputs: per chunk of 16 g-points the downward and
// upward flux of the call with aerosol at every level and eight small sums (lw_rt_fast in lw_kernel.hip)
constexpr int LW_RTF_NCHUNK = 9;
__host__ __device__ inline size_t lw_flux_doubles(int nl) { return (size_t)LW_RTF_NCHUNK * (2 * (nl + 1) + 8); }
// scratch doubles one column needs (taug + frac + record + flux slots)
inline size_t lw_scratch_doubles(int nl) { return (size_t)2 * LW_NGPT * nl + lw_rec_doubles(nl) + lw_flux_doubles(nl); }

size_t lw_gas_lds_bytes(int nlayers, bool stage_continua = false);
size_t lw_rt_lds_bytes(int nlayers, bool cloud = true);
size_t lw_rt_fast_lds_bytes(int nlayers, bool cloud = true);   // the two-column kernel of calls without flux profiles
hipError_t lw_upload_constants(const double *preflog, const double *tref);
hipError_t lw_fill_pairs(double *out /*[LW_NTBL + 1][2]*/, double bpade);   // LwDeviceTables::tbl_pairs
hipError_t lw_upload_segments(const LwGDesc *gdesc);   // [LW_NGPT] host descriptors
// runs ncol columns in batches of at most cap_cols through the two kernels; `scratch` holds
// cap_cols * lw_scratch_doubles(nlayers) doubles
hipError_t lw_launch(LwKernelArgs a, int ncol, double *scratch, int cap_cols, hipStream_t stream);

}  // namespace rrtmg
