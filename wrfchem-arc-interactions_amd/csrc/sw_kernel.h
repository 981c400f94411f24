// Launch interface of the shortwave column kernel (sw_kernel.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "rrtmg_common.h"

namespace rrtmg {

struct SwKernelArgs {
    // memory dims (ims:ime, kms:kme, jms:jme) and tile origin
    int ims, jms, kms, ni_m, nk_m, nj_m;
    int its, jts, kts, ni_t, nj_t, nk;   // nk = kte-kts+1 model layers
    int real4;                           // 1: REAL*4 arrays, 0: REAL*8
    int aer_ra_feedback, clean, o3input;
    double solcon;
    const void *xcoszen, *albedo, *tsk;
    const void *t3d, *t8w, *p3d, *p8w, *pi3d, *qv3d, *o33d;
    const void *tauaer300, *tauaer400, *tauaer600, *tauaer999;
    const void *gaer400, *gaer600, *waer400, *waer600;
    void *rthratensw, *coszr, *gsw, *swcf;
    void *swupt, *swuptc, *swuptcln, *swdnt, *swdntc, *swdntcln;
    void *swupb, *swupbc, *swupbcln, *swdnb, *swdnbc, *swdnbcln;
    void *swvisdir, *swvisdif, *swnirdir, *swnirdif, *swddir, *swddni, *swddif;
    void *swupflx, *swupflxc, *swupflxcln, *swdnflx, *swdnflxc, *swdnflxcln;
    int *status;                         // device word: 0 ok, (kind << 28) | (1+col); kind 0 = negative column AOD,
                                         // 1 ice size, 2 snow size, 3 liquid radius out of bounds (cldprmc_sw)
    // clouds (icloud != 0)
    int icloud, has_reqc, has_reqi, has_reqs, warm_rain, progn;
    int f_ice_phy, fer_hires;            // "MP option 5" split of snow; Ferrier-Aligo (qi3d = all frozen water)
    const void *iradius, *lradius;       // CAM-MG effective radii (um), both or none
    double grav;
    const void *cldfra3d, *qc3d, *qr3d, *qi3d, *qs3d, *re_cloud, *re_ice, *re_snow, *qndrop3d, *xland, *xice, *snow;
    const void *tauaer3d_sw, *ssaaer3d_sw, *asyaer3d_sw;   // (i,k,j,band) band-resolved aerosol input, optional
    const void *aerod;                   // (i,k,j,6) ECMWF aerosol types, aer_opt = 1 / iaer = 6, optional
    const void *alswvisdir, *alswvisdif, *alswnirdir, *alswnirdif;   // SSiB spectral albedos (2-D), optional
    const uint32_t *jump;                // KISS jump-ahead table for this layer count: [SW_NGPT][KISS_JUMP_WORDS]
    const double *cldtab;                // cloud optics coefficients [CLD_NTAB][SW_NBND][CLD_STRIDE]
    double *scratch;                     // ncol x scratch_stride doubles
    size_t scratch_stride;               // sw_scratch_doubles_per_column(nk)
    int col0, ncol;                      // this launch covers columns col0 .. col0+ncol-1 of the tile
    // ... or, with a column list (cloud_lists.h: the cloud-free and the cloudy columns of a tile go to their own kernel
    // instantiations), the list entries col0 .. col0+ncol-1 that exist: workgroups beyond *colcount leave at once
    const int *colmap, *colcount;
    SwDeviceTables tab;
};

// KISS jump-ahead (mcica.hip): per sub-column the transform that advances the four generators by
// change_seed + isubcol*nlay draws: LCG (A, C), xorshift matrix (32 rows), two multiply-with-carry factors
constexpr int KISS_JUMP_WORDS = 36;
void kiss_jump_table(int nlay, int change_seed, int nsub, uint32_t *out /*[nsub][KISS_JUMP_WORDS]*/);
// cloud optics tables: 0 extliq1 1 ssaliq1 2 asyliq1 3 extice3 4 ssaice3 5 asyice3 6 fdlice3
constexpr int CLD_NTAB = 7, CLD_STRIDE = 58;
void sw_cloud_tables(double *out /*[CLD_NTAB][SW_NBND][CLD_STRIDE]*/);
hipError_t sw_upload_retab();

size_t sw_lds_bytes(int nk, bool cloud, bool prof, bool compact_aerosol);
hipError_t sw_upload_constants(hipStream_t stream);
size_t sw_scratch_doubles_per_column(int nk);
hipError_t sw_launch(const SwKernelArgs &a, hipStream_t stream);

}  // namespace rrtmg
