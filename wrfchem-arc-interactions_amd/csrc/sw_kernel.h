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
    int *status;                         // device word: 0 ok, 1+col = negative column AOD at that column
    SwDeviceTables tab;
};

size_t sw_lds_bytes(int nk);
hipError_t sw_upload_constants(hipStream_t stream);
hipError_t sw_launch(const SwKernelArgs &a, int ncol, hipStream_t stream);

}  // namespace rrtmg
