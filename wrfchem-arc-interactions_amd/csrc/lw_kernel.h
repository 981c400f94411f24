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
    LwDeviceTables tab;
};

size_t lw_lds_bytes(int nlayers);
hipError_t lw_upload_constants(const double *preflog, const double *tref);
hipError_t lw_launch(const LwKernelArgs &a, int ncol, hipStream_t stream);

}  // namespace rrtmg
