// Launch interface of the aerosol optical-averaging kernel (optics_kernel.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "optics_common.h"

namespace rrtmg {

hipError_t optics_upload_tables(const OpticsTables &t);
// precision: 0 = fp64 throughout, 1 = "mixed" (fp64 recurrences, packed-fp32 coefficients and sums)
hipError_t optics_launch(const OpticsKernelArgs &a, hipStream_t stream, int precision);
// cloud-borne aerosol: out[b * 10 + s][ncell] = chem[ia[b][s]][.] + chem[ic[b][s]][.] (an index of -1 adds nothing)
hipError_t optics_combine_cw(const void *chem, void *out, const int (*ia)[OPT_NSLOT], const int (*ic)[OPT_NSLOT], int nbin,
                             size_t ncell, int real4, hipStream_t stream);

}  // namespace rrtmg
