// Cloud-free and cloudy columns of a tile go to different kernel instantiations.
//
// With icloud /= 0 the reference runs McICA in every column; a column without cloud (cldfra3d < 1e-20 in all of its
// layers: cldprmc / the sub-column generator then see no cloud, SW:10371-10618, LW:11921-12140) gives clear == total
// and needs none of it.  The McICA instantiations of the column kernels are the expensive ones (a third shortwave pass,
// spills, the hydrometeor arrays), and a realistic domain is far from overcast, so the entry points first sort the
// columns of a launch unit into two lists and launch the cloud-free instantiation over one and the McICA
// instantiation over the other.  The lists live on the device; a kernel takes entry col0 + slot of its list and
// workgroups beyond the list's length leave at once, so no count travels to the host.  Which kernel serves a column
// depends on the column alone: results do not depend on the tiling.
#pragma once
#include <hip/hip_runtime.h>

namespace rrtmg {

// lists: [2][ncol] ints (cloud-free, cloudy: tile column indices, consecutive runs of up to 64 kept together);
// counts: [2] ints, zeroed by the call.  Enqueued on `stream`.
hipError_t cloud_lists_launch(const void *cldfra3d, int real4, int ni_m, int nk_m, int i0, int j0, int k0, int ni_t, int nj_t,
                              int nk, int *lists, int *counts, hipStream_t stream);
// RRTMG_HIP_CLOUD_LISTS=0 switches the sorting off (every column through the McICA instantiation, as round 2 did)
bool cloud_lists_enabled();

}  // namespace rrtmg
