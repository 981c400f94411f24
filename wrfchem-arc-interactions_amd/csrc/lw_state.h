// Host-side state of the longwave engine.
#pragma once
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "lw_common.h"

namespace rrtmg {

struct LwHostTables {
    std::vector<double> K, exp_tbl, tfn_tbl, totplnk, chi_mls;
    std::vector<LwGDesc> gdesc;
    double bpade = 0, heatfac = 0, fluxfac = 0, oneminus = 0;
};

int lw_build_tables(const char *path, int real_bytes, double cpdair, LwHostTables &out, std::string &err);

struct LwState {
    bool ready = false;
    int device = 0;
    int nlayers = 0;         // module variable NLAYERS of rrtmg_lwinit (LW:12861), in the REAL of the table file
    int kme = 0;             // what NLAYERS was formed from: a call whose arrays have the other REAL width forms it
    double p_top_pa = 0.;    // again in its own precision (rrtmg_hip_lwrad)
    LwHostTables host;
    LwDeviceTables dev{};
    void *dK = nullptr, *dExp = nullptr, *dTfn = nullptr, *dPlk = nullptr, *dChi = nullptr, *dG = nullptr,
         *dStatus = nullptr;
    void *dPairs = nullptr;                // LwDeviceTables::tbl_pairs
    void *dCld = nullptr;                  // cloud absorption coefficient tables (lw_cloud_tables)
    std::map<int, void *> dJump;           // KISS jump tables by NLAYERS (guarded by scratch_mutex)
    // Scratch between the two kernels (taug, frac, column record), one grow-only buffer per stream: calls
    // on one stream run in order and may share it, calls on different streams (the OpenMP tile loop of
    // DRV:975-982) must not.
    struct Scratch {
        double *p = nullptr; size_t doubles = 0; size_t limit = 0; /* doubles that fitted after an OOM */
        int *lists = nullptr; size_t lists_ints = 0;   // column lists of the cloud-free / cloudy dispatch (cloud_lists.h)
    };
    std::map<void *, Scratch> scratch;
    std::mutex scratch_mutex;
};

LwState &lw_state();

}  // namespace rrtmg
