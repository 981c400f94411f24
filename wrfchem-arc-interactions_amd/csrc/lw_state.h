// Host-side state of the longwave engine.
#pragma once
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
    int nlayers = 0;         // module variable NLAYERS of rrtmg_lwinit (LW:12861)
    LwHostTables host;
    LwDeviceTables dev{};
    void *dK = nullptr, *dExp = nullptr, *dTfn = nullptr, *dPlk = nullptr, *dChi = nullptr, *dG = nullptr,
         *dStatus = nullptr;
};

LwState &lw_state();

}  // namespace rrtmg
