// Host-side state of the engine: tables built at init, device copies.
#pragma once
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "rrtmg_common.h"

namespace rrtmg {

struct SwHostTables {
    std::vector<double> K;
    std::vector<double> exp_tbl;
    std::vector<GDesc> gdesc;
    double bpade = 0, heatfac = 0, oneminus = 0;
};

// sw_tables.hip
int sw_build_tables(const char *path, int real_bytes, double cpdair, SwHostTables &out, std::string &err);

struct SwState {
    bool ready = false;
    int device = 0;
    SwHostTables host;
    SwDeviceTables dev{};
    void *dK = nullptr, *dExp = nullptr, *dG = nullptr, *dStatus = nullptr;
    void *dCld = nullptr;                  // cloud optics coefficient tables (sw_cloud_tables)
    std::map<int, void *> dJump;           // KISS jump tables by layer count
    std::mutex jump_mutex;
    // layer scratch of the column kernel, one grow-only buffer per stream (guarded by jump_mutex)
    struct Scratch {
        double *p = nullptr; size_t doubles = 0; size_t limit = 0; /* doubles that fitted after an OOM */
        int *lists = nullptr; size_t lists_ints = 0;   // column lists of the cloud-free / cloudy dispatch (cloud_lists.h)
    };
    std::map<void *, Scratch> scratch;
};

SwState &sw_state();

}  // namespace rrtmg
