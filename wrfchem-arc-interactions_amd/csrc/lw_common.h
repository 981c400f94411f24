// Longwave: shared host/device definitions.
//
// Reference: phys/module_ra_rrtmg_lw.F -- 16 bands, 140 reduced g-points
// (ngc, LW:8080); taugb1..16 (LW:4712-7828) are encoded per g-point so that a
// wavefront whose lanes are g-points runs one instruction stream.
#pragma once
#include <cstdint>

namespace rrtmg {

constexpr int LW_NBND = 16;
constexpr int LW_NGPT = 140;
constexpr int LW_NTBL = 10000;

// multiplier slots of the per-layer array mul[slot][k] (LDS)
enum LwSlot : int {
    LS_H2O = 0, LS_CO2, LS_O3, LS_N2O, LS_CO, LS_CH4, LS_O2, LS_BRD, LS_ZERO,
    LS_ADJ_N2O,      // bands 3, 9:   thr 1.5, a0 0.5, exp 0.65, chi_mls(4,jp+1)
    LS_ADJ_CO2_6,    // band 6:       thr 3, a0 2, exp 0.77
    LS_ADJ_CO2_7L,   // band 7 lower: thr 3, a0 3, exp 0.79
    LS_ADJ_CO2_7U,   // band 7 upper: thr 3, a0 2, exp 0.79
    LS_ADJ_CO2_8,    // band 8:       thr 3, a0 2, exp 0.65
    LS_ADJ_CO2_13,   // band 13:      thr 3, a0 2, exp 0.68, chi = 3.55e-4
    LS_SCALEN2A,     // colbrd*scaleminorn2 (band 1)
    LS_SCALEN2B,     // colbrd*scaleminor   (band 15)
    LS_SCALEO2,      // colo2*scaleminor    (band 11)
    LS_WX1, LS_WX2, LS_WX3, LS_WX4,   // ccl4, cfc11, cfc12, cfc22 amounts
    LS_COUNT
};
// species-ratio slots rat[slot][2][k]: chi_mls ratios at jp and jp+1 (LW:3713-3726, 3769-3772)
enum LwRat : int { LR_H2OCO2 = 0, LR_H2OO3, LR_H2ON2O, LR_H2OCH4, LR_N2OCO2, LR_O3CO2, LR_ONE, LR_COUNT };

constexpr uint32_t LC_TYPE_SHIFT = 0;    // 2 bits: 0 none, 1 single species, 2 binary species
constexpr uint32_t LC_SPA_SHIFT = 2;     // 4 bits
constexpr uint32_t LC_SPB_SHIFT = 6;     // 4 bits
constexpr uint32_t LC_RAT_SHIFT = 10;    // 3 bits
constexpr uint32_t LC_NSP_SHIFT = 13;    // 4 bits (1, 5, 9; band 16 upper: 0)
constexpr uint32_t LC_MULT4 = 1u << 17;  // specmult = 4*specparm
constexpr uint32_t LC_EDGES = 1u << 18;  // 3-point stencils below 0.125 / above 0.875 (lower atmosphere)
constexpr uint32_t LC_SELF = 1u << 19;
constexpr uint32_t LC_FOR = 1u << 20;
constexpr uint32_t LC_CORR_SHIFT = 21;   // 2 bits: 0 none, 1 band-1 lower, 2 band-1 upper, 3 band-2 lower
constexpr uint32_t LC_FRAC_SHIFT = 23;   // 2 bits: 0 zero, 1 constant, 2 interpolated in the Planck species index

struct LwMinor {
    int32_t off;      // table [row = indm*nsp + js][ig], 0 = absent (zero pad)
    int32_t nsp;      // 1 (temperature only), 5 or 9
    int32_t slot;     // LwSlot multiplier
    int32_t pad_;
    double refrat;    // species ratio of the minor's reference atmosphere
};

struct LwRegime {
    uint32_t cfg;
    int32_t off_major;
    int32_t off_frac;
    int32_t pad_;
    double fracconst;
    double planck_ratio;
    double postscale;
    LwMinor minor[3];
    int32_t halo_slot[2];
    double halo_coef[2];
};

struct LwGDesc {
    LwRegime lo, up;
    int32_t band, ig, ng, off_self, off_for;
    int32_t stride;    // row stride of every table of the band: ng rounded up to a multiple of 8 doubles
    double delwave;
};

struct LwDeviceTables {
    const double *K;
    const double *exp_tbl, *tfn_tbl;   // tau_tbl is not needed by cloud-free columns (LW:3284-3296)
    const double *tbl_pairs;           // [LW_NTBL + 1][2]: (1 - exp_tbl, tfn_tbl) as the kernels evaluate them (lw_fill_pairs)
    const double *totplnk;             // [16][181]
    const double *chi_mls;             // [7][59]
    const LwGDesc *gdesc;
    double bpade, heatfac, fluxfac, oneminus;
};

}  // namespace rrtmg
