// Host side of the longwave init: RRTMG_LW_DATA loader, 256->140 g-point
// reduction, lookup tables, packing into the device layout.
//
// Replaces rrtmg_lwinit / rrtmg_lwlookuptable / lw_kgb01..16 / rrtmg_lw_ini
// (phys/module_ra_rrtmg_lw.F:12845-14400, 7862-8009, cmbgb1..16 8260-10500).
// Layout: every table of a band is [row][ig] (g fastest) in one allocation;
// the per-g descriptor (LwGDesc) encodes what taugb1..16 do with it.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "lw_common.h"
#include "lw_ref_tables.h"
#include "lw_state.h"

namespace rrtmg {

namespace {

constexpr int MG16 = 16;
// lwcmbdat, LW:8079-8170
const int NGC[LW_NBND] = {10, 12, 16, 14, 16, 8, 12, 8, 12, 6, 8, 8, 4, 2, 2, 2};
const int NGN[LW_NGPT] = {
    1, 1, 2, 2, 2, 2, 2, 2, 1, 1,   1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2,   1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
    1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 3,   1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,   2, 2, 2, 2, 2, 2, 2, 2,
    2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2,   2, 2, 2, 2, 2, 2, 2, 2,   1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2,   2, 2, 2, 2, 4, 4,
    1, 1, 2, 2, 2, 2, 3, 3,   1, 1, 1, 1, 2, 2, 4, 4,   3, 3, 4, 6,   8, 8,   8, 8,   4, 12};
const double WT[MG16] = {0.1527534276, 0.1491729617, 0.1420961469, 0.1316886544, 0.1181945205, 0.1019300893,
                         0.0832767040, 0.0626720116, 0.0424925000, 0.0046269894, 0.0038279891, 0.0030260086,
                         0.0022199750, 0.0014140010, 0.0005330000, 0.0000750000};
const int NSPA[LW_NBND] = {1, 1, 9, 9, 9, 1, 9, 1, 9, 1, 1, 9, 9, 1, 9, 9};   // LW:8037
const int NSPB[LW_NBND] = {1, 1, 5, 5, 5, 0, 1, 1, 1, 1, 1, 0, 0, 1, 0, 0};   // LW:8038 (band 16: 0 although kbo exists)
const double DELWAVE[LW_NBND] = {340., 150., 130., 70., 120., 160., 100., 100.,
                                 210., 90., 320., 280., 170., 130., 220., 650.};
const int NFA[LW_NBND] = {1, 1, 9, 9, 9, 1, 9, 1, 9, 1, 1, 9, 9, 1, 9, 9};
const int NFB[LW_NBND] = {1, 1, 5, 5, 5, 0, 1, 1, 1, 1, 1, 0, 1, 1, 0, 1};
const bool HAS_KB[LW_NBND] = {1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 0, 0, 1, 0, 1};

struct Reader {
    const unsigned char *p, *end;
    int rb;
    bool bad = false;
    double real()
    {
        if (p + rb > end) { bad = true; return 0; }
        double v;
        if (rb == 8) std::memcpy(&v, p, 8);
        else { float f; std::memcpy(&f, p, 4); v = f; }
        p += rb;
        return v;
    }
    std::vector<double> arr(size_t n)
    {
        std::vector<double> a(n);
        for (auto &x : a) x = real();
        return a;
    }
};

struct Groups {
    int ng;
    int first[MG16 + 1];
    double rw[MG16];
};
Groups make_groups(int ib)
{
    Groups g;
    g.ng = NGC[ib];
    int g0 = 0;
    for (int b = 0; b < ib; ++b) g0 += NGC[b];
    int pos = 0;
    for (int igc = 0; igc < g.ng; ++igc) { g.first[igc] = pos; pos += NGN[g0 + igc]; }
    g.first[g.ng] = pos;
    for (int igc = 0; igc < g.ng; ++igc) {
        double wtsum = 0.;
        for (int i = g.first[igc]; i < g.first[igc + 1]; ++i) wtsum = wtsum + WT[i];
        for (int i = g.first[igc]; i < g.first[igc + 1]; ++i)
            g.rw[i] = g.ng < MG16 ? WT[i] / wtsum : 1.0;   // LW:7962-7988
    }
    return g;
}
// src [16][nrow] -> dst [nrow][stride] (weighted)
void reduce_rows(const std::vector<double> &src, int nrow, const Groups &g, int stride, double *dst)
{
    for (int r = 0; r < nrow; ++r)
        for (int igc = 0; igc < g.ng; ++igc) {
            double s = 0.;
            for (int i = g.first[igc]; i < g.first[igc + 1]; ++i) s = s + src[(size_t)i * nrow + r] * g.rw[i];
            dst[(size_t)r * stride + igc] = s;
        }
}
// src [nj][16] -> dst [nj][stride]
void reduce_vec(const std::vector<double> &src, int nj, const Groups &g, bool weighted, int stride, double *dst)
{
    for (int j = 0; j < nj; ++j)
        for (int igc = 0; igc < g.ng; ++igc) {
            double s = 0.;
            for (int i = g.first[igc]; i < g.first[igc + 1]; ++i)
                s = weighted ? s + src[j * MG16 + i] * g.rw[i] : s + src[j * MG16 + i];
            dst[j * stride + igc] = s;
        }
}

uint32_t cfg(int type, int spa, int spb, int rat, int nsp, uint32_t flags, int corr, int frac)
{
    return (uint32_t(type) << LC_TYPE_SHIFT) | (uint32_t(spa) << LC_SPA_SHIFT) | (uint32_t(spb) << LC_SPB_SHIFT) |
           (uint32_t(rat) << LC_RAT_SHIFT) | (uint32_t(nsp) << LC_NSP_SHIFT) | flags | (uint32_t(corr) << LC_CORR_SHIFT) |
           (uint32_t(frac) << LC_FRAC_SHIFT);
}
inline double chi(int m, int l) { return LWREF_CHI_MLS[m - 1][l - 1]; }

}  // namespace

int lw_build_tables(const char *path, int real_bytes, double cpdair, LwHostTables &out, std::string &err)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) { err = "module_ra_rrtmg_lw: error opening RRTMG_LW_DATA (" + std::string(path) + ")"; return 1; }
    std::vector<unsigned char> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (real_bytes != 4 && real_bytes != 8) { err = "RRTMG_LW_DATA: real_bytes must be 4 or 8"; return 1; }

    const double grav = 9.8066, secdy = 8.6400e4;     // lwdatinit, LW:8046-8072
    out.oneminus = 1. - 1.e-6;
    out.fluxfac = (2. * std::asin(1.)) * 2.e4;
    out.heatfac = grav * secdy / (cpdair * 1.e2);
    const double pade = 0.278, expeps = 1.e-20;       // LW:7940-7957
    out.bpade = 1.0 / pade;
    out.exp_tbl.assign(LW_NTBL + 1, 0.);
    out.tfn_tbl.assign(LW_NTBL + 1, 0.);
    out.exp_tbl[0] = 1.0; out.exp_tbl[LW_NTBL] = expeps;
    out.tfn_tbl[0] = 0.0; out.tfn_tbl[LW_NTBL] = 1.0;
    for (int itr = 1; itr < LW_NTBL; ++itr) {
        double tfn = double(itr) / double(LW_NTBL);
        double tau = out.bpade * tfn / (1. - tfn);
        double e = std::exp(-tau);
        if (e <= expeps) e = expeps;
        out.exp_tbl[itr] = e;
        out.tfn_tbl[itr] = tau < 0.06 ? tau / 6. : 1. - 2. * ((1. / tau) - (e / (1. - e)));
    }

    const int ZPAD = 1200 * MG16, ZBASE = 64 * MG16;   // absent tables point into the zero pad (row -1 stays inside)
    std::vector<double> &K = out.K;
    K.assign(ZPAD, 0.0);
    out.gdesc.assign(LW_NGPT, LwGDesc{});

    Reader c{buf.data(), buf.data() + buf.size(), real_bytes};
    int g0 = 0;
    for (int ib = 0; ib < LW_NBND; ++ib) {
        const int band = ib + 1;
        const Groups grp = make_groups(ib);
        const int ng = grp.ng;
        // rows are padded to a multiple of 8 doubles and tables start on 64-byte lines: a wavefront's unit-stride read
        // of one band's row then touches one or two cache lines instead of straddling three
        const int ngs = (ng + 7) & ~7;
        const int nrowa = NSPA[ib] * 65, nrowb = HAS_KB[ib] ? (band == 16 ? 235 : NSPB[ib] * 235) : 0;
        if (c.p + 4 > c.end) { err = "RRTMG_LW_DATA: truncated"; return 1; }
        int32_t reclen; std::memcpy(&reclen, c.p, 4); c.p += 4;
        const unsigned char *rec_end = c.p + reclen;
        // READ field orders of lw_kgb01..16, LW:13021 ... 14386
        std::vector<double> fao = c.arr((size_t)MG16 * NFA[ib]);
        std::vector<double> fbo = NFB[ib] ? c.arr((size_t)MG16 * NFB[ib]) : std::vector<double>();
        std::vector<double> kao = c.arr((size_t)nrowa * MG16);
        std::vector<double> kbo = nrowb ? c.arr((size_t)nrowb * MG16) : std::vector<double>();
        std::vector<double> ma[3], mb[2], gv[2];
        int man[3] = {0, 0, 0}, mbn[2] = {0, 0};
        auto M = [&](int n) { return c.arr((size_t)n * 19 * MG16); };
        switch (band) {
        case 1: case 11: ma[0] = M(1); man[0] = 1; mb[0] = M(1); mbn[0] = 1; break;
        case 3: ma[0] = M(9); man[0] = 9; mb[0] = M(5); mbn[0] = 5; break;
        case 5: ma[0] = M(9); man[0] = 9; gv[0] = c.arr(MG16); break;
        case 6: ma[0] = M(1); man[0] = 1; gv[0] = c.arr(MG16); gv[1] = c.arr(MG16); break;
        case 7: case 9: ma[0] = M(9); man[0] = 9; mb[0] = M(1); mbn[0] = 1; break;
        case 8:
            ma[0] = M(1); man[0] = 1; mb[0] = M(1); mbn[0] = 1; ma[1] = M(1); man[1] = 1; mb[1] = M(1); mbn[1] = 1;
            ma[2] = M(1); man[2] = 1; gv[0] = c.arr(MG16); gv[1] = c.arr(MG16);
            break;
        case 13: ma[0] = M(9); man[0] = 9; ma[1] = M(9); man[1] = 9; mb[0] = M(1); mbn[0] = 1; break;
        case 15: ma[0] = M(9); man[0] = 9; break;
        default: break;
        }
        std::vector<double> selfo = c.arr(10 * MG16), foro = c.arr(4 * MG16);
        if (c.bad || c.p != rec_end) {
            char m[128];
            std::snprintf(m, sizeof m, "module_ra_rrtmg_lw: error reading RRTMG_LW_DATA (band %d record)", band);
            err = m;
            return 1;
        }
        c.p += 4;

        auto put_rows = [&](const std::vector<double> &src, int nrow, int pad_rows) -> int {
            if (src.empty()) return 0;
            // one guard row in front: the > 0.875 stencil reads row-1 with zero weight at the first row
            K.resize((K.size() + 7) & ~size_t(7), 0.0);
            K.resize(K.size() + (size_t)ngs, 0.0);
            int off = (int)K.size();
            K.resize(K.size() + (size_t)(nrow + pad_rows) * ngs, 0.0);
            reduce_rows(src, nrow, grp, ngs, &K[off]);
            return off;
        };
        auto put_vec = [&](const std::vector<double> &src, int nj, bool weighted) -> int {
            if (src.empty()) return 0;
            K.resize((K.size() + 7) & ~size_t(7), 0.0);
            int off = (int)K.size();
            K.resize(K.size() + (size_t)nj * ngs, 0.0);
            reduce_vec(src, nj, grp, weighted, ngs, &K[off]);
            return off;
        };
        const int offA = put_rows(kao, nrowa, 3), offB = put_rows(kbo, nrowb, 3);
        const int offS = put_rows(selfo, 10, 1), offF = put_rows(foro, 4, 1);
        const int offFA = put_vec(fao, NFA[ib], false), offFB = put_vec(fbo, NFB[ib], false);
        int offMA[3], offMB[2];
        for (int m = 0; m < 3; ++m) offMA[m] = put_rows(ma[m], man[m] * 19, 12);
        for (int m = 0; m < 2; ++m) offMB[m] = put_rows(mb[m], mbn[m] * 19, 12);
        std::vector<double> g0v(ng, 0.), g1v(ng, 0.);
        if (!gv[0].empty()) reduce_vec(gv[0], 1, grp, true, ng, g0v.data());
        if (!gv[1].empty()) reduce_vec(gv[1], 1, grp, true, ng, g1v.data());

        // post-hoc scaling of individual g-points: band 4 upper (LW:5687-5696), band 7 upper (LW:6323-6331)
        auto postscale = [&](int ig1) -> double {
            if (band == 4) { const double s[7] = {0.92, 0.88, 1.07, 1.1, 0.99, 0.88, 0.943}; return ig1 >= 8 && ig1 <= 14 ? s[ig1 - 8] : 1.0; }
            if (band == 7) { const double s[6] = {0.92, 0.88, 1.07, 1.1, 0.99, 0.855}; return ig1 >= 6 && ig1 <= 11 ? s[ig1 - 6] : 1.0; }
            return 1.0;
        };

        for (int ig = 0; ig < ng; ++ig) {
            LwGDesc &d = out.gdesc[g0 + ig];
            d.band = ib; d.ig = ig; d.ng = ng; d.stride = ngs; d.off_self = offS + ig; d.off_for = offF + ig; d.delwave = DELWAVE[ib];
            LwRegime &L = d.lo, &U = d.up;
            auto init = [&](LwRegime &R) {
                R = LwRegime{};
                R.postscale = 1.0; R.planck_ratio = 1.0; R.off_major = ZBASE + ig; R.off_frac = ZBASE + ig;
                for (int m = 0; m < 3; ++m) R.minor[m] = LwMinor{ZBASE + ig, 1, LS_ZERO, 0, 1.0};
                R.halo_slot[0] = R.halo_slot[1] = LS_ZERO;
            };
            init(L); init(U);
            const uint32_t SF = LC_SELF | LC_FOR;
            const double fa0 = K[offFA + ig], fb0 = offFB ? K[offFB + ig] : 0.;
            auto single_lo = [&](int sp, uint32_t flags, int corr) {
                L.cfg = cfg(1, sp, LS_ZERO, LR_ONE, NSPA[ib], flags, corr, 1);
                L.off_major = offA + ig; L.fracconst = fa0;
            };
            auto single_up = [&](int sp, uint32_t flags, int corr) {
                U.cfg = cfg(1, sp, LS_ZERO, LR_ONE, NSPB[ib], flags, corr, 1);
                U.off_major = offB + ig; U.fracconst = fb0;
            };
            auto binary_lo = [&](int spa, int spb, int rat, double planck) {
                L.cfg = cfg(2, spa, spb, rat, 9, SF | LC_EDGES, 0, 2);
                L.off_major = offA + ig; L.off_frac = offFA + ig; L.planck_ratio = planck;
            };
            auto binary_up = [&](int spa, int spb, int rat, double planck, uint32_t flags) {
                U.cfg = cfg(2, spa, spb, rat, 5, flags | LC_MULT4, 0, 2);
                U.off_major = offB + ig; U.off_frac = offFB + ig; U.planck_ratio = planck;
            };
            auto minor = [&](LwRegime &R, int m, int off, int nsp, int slot, double refrat) {
                R.minor[m] = LwMinor{off + ig, nsp, slot, 0, refrat};
            };
            switch (band) {   // taugb1..16, LW:4860-7826
            case 1:
                single_lo(LS_H2O, SF, 1); minor(L, 0, offMA[0], 1, LS_SCALEN2A, 1.0);
                single_up(LS_H2O, LC_FOR, 2); minor(U, 0, offMB[0], 1, LS_SCALEN2A, 1.0);
                break;
            case 2: single_lo(LS_H2O, SF, 3); single_up(LS_H2O, LC_FOR, 0); break;
            case 3:
                binary_lo(LS_H2O, LS_CO2, LR_H2OCO2, chi(1, 9) / chi(2, 9));
                minor(L, 0, offMA[0], 9, LS_ADJ_N2O, chi(1, 3) / chi(2, 3));
                binary_up(LS_H2O, LS_CO2, LR_H2OCO2, chi(1, 13) / chi(2, 13), LC_FOR);
                minor(U, 0, offMB[0], 5, LS_ADJ_N2O, chi(1, 13) / chi(2, 13));
                break;
            case 4:
                binary_lo(LS_H2O, LS_CO2, LR_H2OCO2, chi(1, 11) / chi(2, 11));
                binary_up(LS_O3, LS_CO2, LR_O3CO2, chi(3, 13) / chi(2, 13), 0);
                U.postscale = postscale(ig + 1);
                break;
            case 5:
                binary_lo(LS_H2O, LS_CO2, LR_H2OCO2, chi(1, 5) / chi(2, 5));
                minor(L, 0, offMA[0], 9, LS_O3, chi(1, 7) / chi(2, 7));
                L.halo_slot[0] = LS_WX1; L.halo_coef[0] = g0v[ig];
                binary_up(LS_O3, LS_CO2, LR_O3CO2, chi(3, 43) / chi(2, 43), 0);
                U.halo_slot[0] = LS_WX1; U.halo_coef[0] = g0v[ig];
                break;
            case 6:
                single_lo(LS_H2O, SF, 0); minor(L, 0, offMA[0], 1, LS_ADJ_CO2_6, 1.0);
                L.halo_slot[0] = LS_WX2; L.halo_coef[0] = g0v[ig]; L.halo_slot[1] = LS_WX3; L.halo_coef[1] = g1v[ig];
                U.cfg = cfg(0, LS_ZERO, LS_ZERO, LR_ONE, 1, 0, 0, 1); U.fracconst = fa0;   // fracs = fracrefa (LW:5932)
                U.halo_slot[0] = LS_WX2; U.halo_coef[0] = g0v[ig]; U.halo_slot[1] = LS_WX3; U.halo_coef[1] = g1v[ig];
                break;
            case 7:
                binary_lo(LS_H2O, LS_O3, LR_H2OO3, chi(1, 3) / chi(3, 3));
                minor(L, 0, offMA[0], 9, LS_ADJ_CO2_7L, chi(1, 3) / chi(3, 3));
                single_up(LS_O3, 0, 0); minor(U, 0, offMB[0], 1, LS_ADJ_CO2_7U, 1.0);
                U.postscale = postscale(ig + 1);
                break;
            case 8:
                single_lo(LS_H2O, SF, 0);
                minor(L, 0, offMA[0], 1, LS_ADJ_CO2_8, 1.0); minor(L, 1, offMA[2], 1, LS_O3, 1.0);
                minor(L, 2, offMA[1], 1, LS_N2O, 1.0);
                L.halo_slot[0] = LS_WX3; L.halo_coef[0] = g0v[ig]; L.halo_slot[1] = LS_WX4; L.halo_coef[1] = g1v[ig];
                single_up(LS_O3, 0, 0);
                minor(U, 0, offMB[0], 1, LS_ADJ_CO2_8, 1.0); minor(U, 1, offMB[1], 1, LS_N2O, 1.0);
                U.halo_slot[0] = LS_WX3; U.halo_coef[0] = g0v[ig]; U.halo_slot[1] = LS_WX4; U.halo_coef[1] = g1v[ig];
                break;
            case 9:
                binary_lo(LS_H2O, LS_CH4, LR_H2OCH4, chi(1, 9) / chi(6, 9));
                minor(L, 0, offMA[0], 9, LS_ADJ_N2O, chi(1, 3) / chi(6, 3));
                single_up(LS_CH4, 0, 0); minor(U, 0, offMB[0], 1, LS_ADJ_N2O, 1.0);
                break;
            case 10: single_lo(LS_H2O, SF, 0); single_up(LS_H2O, LC_FOR, 0); break;
            case 11:
                single_lo(LS_H2O, SF, 0); minor(L, 0, offMA[0], 1, LS_SCALEO2, 1.0);
                single_up(LS_H2O, LC_FOR, 0); minor(U, 0, offMB[0], 1, LS_SCALEO2, 1.0);
                break;
            case 12:
                binary_lo(LS_H2O, LS_CO2, LR_H2OCO2, chi(1, 10) / chi(2, 10));
                U.cfg = cfg(0, LS_ZERO, LS_ZERO, LR_ONE, 1, 0, 0, 0);
                break;
            case 13:
                binary_lo(LS_H2O, LS_N2O, LR_H2ON2O, chi(1, 5) / chi(4, 5));
                minor(L, 0, offMA[0], 9, LS_ADJ_CO2_13, chi(1, 1) / chi(4, 1));
                minor(L, 1, offMA[1], 9, LS_CO, chi(1, 3) / chi(4, 3));
                U.cfg = cfg(0, LS_ZERO, LS_ZERO, LR_ONE, 1, 0, 0, 1); U.fracconst = fb0;
                minor(U, 0, offMB[0], 1, LS_O3, 1.0);
                break;
            case 14: single_lo(LS_CO2, SF, 0); single_up(LS_CO2, 0, 0); break;
            case 15:
                binary_lo(LS_N2O, LS_CO2, LR_N2OCO2, chi(4, 1) / chi(2, 1));
                minor(L, 0, offMA[0], 9, LS_SCALEN2B, chi(4, 1) / chi(2, 1));
                U.cfg = cfg(0, LS_ZERO, LS_ZERO, LR_ONE, 1, 0, 0, 0);
                break;
            case 16:
                binary_lo(LS_H2O, LS_CH4, LR_H2OCH4, chi(1, 6) / chi(6, 6));
                single_up(LS_CH4, 0, 0);   // nspb(16) = 0: rows 1..2 of absb only, as in the reference
                break;
            }
        }
        g0 += ng;
    }
    K.resize(K.size() + 64 * MG16, 0.0);
    out.totplnk.assign(&LWREF_TOTPLNK[0][0], &LWREF_TOTPLNK[0][0] + 16 * 181);
    out.chi_mls.assign(&LWREF_CHI_MLS[0][0], &LWREF_CHI_MLS[0][0] + 7 * 59);
    // behind the 7 x 59 mixing ratios: the six species ratios setcoef forms per layer (LW:3713-3726, 3769-3772),
    // tabulated per reference level so that the kernel reads a quotient instead of dividing -- [LR_*][59], the same
    // IEEE quotients
    {
        static const int num[6] = {1, 1, 1, 1, 4, 3}, den[6] = {2, 3, 4, 6, 2, 2};   // LR_H2OCO2 .. LR_O3CO2
        for (int r = 0; r < 6; ++r)
            for (int l = 0; l < 59; ++l)
                out.chi_mls.push_back(LWREF_CHI_MLS[num[r] - 1][l] / LWREF_CHI_MLS[den[r] - 1][l]);
    }
    return 0;
}

}  // namespace rrtmg
