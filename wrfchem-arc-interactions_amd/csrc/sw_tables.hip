// Host side of the shortwave init: RRTMG_SW_DATA loader, 224->112 g-point
// reduction, packing into the device layout, upload.
//
// Replaces rrtmg_swinit / rrtmg_swlookuptable / sw_kgb16..29 / rrtmg_sw_ini
// (phys/module_ra_rrtmg_sw.F:11211-12366, 4571-4698, cmbgb16s..29 5022-6065).
// The reference keeps one Fortran module per band with absa(ind,ig) (ind
// fastest); here every table of a band is stored [row][ig] (ig fastest) so a
// wavefront whose lanes are g-points reads a row with unit stride, and all
// bands live in one allocation addressed by per-g-point offsets.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "rrtmg_common.h"
#include "rrtmg_state.h"

namespace rrtmg {

// g-point combination maps (SW:4812-4870, swcmbdat)
static const int NGC[SW_NBND] = {6, 12, 8, 8, 10, 10, 2, 10, 8, 6, 6, 8, 6, 12};
static const int NGN[SW_NGPT] = {
    2, 2, 2, 2, 4, 4,  1, 1, 1, 1, 1, 2, 1, 2, 1, 2, 1, 2,  1, 1, 1, 1, 2, 2, 4, 4,  1, 1, 1, 1, 2, 2, 4, 4,
    1, 1, 1, 1, 1, 1, 1, 1, 2, 6,  1, 1, 1, 1, 1, 1, 1, 1, 2, 6,  8, 8,  2, 2, 1, 1, 1, 1, 1, 1, 2, 4,
    2, 2, 2, 2, 2, 2, 2, 2,  1, 1, 2, 2, 4, 6,  1, 1, 2, 2, 4, 6,  1, 1, 1, 1, 1, 1, 4, 6,  1, 1, 2, 2, 4, 6,
    1, 1, 1, 1, 2, 2, 2, 2, 1, 1, 1, 1};
static const double WT[MG] = {0.1527534276, 0.1491729617, 0.1420961469, 0.1316886544, 0.1181945205, 0.1019300893,
                              0.0832767040, 0.0626720116, 0.0424925000, 0.0046269894, 0.0038279891, 0.0030260086,
                              0.0022199750, 0.0014140010, 0.0005330000, 0.0000750000};
static const int NSPA[SW_NBND] = {9, 9, 9, 9, 1, 9, 9, 1, 9, 1, 0, 1, 9, 1};   // SW:4724
static const int NSPB[SW_NBND] = {1, 5, 1, 1, 1, 5, 1, 0, 1, 0, 0, 1, 5, 1};   // SW:4725
static const int NFOR[SW_NBND] = {3, 4, 3, 3, 4, 4, 3, 3, 3, 0, 0, 0, 0, 4};
static const int NSF[SW_NBND] = {1, 5, 9, 9, 1, 9, 9, 1, 9, 1, 1, 1, 5, 1};

namespace {

struct Reader {
    const unsigned char *p, *end;
    int rb;
    bool bad = false;
    double real() {
        if (p + rb > end) { bad = true; return 0; }
        double v;
        if (rb == 8) std::memcpy(&v, p, 8);
        else { float f; std::memcpy(&f, p, 4); v = f; }
        p += rb;
        return v;
    }
    int integer() {
        if (p + 4 > end) { bad = true; return 0; }
        int32_t v; std::memcpy(&v, p, 4); p += 4;
        return v;
    }
    std::vector<double> arr(size_t n) {
        std::vector<double> a(n);
        for (auto &x : a) x = real();
        return a;
    }
};

// original-g group boundaries of band ib: new g igc <- original [first[igc], first[igc+1])
struct Groups {
    int ng;
    int first[MG + 1];
    double rw[MG];   // rwgt(ib*16 + ig) = wt(ig)/sum of wt over its group (SW:4655-4681)
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
        for (int i = g.first[igc]; i < g.first[igc + 1]; ++i) g.rw[i] = WT[i] / wtsum;
    }
    return g;
}

// src is [16][nrow] (Fortran x(row..., ig)); result [nrow][ng] weighted (cmbgbNN k loops)
void reduce_rows(const std::vector<double> &src, int nrow, const Groups &g, double *dst, int dst_stride)
{
    for (int r = 0; r < nrow; ++r)
        for (int igc = 0; igc < g.ng; ++igc) {
            double s = 0.;
            for (int i = g.first[igc]; i < g.first[igc + 1]; ++i) s = s + src[(size_t)i * nrow + r] * g.rw[i];
            dst[(size_t)r * dst_stride + igc] = s;
        }
}
// src is [nj][16] (Fortran x(ig, j)); result [nj][ng]; weighted or plain sum (sfluxref)
void reduce_vec(const std::vector<double> &src, int nj, const Groups &g, bool weighted, double *dst)
{
    for (int j = 0; j < nj; ++j)
        for (int igc = 0; igc < g.ng; ++igc) {
            double s = 0.;
            for (int i = g.first[igc]; i < g.first[igc + 1]; ++i)
                s = weighted ? s + src[j * MG + i] * g.rw[i] : s + src[j * MG + i];
            dst[j * g.ng + igc] = s;
        }
}

uint32_t cfg(int spa, int spb, int spm, int nsp, uint32_t flags)
{
    return (uint32_t(spa) << CFG_SPA_SHIFT) | (uint32_t(spb) << CFG_SPB_SHIFT) | (uint32_t(spm) << CFG_SPM_SHIFT) |
           (uint32_t(nsp) << CFG_NSP_SHIFT) | flags;
}

}  // namespace

int sw_build_tables(const char *path, int real_bytes, double cpdair, SwHostTables &out, std::string &err)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) {
        err = "module_ra_rrtmg_sw: error opening RRTMG_SW_DATA (" + std::string(path) + ")";   // cf. SW:11291
        return 1;
    }
    std::vector<unsigned char> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (real_bytes != 4 && real_bytes != 8) { err = "RRTMG_SW_DATA: real_bytes must be 4 or 8"; return 1; }

    const double grav = 9.8066, secdy = 8.6400e4;            // SW:4732, SW:4749
    out.oneminus = 1.0 - 1.e-06;                             // SW:4756
    out.heatfac = grav * secdy / (cpdair * 1.e2);            // SW:4788
    const double pade = 0.278, expeps = 1.e-20;              // SW:1229, SW:4596
    out.bpade = 1.0 / pade;
    out.exp_tbl.assign(NTBL + 1, 0.0);
    out.exp_tbl[0] = 1.0;
    out.exp_tbl[NTBL] = expeps;
    for (int itr = 1; itr < NTBL; ++itr) {                   // SW:4643-4648
        double tfn = double(itr) / double(NTBL);
        double tau_tbl = out.bpade * tfn / (1. - tfn);
        double e = std::exp(-tau_tbl);
        out.exp_tbl[itr] = e <= expeps ? expeps : e;
    }

    // zero pad large enough for any generic (row, ig) read of a band without that table
    const int ZPAD = 1200 * MG;
    std::vector<double> &K = out.K;
    K.assign(ZPAD, 0.0);
    out.gdesc.assign(SW_NGPT, GDesc{});

    Reader c{buf.data(), buf.data() + buf.size(), real_bytes};
    int g0 = 0;
    for (int ib = 0; ib < SW_NBND; ++ib) {
        const int band = ib + SW_JPB1;
        const Groups grp = make_groups(ib);
        const int ng = grp.ng;
        const int nrowa = NSPA[ib] * 65, nrowb = NSPB[ib] * 235;
        if (c.p + 4 > c.end) { err = "RRTMG_SW_DATA: truncated"; return 1; }
        int reclen = c.integer();
        const unsigned char *rec_end = c.p + reclen;
        double rayl = 0, strrat = 0, givfac = 1, scalekur = 1;
        int layreffr = 0;
        std::vector<double> kao, kbo, selfo, foro, sflo, raylo, raylao, raylbo, o3ao, o3bo, ch4o, h2oo, co2o;
        switch (band) {   // READ field order, SW:11379 ... SW:12365
        case 16: case 17: case 18: case 19: case 21: case 22: case 28:
            rayl = c.real(); strrat = c.real(); layreffr = c.integer();
            kao = c.arr((size_t)nrowa * MG); kbo = c.arr((size_t)nrowb * MG);
            if (band != 28) { selfo = c.arr(10 * MG); foro = c.arr(NFOR[ib] * MG); }
            sflo = c.arr(MG * NSF[ib]);
            break;
        case 20:
            rayl = c.real(); layreffr = c.integer(); ch4o = c.arr(MG);
            kao = c.arr((size_t)nrowa * MG); kbo = c.arr((size_t)nrowb * MG);
            selfo = c.arr(10 * MG); foro = c.arr(NFOR[ib] * MG); sflo = c.arr(MG);
            break;
        case 23:
            raylo = c.arr(MG); givfac = c.real(); layreffr = c.integer();
            kao = c.arr((size_t)nrowa * MG); selfo = c.arr(10 * MG); foro = c.arr(NFOR[ib] * MG); sflo = c.arr(MG);
            break;
        case 24:
            raylao = c.arr(MG * 9); raylbo = c.arr(MG); strrat = c.real(); layreffr = c.integer();
            o3ao = c.arr(MG); o3bo = c.arr(MG);
            kao = c.arr((size_t)nrowa * MG); kbo = c.arr((size_t)nrowb * MG);
            selfo = c.arr(10 * MG); foro = c.arr(NFOR[ib] * MG); sflo = c.arr(MG * 9);
            break;
        case 25:
            raylo = c.arr(MG); layreffr = c.integer(); o3ao = c.arr(MG); o3bo = c.arr(MG);
            kao = c.arr((size_t)nrowa * MG); sflo = c.arr(MG);
            break;
        case 26:
            raylo = c.arr(MG); sflo = c.arr(MG);
            break;
        case 27:
            raylo = c.arr(MG); scalekur = c.real(); layreffr = c.integer();
            kao = c.arr((size_t)nrowa * MG); kbo = c.arr((size_t)nrowb * MG); sflo = c.arr(MG);
            break;
        case 29:
            rayl = c.real(); layreffr = c.integer(); h2oo = c.arr(MG); co2o = c.arr(MG);
            kao = c.arr((size_t)nrowa * MG); kbo = c.arr((size_t)nrowb * MG);
            selfo = c.arr(10 * MG); foro = c.arr(NFOR[ib] * MG); sflo = c.arr(MG);
            break;
        }
        if (c.bad || c.p != rec_end) {
            char m[128];
            std::snprintf(m, sizeof m, "module_ra_rrtmg_sw: error reading RRTMG_SW_DATA (band %d record)", band);
            err = m;
            return 1;
        }
        c.p += 4;

        auto put_rows = [&](const std::vector<double> &src, int nrow, int nrow_padded) -> int {
            if (src.empty()) return 0;   // zero pad
            int off = (int)K.size();
            K.resize(K.size() + (size_t)nrow_padded * ng, 0.0);
            reduce_rows(src, nrow, grp, &K[off], ng);
            return off;
        };
        auto put_vec = [&](const std::vector<double> &src, int nj, bool weighted) -> int {
            if (src.empty()) return 0;
            int off = (int)K.size();
            K.resize(K.size() + (size_t)nj * ng, 0.0);
            reduce_vec(src, nj, grp, weighted, &K[off]);
            return off;
        };
        // +2 spare rows: the generic 8-term stencil of a single-species band reads row+2 with weight 0
        const int offA = put_rows(kao, nrowa, nrowa + 2);
        const int offB = put_rows(kbo, nrowb, nrowb + 2);
        const int offS = put_rows(selfo, 10, 11);
        const int offF = put_rows(foro, NFOR[ib], 5);
        const int offSF = put_vec(sflo, NSF[ib], false);
        const int offRA = put_vec(raylao, 9, true);
        std::vector<double> raylg(ng, rayl), raylb(ng, rayl), o3a(ng, 0.), o3b(ng, 0.), ch4(ng, 0.), h2o(ng, 0.),
            co2(ng, 0.);
        if (!raylo.empty()) { reduce_vec(raylo, 1, grp, true, raylg.data()); raylb = raylg; }
        if (!raylbo.empty()) reduce_vec(raylbo, 1, grp, true, raylb.data());
        if (!o3ao.empty()) reduce_vec(o3ao, 1, grp, true, o3a.data());
        if (!o3bo.empty()) reduce_vec(o3bo, 1, grp, true, o3b.data());
        if (!ch4o.empty()) reduce_vec(ch4o, 1, grp, true, ch4.data());
        if (!h2oo.empty()) reduce_vec(h2oo, 1, grp, true, h2o.data());
        if (!co2o.empty()) reduce_vec(co2o, 1, grp, true, co2.data());

        const double o2adj = 1.6;   // SW:3927
        for (int ig = 0; ig < ng; ++ig) {
            GDesc &d = out.gdesc[g0 + ig];
            d.band = ib; d.ig = ig; d.ng = ng;
            d.off_self = offS + ig; d.off_for = offF + ig; d.off_sflux = offSF + ig; d.off_rayla = offRA + ig;
            d.nsf = NSF[ib]; d.layreffr = layreffr; d.scalekur = scalekur;
            d.uvband = (ib + 1 >= 10 && ib + 1 <= 13) ? 1 : 0;
            d.solar_kind = (band == 16 || band == 17 || band == 27 || band == 28 || band == 29) ? SOL_UPPER
                           : (band == 26) ? SOL_LAYTROP : SOL_LOWER;
            GRegime &L = d.lo, &U = d.up;
            L = GRegime{0, offA + ig, strrat, 1.0, 0.0, raylg[ig]};
            U = GRegime{0, offB + ig, strrat, 1.0, 0.0, raylb[ig]};
            const uint32_t SF = CFG_SELF | CFG_FOR, BIN = CFG_BINARY | CFG_MAJOR, ONE = CFG_MAJOR;
            switch (band) {   // roles of the species: taumol16..29, SW:3293-4540
            case 16:
                L.cfg = cfg(SP_H2O, SP_CH4, SP_ZERO, 9, BIN | SF);
                U.cfg = cfg(SP_CH4, SP_ZERO, SP_ZERO, 1, ONE); break;
            case 17:
                L.cfg = cfg(SP_H2O, SP_CO2, SP_ZERO, 9, BIN | SF);
                U.cfg = cfg(SP_H2O, SP_CO2, SP_ZERO, 5, BIN | CFG_MULT4 | CFG_FOR); break;
            case 18:
                L.cfg = cfg(SP_H2O, SP_CH4, SP_ZERO, 9, BIN | SF);
                U.cfg = cfg(SP_CH4, SP_ZERO, SP_ZERO, 1, ONE); break;
            case 19:
                L.cfg = cfg(SP_H2O, SP_CO2, SP_ZERO, 9, BIN | SF);
                U.cfg = cfg(SP_CO2, SP_ZERO, SP_ZERO, 1, ONE); break;
            case 20:
                L.cfg = cfg(SP_H2O, SP_ZERO, SP_CH4, 1, ONE | SF); L.minorabs = ch4[ig];
                U.cfg = cfg(SP_H2O, SP_ZERO, SP_CH4, 1, ONE | CFG_FOR); U.minorabs = ch4[ig]; break;
            case 21:
                L.cfg = cfg(SP_H2O, SP_CO2, SP_ZERO, 9, BIN | SF);
                U.cfg = cfg(SP_H2O, SP_CO2, SP_ZERO, 5, BIN | CFG_MULT4 | CFG_FOR); break;
            case 22:
                L.cfg = cfg(SP_H2O, SP_O2, SP_ZERO, 9, BIN | SF | CFG_O2CONT); L.strrat = o2adj * strrat;
                U.cfg = cfg(SP_O2, SP_ZERO, SP_ZERO, 1, ONE | CFG_O2CONT); U.mfac = o2adj; break;
            case 23:
                L.cfg = cfg(SP_H2O, SP_ZERO, SP_ZERO, 1, ONE | SF); L.mfac = givfac;
                U.cfg = cfg(SP_ZERO, SP_ZERO, SP_ZERO, 1, 0); break;
            case 24:
                L.cfg = cfg(SP_H2O, SP_O2, SP_O3, 9, BIN | SF | CFG_RAYL24); L.minorabs = o3a[ig];
                U.cfg = cfg(SP_O2, SP_ZERO, SP_O3, 1, ONE); U.minorabs = o3b[ig]; break;
            case 25:
                L.cfg = cfg(SP_H2O, SP_ZERO, SP_O3, 1, ONE); L.minorabs = o3a[ig];
                U.cfg = cfg(SP_ZERO, SP_ZERO, SP_O3, 1, 0); U.minorabs = o3b[ig]; break;
            case 26:
                L.cfg = cfg(SP_ZERO, SP_ZERO, SP_ZERO, 1, 0);
                U.cfg = cfg(SP_ZERO, SP_ZERO, SP_ZERO, 1, 0); break;
            case 27:
                L.cfg = cfg(SP_O3, SP_ZERO, SP_ZERO, 1, ONE);
                U.cfg = cfg(SP_O3, SP_ZERO, SP_ZERO, 1, ONE); break;
            case 28:
                L.cfg = cfg(SP_O3, SP_O2, SP_ZERO, 9, BIN);
                U.cfg = cfg(SP_O3, SP_O2, SP_ZERO, 5, BIN | CFG_MULT4); break;
            case 29:
                L.cfg = cfg(SP_H2O, SP_ZERO, SP_CO2, 1, ONE | SF); L.minorabs = co2[ig];
                U.cfg = cfg(SP_CO2, SP_ZERO, SP_H2O, 1, ONE); U.minorabs = h2o[ig]; break;
            }
            if (!(L.cfg & CFG_MAJOR)) L.off_major = ig;   // zero pad
            if (!(U.cfg & CFG_MAJOR)) U.off_major = ig;
        }
        g0 += ng;
    }
    // tail pad so that stencil reads past the last table stay inside the allocation
    K.resize(K.size() + 32 * MG, 0.0);
    return 0;
}

}  // namespace rrtmg
