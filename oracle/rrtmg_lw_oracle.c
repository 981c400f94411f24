/*
 * oracle/rrtmg_lw_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, one-column, fp64 restatement of the reference longwave path
 * (WRF-Chem v3.9.1 patch, phys/module_ra_rrtmg_lw.F, cited as LW:line):
 * rrtmg_lw_ini + cmbgb1..16 (LW:7862-8009, 8260-10500), inatm (LW:11248-11340),
 * setcoef (LW:3444-3809), taugb1..16 (LW:4712-7828), rtrnmc (LW:2974-3410),
 * rrtmg_lw with the gas-only (clean) and gas+aerosol calls (LW:10942-11062) and
 * the RRTMG_LWRAD wrapper arithmetic (LW:11876-12692).  Cloud-free columns.
 * Expression order follows the Fortran so results can be compared bitwise with
 * the reference built by oracle/build_ref.sh.
 */
#include "rrtmg_oracle.h"
#include "lw_ref_tables.h"

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MG 16
#define MAXLAY 256
#define NTBL 10000

/* lwcmbdat, LW:8079-8170 */
static const int ngc[LW_NBND] = {10, 12, 16, 14, 16, 8, 12, 8, 12, 6, 8, 8, 4, 2, 2, 2};
static const int ngs[LW_NBND] = {10, 22, 38, 52, 68, 76, 88, 96, 108, 114, 122, 130, 134, 136, 138, 140};
static const int ngm[LW_NBND * MG] = {
    1, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 10,            1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 10, 10, 11, 11, 12, 12,
    1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16,      1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 14, 14,
    1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16,      1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8,
    1, 1, 2, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 11, 12, 12,        1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8,
    1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 10, 10, 11, 11, 12, 12,       1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 5, 5, 6, 6, 6, 6,
    1, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 7, 8, 8, 8,             1, 2, 3, 4, 5, 5, 6, 6, 7, 7, 7, 7, 8, 8, 8, 8,
    1, 1, 1, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 4, 4,             1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2,
    1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2,             1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2};
static const int ngn[LW_NGPT] = {
    1, 1, 2, 2, 2, 2, 2, 2, 1, 1,   1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2,   1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
    1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 3,   1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,   2, 2, 2, 2, 2, 2, 2, 2,
    2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2,   2, 2, 2, 2, 2, 2, 2, 2,   1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2,   2, 2, 2, 2, 4, 4,
    1, 1, 2, 2, 2, 2, 3, 3,   1, 1, 1, 1, 2, 2, 4, 4,   3, 3, 4, 6,   8, 8,   8, 8,   4, 12};
static const double wt[MG] = {
    0.1527534276, 0.1491729617, 0.1420961469, 0.1316886544, 0.1181945205, 0.1019300893,
    0.0832767040, 0.0626720116, 0.0424925000, 0.0046269894, 0.0038279891, 0.0030260086,
    0.0022199750, 0.0014140010, 0.0005330000, 0.0000750000};
static const int nspa[LW_NBND] = {1, 1, 9, 9, 9, 1, 9, 1, 9, 1, 1, 9, 9, 1, 9, 9};   /* LW:8037 */
static const int nspb[LW_NBND] = {1, 1, 5, 5, 5, 0, 1, 1, 1, 1, 1, 0, 0, 1, 0, 0};   /* LW:8038 */
static const double delwave[LW_NBND] = {340., 150., 130., 70., 120., 160., 100., 100.,
                                        210., 90., 320., 280., 170., 130., 220., 650.};   /* LW:8031 */
static const int nfa_b[LW_NBND] = {1, 1, 9, 9, 9, 1, 9, 1, 9, 1, 1, 9, 9, 1, 9, 9};
static const int nfb_b[LW_NBND] = {1, 1, 5, 5, 5, 0, 1, 1, 1, 1, 1, 0, 1, 1, 0, 1};

/* minor-species tables: up to 3 "a" and 2 "b" tables per band; nm = 9/5 (binary) or 1 */
typedef struct {
    int ng, nrowa, nrowb, nfa, nfb;
    double *absa, *absb, *selfref, *forref, *fracrefa, *fracrefb;
    double *kam[3], *kbm[2];
    int kam_n[3], kbm_n[2];   /* species dimension of each minor table (1, 5 or 9) */
    double *gv[2];            /* per-g halocarbon coefficients: band 5 ccl4; 6 cfc11adj,cfc12; 8 cfc12,cfc22adj */
} LwBand;

struct LwTables {
    LwBand b[LW_NBND];
    double rwgt[LW_NBND * MG];
    double tau_tbl[NTBL + 1], exp_tbl[NTBL + 1], tfn_tbl[NTBL + 1];
    double bpade, heatfac, fluxfac, oneminus;
};

typedef struct { const unsigned char *p, *end; int rb, bad; } Cur;
static double rd_real(Cur *c)
{
    if (c->p + c->rb > c->end) { c->bad = 1; return 0.0; }
    double v;
    if (c->rb == 8) memcpy(&v, c->p, 8); else { float f; memcpy(&f, c->p, 4); v = f; }
    c->p += c->rb;
    return v;
}
static double *rd_arr(Cur *c, size_t n)
{
    double *a = (double *)malloc((n ? n : 1) * sizeof(double));
    for (size_t i = 0; i < n; ++i) a[i] = rd_real(c);
    return a;
}
/* [16][nrow] -> [ng][nrow], weighted (all k-type tables; cmbgbN) */
static double *reduce_w(const double *src, int nrow, int ibnd, const double *rwgt)
{
    int ng = ngc[ibnd], g0 = ibnd == 0 ? 0 : ngs[ibnd - 1];
    double *dst = (double *)malloc((size_t)ng * nrow * sizeof(double));
    for (int r = 0; r < nrow; ++r) {
        int iprsm = 0;
        for (int igc = 0; igc < ng; ++igc) {
            double sumk = 0.;
            for (int ipr = 0; ipr < ngn[g0 + igc]; ++ipr) {
                sumk = sumk + src[(size_t)iprsm * nrow + r] * rwgt[iprsm + ibnd * MG];
                ++iprsm;
            }
            dst[(size_t)igc * nrow + r] = sumk;
        }
    }
    return dst;
}
/* [nj][16] -> [nj][ng]; weighted (ccl4, cfc) or plain (fracref) */
static double *reduce_g(const double *src, int nj, int ibnd, const double *rwgt, int weighted)
{
    int ng = ngc[ibnd], g0 = ibnd == 0 ? 0 : ngs[ibnd - 1];
    double *dst = (double *)malloc((size_t)ng * (nj ? nj : 1) * sizeof(double));
    for (int j = 0; j < nj; ++j) {
        int iprsm = 0;
        for (int igc = 0; igc < ng; ++igc) {
            double sumf = 0.;
            for (int ipr = 0; ipr < ngn[g0 + igc]; ++ipr) {
                sumf = weighted ? sumf + src[j * MG + iprsm] * rwgt[iprsm + ibnd * MG] : sumf + src[j * MG + iprsm];
                ++iprsm;
            }
            dst[j * ng + igc] = sumf;
        }
    }
    return dst;
}

/* minor tables per band in READ order (lw_kgb01..16, LW:13021-14387): species dim of a-tables / b-tables */
static const int kam_dims[LW_NBND][3] = {{1, 0, 0}, {0, 0, 0}, {9, 0, 0}, {0, 0, 0}, {9, 0, 0}, {1, 0, 0}, {9, 0, 0},
                                         {1, 1, 1}, {9, 0, 0}, {0, 0, 0}, {1, 0, 0}, {0, 0, 0}, {9, 9, 0}, {0, 0, 0},
                                         {9, 0, 0}, {0, 0, 0}};
static const int kbm_dims[LW_NBND][2] = {{1, 0}, {0, 0}, {5, 0}, {0, 0}, {0, 0}, {0, 0}, {1, 0}, {1, 1},
                                         {1, 0}, {0, 0}, {1, 0}, {0, 0}, {1, 0}, {0, 0}, {0, 0}, {0, 0}};

LwTables *lw_oracle_load(const char *path, int real_bytes, double cpdair, char *err, int errlen)
{
    FILE *f = fopen(path, "rb");
    if (!f) { if (err) snprintf(err, errlen, "cannot open %s", path); return NULL; }
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    unsigned char *buf = (unsigned char *)malloc(sz);
    if (fread(buf, 1, sz, f) != (size_t)sz) { fclose(f); free(buf); return NULL; }
    fclose(f);
    LwTables *t = (LwTables *)calloc(1, sizeof(LwTables));
    const double grav = 9.8066, secdy = 8.6400e4;      /* lwdatinit, LW:8046-8072 */
    t->oneminus = 1. - 1.e-6;
    double pi = 2. * asin(1.);
    t->fluxfac = pi * 2.e4;
    t->heatfac = grav * secdy / (cpdair * 1.e2);
    const double pade = 0.278, expeps = 1.e-20;        /* LW:7940-7957 */
    t->tau_tbl[0] = 0.0; t->tau_tbl[NTBL] = 1.e10;
    t->exp_tbl[0] = 1.0; t->exp_tbl[NTBL] = expeps;
    t->tfn_tbl[0] = 0.0; t->tfn_tbl[NTBL] = 1.0;
    t->bpade = 1.0 / pade;
    for (int itr = 1; itr <= NTBL - 1; ++itr) {
        double tfn = (double)itr / (double)NTBL;
        t->tau_tbl[itr] = t->bpade * tfn / (1. - tfn);
        t->exp_tbl[itr] = exp(-t->tau_tbl[itr]);
        if (t->exp_tbl[itr] <= expeps) t->exp_tbl[itr] = expeps;
        if (t->tau_tbl[itr] < 0.06)
            t->tfn_tbl[itr] = t->tau_tbl[itr] / 6.;
        else
            t->tfn_tbl[itr] = 1. - 2. * ((1. / t->tau_tbl[itr]) - (t->exp_tbl[itr] / (1. - t->exp_tbl[itr])));
    }
    {   /* rwgt, LW:7962-7988 */
        int igcsm = 0;
        for (int ibnd = 0; ibnd < LW_NBND; ++ibnd) {
            int iprsm = 0;
            double wtsm[MG];
            if (ngc[ibnd] < MG) {
                for (int igc = 0; igc < ngc[ibnd]; ++igc) {
                    double wtsum = 0.;
                    for (int ipr = 0; ipr < ngn[igcsm]; ++ipr) { wtsum = wtsum + wt[iprsm]; ++iprsm; }
                    ++igcsm;
                    wtsm[igc] = wtsum;
                }
                for (int ig = 0; ig < MG; ++ig) t->rwgt[ibnd * MG + ig] = wt[ig] / wtsm[ngm[ibnd * MG + ig] - 1];
            } else {
                for (int ig = 0; ig < MG; ++ig) { ++igcsm; t->rwgt[ibnd * MG + ig] = 1.0; }
            }
        }
    }
    Cur c = {buf, buf + sz, real_bytes, 0};
    for (int ib = 0; ib < LW_NBND; ++ib) {
        LwBand *b = &t->b[ib];
        b->ng = ngc[ib];
        b->nrowa = nspa[ib] * 65;
        /* band 16 ships kbo(5,13:59,16) although nspb(16) = 0 (LW:8038): the index arithmetic of
         * taugb16 then always lands on rows 1..2 -- reproduced as is */
        b->nrowb = (ib == 15) ? 235 : nspb[ib] * 235;
        b->nfa = nfa_b[ib];
        b->nfb = nfb_b[ib];
        if (c.p + 4 > c.end) { c.bad = 1; break; }
        int32_t reclen; memcpy(&reclen, c.p, 4); c.p += 4;
        const unsigned char *rec_end = c.p + reclen;
        /* common record shape: fracrefao [, fracrefbo], kao [, kbo], minors (a,b interleaved as in the READ), gvecs, selfrefo, forrefo */
        double *fao = rd_arr(&c, (size_t)MG * b->nfa);
        double *fbo = b->nfb ? rd_arr(&c, (size_t)MG * b->nfb) : NULL;
        double *kao = rd_arr(&c, (size_t)b->nrowa * MG);
        double *kbo = b->nrowb ? rd_arr(&c, (size_t)b->nrowb * MG) : NULL;
        double *ma[3] = {0, 0, 0}, *mb[2] = {0, 0}, *gv[2] = {0, 0};
        switch (ib + 1) {   /* READ field orders, LW:13021 ... 14386 */
        case 1: case 3: case 7: case 9: case 11:
            ma[0] = rd_arr(&c, (size_t)kam_dims[ib][0] * 19 * MG); mb[0] = rd_arr(&c, (size_t)kbm_dims[ib][0] * 19 * MG);
            break;
        case 5: ma[0] = rd_arr(&c, 9 * 19 * MG); gv[0] = rd_arr(&c, MG); break;
        case 6: ma[0] = rd_arr(&c, 19 * MG); gv[0] = rd_arr(&c, MG); gv[1] = rd_arr(&c, MG); break;
        case 8:   /* kao_mco2, kbo_mco2, kao_mn2o, kbo_mn2o, kao_mo3, cfc12o, cfc22adjo */
            ma[0] = rd_arr(&c, 19 * MG); mb[0] = rd_arr(&c, 19 * MG); ma[1] = rd_arr(&c, 19 * MG);
            mb[1] = rd_arr(&c, 19 * MG); ma[2] = rd_arr(&c, 19 * MG); gv[0] = rd_arr(&c, MG); gv[1] = rd_arr(&c, MG);
            break;
        case 13:  /* kao_mco2, kao_mco, kbo_mo3 */
            ma[0] = rd_arr(&c, 9 * 19 * MG); ma[1] = rd_arr(&c, 9 * 19 * MG); mb[0] = rd_arr(&c, 19 * MG);
            break;
        case 15: ma[0] = rd_arr(&c, 9 * 19 * MG); break;
        default: break;
        }
        double *selfo = rd_arr(&c, 10 * MG), *foro = rd_arr(&c, 4 * MG);
        if (c.bad || c.p != rec_end) {
            if (err) snprintf(err, errlen, "RRTMG_LW_DATA: bad record for band %d", ib + 1);
            free(buf); return NULL;
        }
        c.p += 4;
        b->absa = reduce_w(kao, b->nrowa, ib, t->rwgt);
        if (kbo) b->absb = reduce_w(kbo, b->nrowb, ib, t->rwgt);
        b->selfref = reduce_w(selfo, 10, ib, t->rwgt);
        b->forref = reduce_w(foro, 4, ib, t->rwgt);
        b->fracrefa = reduce_g(fao, b->nfa, ib, t->rwgt, 0);
        if (fbo) b->fracrefb = reduce_g(fbo, b->nfb, ib, t->rwgt, 0);
        for (int m = 0; m < 3; ++m)
            if (ma[m]) { b->kam_n[m] = kam_dims[ib][m]; b->kam[m] = reduce_w(ma[m], kam_dims[ib][m] * 19, ib, t->rwgt); }
        for (int m = 0; m < 2; ++m)
            if (mb[m]) { b->kbm_n[m] = kbm_dims[ib][m]; b->kbm[m] = reduce_w(mb[m], kbm_dims[ib][m] * 19, ib, t->rwgt); }
        for (int m = 0; m < 2; ++m)
            if (gv[m]) b->gv[m] = reduce_g(gv[m], 1, ib, t->rwgt, 1);
        free(fao); free(fbo); free(kao); free(kbo); free(selfo); free(foro);
        for (int m = 0; m < 3; ++m) free(ma[m]);
        for (int m = 0; m < 2; ++m) { free(mb[m]); free(gv[m]); }
    }
    free(buf);
    if (c.bad) { if (err) snprintf(err, errlen, "RRTMG_LW_DATA: truncated"); return NULL; }
    return t;
}

void lw_oracle_free(LwTables *t)
{
    if (!t) return;
    for (int ib = 0; ib < LW_NBND; ++ib) {
        LwBand *b = &t->b[ib];
        free(b->absa); free(b->absb); free(b->selfref); free(b->forref); free(b->fracrefa); free(b->fracrefb);
        for (int m = 0; m < 3; ++m) free(b->kam[m]);
        for (int m = 0; m < 2; ++m) { free(b->kbm[m]); free(b->gv[m]); }
    }
    free(t);
}

const double *lw_oracle_table(const LwTables *t, int band, const char *name, int *n)
{
    if (!strcmp(name, "exp_tbl")) { *n = NTBL + 1; return t->exp_tbl; }
    if (!strcmp(name, "tau_tbl")) { *n = NTBL + 1; return t->tau_tbl; }
    if (!strcmp(name, "tfn_tbl")) { *n = NTBL + 1; return t->tfn_tbl; }
    const LwBand *b = &t->b[band - 1];
#define TB(nm, ptr, cnt) if (!strcmp(name, nm)) { *n = (ptr) ? (cnt) : 0; return (ptr); }
    TB("absa", b->absa, b->ng * b->nrowa)
    TB("absb", b->absb, b->ng * b->nrowb)
    TB("selfref", b->selfref, b->ng * 10)
    TB("forref", b->forref, b->ng * 4)
    TB("fracrefa", b->fracrefa, b->ng * b->nfa)
    TB("fracrefb", b->fracrefb, b->ng * b->nfb)
    TB("kam0", b->kam[0], b->ng * b->kam_n[0] * 19)
    TB("kam1", b->kam[1], b->ng * b->kam_n[1] * 19)
    TB("kam2", b->kam[2], b->ng * b->kam_n[2] * 19)
    TB("kbm0", b->kbm[0], b->ng * b->kbm_n[0] * 19)
    TB("kbm1", b->kbm[1], b->ng * b->kbm_n[1] * 19)
    TB("gv0", b->gv[0], b->ng)
    TB("gv1", b->gv[1], b->ng)
#undef TB
    *n = 0;
    return NULL;
}

/* ------------------------------------------------------------------------ */
typedef struct {
    int nlayers, laytrop;
    int jp[MAXLAY], jt[MAXLAY], jt1[MAXLAY], indself[MAXLAY], indfor[MAXLAY], indminor[MAXLAY];
    double pavel[MAXLAY], tavel[MAXLAY], pz[MAXLAY], tz[MAXLAY], coldry[MAXLAY], wbrodl[MAXLAY];
    double wkl[8][MAXLAY], wx[5][MAXLAY];
    double tbound, pwvcm, semiss[LW_NBND];
    double colh2o[MAXLAY], colco2[MAXLAY], colo3[MAXLAY], coln2o[MAXLAY], colco[MAXLAY], colch4[MAXLAY],
        colo2[MAXLAY], colbrd[MAXLAY];
    double fac00[MAXLAY], fac01[MAXLAY], fac10[MAXLAY], fac11[MAXLAY];
    double selffac[MAXLAY], selffrac[MAXLAY], forfac[MAXLAY], forfrac[MAXLAY];
    double minorfrac[MAXLAY], scaleminor[MAXLAY], scaleminorn2[MAXLAY];
    double rat_h2oco2[MAXLAY], rat_h2oco2_1[MAXLAY], rat_h2oo3[MAXLAY], rat_h2oo3_1[MAXLAY], rat_h2on2o[MAXLAY],
        rat_h2on2o_1[MAXLAY], rat_h2och4[MAXLAY], rat_h2och4_1[MAXLAY], rat_n2oco2[MAXLAY], rat_n2oco2_1[MAXLAY],
        rat_o3co2[MAXLAY], rat_o3co2_1[MAXLAY];
    double planklay[LW_NBND][MAXLAY], planklev[LW_NBND][MAXLAY], plankbnd[LW_NBND];
    double taug[LW_NGPT][MAXLAY], fracs[LW_NGPT][MAXLAY];
} LwCol;

#define CHI(m, l) LWREF_CHI_MLS[(m)-1][(l)-1]
#define TOTPLNK(i, b) LWREF_TOTPLNK[(b)-1][(i)-1]

/* setcoef, LW:3540-3808 (istart = 1) */
static void setcoef_lw(LwCol *c)
{
    const double stpfac = 296. / 1013.;
    int indbound = (int)(c->tbound - 159.);
    if (indbound < 1) indbound = 1; else if (indbound > 180) indbound = 180;
    double tbndfrac = c->tbound - 159. - (double)indbound;
    int indlev0 = (int)(c->tz[0] - 159.);
    if (indlev0 < 1) indlev0 = 1; else if (indlev0 > 180) indlev0 = 180;
    double t0frac = c->tz[0] - 159. - (double)indlev0;
    c->laytrop = 0;
    for (int lay = 1; lay <= c->nlayers; ++lay) {
        int indlay = (int)(c->tavel[lay] - 159.);
        if (indlay < 1) indlay = 1; else if (indlay > 180) indlay = 180;
        double tlayfrac = c->tavel[lay] - 159. - (double)indlay;
        int indlev = (int)(c->tz[lay] - 159.);
        if (indlev < 1) indlev = 1; else if (indlev > 180) indlev = 180;
        double tlevfrac = c->tz[lay] - 159. - (double)indlev;
        for (int iband = 1; iband <= 16; ++iband) {
            double dbdtlev, dbdtlay;
            if (lay == 1) {
                dbdtlev = TOTPLNK(indbound + 1, iband) - TOTPLNK(indbound, iband);
                c->plankbnd[iband - 1] = c->semiss[iband - 1] * (TOTPLNK(indbound, iband) + tbndfrac * dbdtlev);
                dbdtlev = TOTPLNK(indlev0 + 1, iband) - TOTPLNK(indlev0, iband);
                c->planklev[iband - 1][0] = TOTPLNK(indlev0, iband) + t0frac * dbdtlev;
            }
            dbdtlev = TOTPLNK(indlev + 1, iband) - TOTPLNK(indlev, iband);
            dbdtlay = TOTPLNK(indlay + 1, iband) - TOTPLNK(indlay, iband);
            c->planklay[iband - 1][lay] = TOTPLNK(indlay, iband) + tlayfrac * dbdtlay;
            c->planklev[iband - 1][lay] = TOTPLNK(indlev, iband) + tlevfrac * dbdtlev;
        }
        double plog = log(c->pavel[lay]);
        c->jp[lay] = (int)(36. - 5 * (plog + 0.04));
        if (c->jp[lay] < 1) c->jp[lay] = 1; else if (c->jp[lay] > 58) c->jp[lay] = 58;
        int jp1 = c->jp[lay] + 1;
        double fp = 5. * (LWREF_PREFLOG[c->jp[lay] - 1] - plog);
        c->jt[lay] = (int)(3. + (c->tavel[lay] - LWREF_TREF[c->jp[lay] - 1]) / 15.);
        if (c->jt[lay] < 1) c->jt[lay] = 1; else if (c->jt[lay] > 4) c->jt[lay] = 4;
        double ft = ((c->tavel[lay] - LWREF_TREF[c->jp[lay] - 1]) / 15.) - (double)(c->jt[lay] - 3);
        c->jt1[lay] = (int)(3. + (c->tavel[lay] - LWREF_TREF[jp1 - 1]) / 15.);
        if (c->jt1[lay] < 1) c->jt1[lay] = 1; else if (c->jt1[lay] > 4) c->jt1[lay] = 4;
        double ft1 = ((c->tavel[lay] - LWREF_TREF[jp1 - 1]) / 15.) - (double)(c->jt1[lay] - 3);
        double water = c->wkl[1][lay] / c->coldry[lay];
        double scalefac = c->pavel[lay] * stpfac / c->tavel[lay];
        double factor;
        int jpl = c->jp[lay];
        if (plog > 4.56) {
            c->laytrop = c->laytrop + 1;
            c->forfac[lay] = scalefac / (1. + water);
            factor = (332.0 - c->tavel[lay]) / 36.0;
            int ifac = (int)factor;
            c->indfor[lay] = ifac > 1 ? (ifac < 2 ? ifac : 2) : 1;
            c->forfrac[lay] = factor - (double)c->indfor[lay];
            c->selffac[lay] = water * c->forfac[lay];
            factor = (c->tavel[lay] - 188.0) / 7.2;
            int is = (int)factor - 7;
            c->indself[lay] = is > 1 ? (is < 9 ? is : 9) : 1;
            c->selffrac[lay] = factor - (double)(c->indself[lay] + 7);
            c->scaleminor[lay] = c->pavel[lay] / c->tavel[lay];
            c->scaleminorn2[lay] = (c->pavel[lay] / c->tavel[lay]) * (c->wbrodl[lay] / (c->coldry[lay] + c->wkl[1][lay]));
            factor = (c->tavel[lay] - 180.8) / 7.2;
            int im = (int)factor;
            c->indminor[lay] = im > 1 ? (im < 18 ? im : 18) : 1;
            c->minorfrac[lay] = factor - (double)c->indminor[lay];
            c->rat_h2oco2[lay] = CHI(1, jpl) / CHI(2, jpl);
            c->rat_h2oco2_1[lay] = CHI(1, jpl + 1) / CHI(2, jpl + 1);
            c->rat_h2oo3[lay] = CHI(1, jpl) / CHI(3, jpl);
            c->rat_h2oo3_1[lay] = CHI(1, jpl + 1) / CHI(3, jpl + 1);
            c->rat_h2on2o[lay] = CHI(1, jpl) / CHI(4, jpl);
            c->rat_h2on2o_1[lay] = CHI(1, jpl + 1) / CHI(4, jpl + 1);
            c->rat_h2och4[lay] = CHI(1, jpl) / CHI(6, jpl);
            c->rat_h2och4_1[lay] = CHI(1, jpl + 1) / CHI(6, jpl + 1);
            c->rat_n2oco2[lay] = CHI(4, jpl) / CHI(2, jpl);
            c->rat_n2oco2_1[lay] = CHI(4, jpl + 1) / CHI(2, jpl + 1);
        } else {
            c->forfac[lay] = scalefac / (1. + water);
            factor = (c->tavel[lay] - 188.0) / 36.0;
            c->indfor[lay] = 3;
            c->forfrac[lay] = factor - 1.0;
            c->selffac[lay] = water * c->forfac[lay];
            c->scaleminor[lay] = c->pavel[lay] / c->tavel[lay];
            c->scaleminorn2[lay] = (c->pavel[lay] / c->tavel[lay]) * (c->wbrodl[lay] / (c->coldry[lay] + c->wkl[1][lay]));
            factor = (c->tavel[lay] - 180.8) / 7.2;
            int im = (int)factor;
            c->indminor[lay] = im > 1 ? (im < 18 ? im : 18) : 1;
            c->minorfrac[lay] = factor - (double)c->indminor[lay];
            c->rat_h2oco2[lay] = CHI(1, jpl) / CHI(2, jpl);
            c->rat_h2oco2_1[lay] = CHI(1, jpl + 1) / CHI(2, jpl + 1);
            c->rat_o3co2[lay] = CHI(3, jpl) / CHI(2, jpl);
            c->rat_o3co2_1[lay] = CHI(3, jpl + 1) / CHI(2, jpl + 1);
        }
        c->colh2o[lay] = 1.e-20 * c->wkl[1][lay];
        c->colco2[lay] = 1.e-20 * c->wkl[2][lay];
        c->colo3[lay] = 1.e-20 * c->wkl[3][lay];
        c->coln2o[lay] = 1.e-20 * c->wkl[4][lay];
        c->colco[lay] = 1.e-20 * c->wkl[5][lay];
        c->colch4[lay] = 1.e-20 * c->wkl[6][lay];
        c->colo2[lay] = 1.e-20 * c->wkl[7][lay];
        if (c->colco2[lay] == 0.) c->colco2[lay] = 1.e-32 * c->coldry[lay];
        if (c->colo3[lay] == 0.) c->colo3[lay] = 1.e-32 * c->coldry[lay];
        if (c->coln2o[lay] == 0.) c->coln2o[lay] = 1.e-32 * c->coldry[lay];
        if (c->colco[lay] == 0.) c->colco[lay] = 1.e-32 * c->coldry[lay];
        if (c->colch4[lay] == 0.) c->colch4[lay] = 1.e-32 * c->coldry[lay];
        c->colbrd[lay] = 1.e-20 * c->wbrodl[lay];
        double compfp = 1. - fp;
        c->fac10[lay] = compfp * ft;
        c->fac00[lay] = compfp * (1. - ft);
        c->fac11[lay] = fp * ft1;
        c->fac01[lay] = fp * (1. - ft1);
        c->selffac[lay] = c->colh2o[lay] * c->selffac[lay];
        c->forfac[lay] = c->colh2o[lay] * c->forfac[lay];
    }
}

/* ------------------------------------------------------------------------ */
/* taumol helpers                                                           */
/* ------------------------------------------------------------------------ */
#define ABSA(ind, ig) b->absa[(size_t)((ig)-1) * b->nrowa + ((ind)-1)]
#define ABSB(ind, ig) b->absb[(size_t)((ig)-1) * b->nrowb + ((ind)-1)]
#define SELF(i, ig) b->selfref[((ig)-1) * 10 + ((i)-1)]
#define FORR(i, ig) b->forref[((ig)-1) * 4 + ((i)-1)]
#define FRA(ig, j) b->fracrefa[((j)-1) * b->ng + ((ig)-1)]
#define FRB(ig, j) b->fracrefb[((j)-1) * b->ng + ((ig)-1)]
/* binary minor table ka_m(js, indm, ig) with dims (n,19,ng); single ka_m(indm, ig) is n = 1 */
#define KM(tab, n, js, im, ig) (tab)[(size_t)((ig)-1) * (n) * 19 + (size_t)((im)-1) * (n) + ((js)-1)]

typedef struct {
    double speccomb, specparm, fs;
    int js, mode;            /* mode 0: 4-point, 1: specparm < 0.125, 2: specparm > 0.875 */
    double f0a, f1a, f2a;    /* x fac(T) */
    double f0b, f1b, f2b;    /* x fac(T+1) */
} Maj;

/* species interpolation for one pressure level of a binary band, e.g. LW:5205-5290 */
static Maj major_setup(double cola, double ratio, double colb, double mult, double facT, double facT1,
                       double oneminus, int edges)
{
    Maj m;
    m.speccomb = cola + ratio * colb;
    m.specparm = cola / m.speccomb;
    if (m.specparm >= oneminus) m.specparm = oneminus;
    double specmult = mult * (m.specparm);
    m.js = 1 + (int)specmult;
    m.fs = fmod(specmult, 1.0);
    m.f2a = m.f2b = 0.;
    if (edges && m.specparm < 0.125) {
        double p = m.fs - 1;
        double p4 = (p * p) * (p * p);
        double fk0 = p4, fk1 = 1 - p - 2.0 * p4, fk2 = p + p4;
        m.mode = 1;
        m.f0a = fk0 * facT; m.f1a = fk1 * facT; m.f2a = fk2 * facT;
        m.f0b = fk0 * facT1; m.f1b = fk1 * facT1; m.f2b = fk2 * facT1;
    } else if (edges && m.specparm > 0.875) {
        double p = -m.fs;
        double p4 = (p * p) * (p * p);
        double fk0 = p4, fk1 = 1 - p - 2.0 * p4, fk2 = p + p4;
        m.mode = 2;
        m.f0a = fk0 * facT; m.f1a = fk1 * facT; m.f2a = fk2 * facT;
        m.f0b = fk0 * facT1; m.f1b = fk1 * facT1; m.f2b = fk2 * facT1;
    } else {
        m.mode = 0;
        m.f0a = (1. - m.fs) * facT; m.f0b = (1. - m.fs) * facT1;
        m.f1a = m.fs * facT; m.f1b = m.fs * facT1;
    }
    return m;
}
/* tau_major for the lower atmosphere (nspa = 9), LW:5303-5325 */
static double major_a(const LwBand *b, const Maj *m, int ind, int ig)
{
    if (m->mode == 1)
        return m->speccomb * (m->f0a * ABSA(ind, ig) + m->f1a * ABSA(ind + 1, ig) + m->f2a * ABSA(ind + 2, ig) +
                              m->f0b * ABSA(ind + 9, ig) + m->f1b * ABSA(ind + 10, ig) + m->f2b * ABSA(ind + 11, ig));
    if (m->mode == 2)
        return m->speccomb * (m->f2a * ABSA(ind - 1, ig) + m->f1a * ABSA(ind, ig) + m->f0a * ABSA(ind + 1, ig) +
                              m->f2b * ABSA(ind + 8, ig) + m->f1b * ABSA(ind + 9, ig) + m->f0b * ABSA(ind + 10, ig));
    return m->speccomb * (m->f0a * ABSA(ind, ig) + m->f1a * ABSA(ind + 1, ig) + m->f0b * ABSA(ind + 9, ig) +
                          m->f1b * ABSA(ind + 10, ig));
}
/* upper atmosphere binary (nspb = 5), LW:5424-5433 */
static double major_b(const LwBand *b, const Maj *m, int ind, int ig)
{
    return m->speccomb * (m->f0a * ABSB(ind, ig) + m->f1a * ABSB(ind + 1, ig) + m->f0b * ABSB(ind + 5, ig) +
                          m->f1b * ABSB(ind + 6, ig));
}
/* species index/fraction for minor species or Planck fractions, e.g. LW:5217-5222 */
static void spec_idx(double cola, double refrat, double colb, double mult, double oneminus, int *j, double *f)
{
    double speccomb = cola + refrat * colb;
    double specparm = cola / speccomb;
    if (specparm >= oneminus) specparm = oneminus;
    double specmult = mult * specparm;
    *j = 1 + (int)specmult;
    *f = fmod(specmult, 1.0);
}
/* minor absorber interpolated in species and temperature, e.g. LW:5297-5301 */
static double minor2(const double *tab, int n, int jm, double fm, int indm, double minorfrac, int ig)
{
    double m1 = KM(tab, n, jm, indm, ig) + fm * (KM(tab, n, jm + 1, indm, ig) - KM(tab, n, jm, indm, ig));
    double m2 = KM(tab, n, jm, indm + 1, ig) + fm * (KM(tab, n, jm + 1, indm + 1, ig) - KM(tab, n, jm, indm + 1, ig));
    return m1 + minorfrac * (m2 - m1);
}
static double minor1(const double *tab, int indm, double minorfrac, int ig)
{
    return (KM(tab, 1, 1, indm, ig) + minorfrac * (KM(tab, 1, 1, indm + 1, ig) - KM(tab, 1, 1, indm, ig)));
}
/* "too abundant to be minor" adjustment, e.g. LW:5209-5216 */
static double adjcol(double col, double coldry, double chiref, double thr, double a0, double expo)
{
    double chi = col / (coldry);
    double rat = 1.e20 * chi / chiref;
    if (rat > thr) {
        double adjfac = a0 + pow(rat - a0, expo);
        return adjfac * chiref * coldry * 1.e-20;
    }
    return col;
}

#define S4A(ind0, ind1, ig)                                                                          \
    (c->fac00[lay] * ABSA(ind0, ig) + c->fac10[lay] * ABSA((ind0) + 1, ig) + c->fac01[lay] * ABSA(ind1, ig) + \
     c->fac11[lay] * ABSA((ind1) + 1, ig))
#define S4B(ind0, ind1, ig)                                                                          \
    (c->fac00[lay] * ABSB(ind0, ig) + c->fac10[lay] * ABSB((ind0) + 1, ig) + c->fac01[lay] * ABSB(ind1, ig) + \
     c->fac11[lay] * ABSB((ind1) + 1, ig))
#define TAUSELF(ig) (c->selffac[lay] * (SELF(inds, ig) + c->selffrac[lay] * (SELF(inds + 1, ig) - SELF(inds, ig))))
#define TAUFOR(ig) (c->forfac[lay] * (FORR(indf, ig) + c->forfrac[lay] * (FORR(indf + 1, ig) - FORR(indf, ig))))

/* taumol: taugb1..taugb16, LW:4712-7828 */
static void taumol_lw(const LwTables *t, LwCol *c)
{
    const double om = t->oneminus;
    const int nl = c->nlayers, laytrop = c->laytrop;
    for (int ib = 0; ib < LW_NBND; ++ib) {
        const LwBand *b = &t->b[ib];
        const int band = ib + 1, ng = b->ng, g0 = ib == 0 ? 0 : ngs[ib - 1];
        for (int lay = 1; lay <= nl; ++lay) {
            const int lower = lay <= laytrop;
            const int jp = c->jp[lay];
            int ind0, ind1;
            const int inds = c->indself[lay], indf = c->indfor[lay], indm = c->indminor[lay];
            const double mf = c->minorfrac[lay];
            double *tg = NULL, *fr = NULL;
            /* helpers to write outputs */
#define TG(ig) c->taug[g0 + (ig)-1][lay]
#define FR(ig) c->fracs[g0 + (ig)-1][lay]
            (void)tg; (void)fr;
            if (lower) {
                ind0 = ((jp - 1) * 5 + (c->jt[lay] - 1)) * nspa[ib];
                ind1 = (jp * 5 + (c->jt1[lay] - 1)) * nspa[ib];
            } else {
                ind0 = ((jp - 13) * 5 + (c->jt[lay] - 1)) * nspb[ib];
                ind1 = ((jp - 12) * 5 + (c->jt1[lay] - 1)) * nspb[ib];
            }
            switch (band) {
            case 1: {   /* LW:4860-4935: h2o; minor n2 */
                double pp = c->pavel[lay], corradj;
                double scalen2 = c->colbrd[lay] * c->scaleminorn2[lay];
                if (lower) {
                    corradj = 1.;
                    if (pp < 250.) corradj = 1. - 0.15 * (250. - pp) / 154.4;
                    for (int ig = 1; ig <= ng; ++ig) {
                        double tauself = TAUSELF(ig), taufor = TAUFOR(ig);
                        double taun2 = scalen2 * (KM(b->kam[0], 1, 1, indm, ig) +
                                                  mf * (KM(b->kam[0], 1, 1, indm + 1, ig) - KM(b->kam[0], 1, 1, indm, ig)));
                        TG(ig) = corradj * (c->colh2o[lay] * S4A(ind0 + 1, ind1 + 1, ig) + tauself + taufor + taun2);
                        FR(ig) = FRA(ig, 1);
                    }
                } else {
                    corradj = 1. - 0.15 * (pp / 95.6);
                    for (int ig = 1; ig <= ng; ++ig) {
                        double taufor = TAUFOR(ig);
                        double taun2 = scalen2 * (KM(b->kbm[0], 1, 1, indm, ig) +
                                                  mf * (KM(b->kbm[0], 1, 1, indm + 1, ig) - KM(b->kbm[0], 1, 1, indm, ig)));
                        TG(ig) = corradj * (c->colh2o[lay] * S4B(ind0 + 1, ind1 + 1, ig) + taufor + taun2);
                        FR(ig) = FRB(ig, 1);
                    }
                }
                break;
            }
            case 2: {   /* LW:5010-5070 */
                if (lower) {
                    double pp = c->pavel[lay];
                    double corradj = 1. - .05 * (pp - 100.) / 900.;
                    for (int ig = 1; ig <= ng; ++ig) {
                        double tauself = TAUSELF(ig), taufor = TAUFOR(ig);
                        TG(ig) = corradj * (c->colh2o[lay] * S4A(ind0 + 1, ind1 + 1, ig) + tauself + taufor);
                        FR(ig) = FRA(ig, 1);
                    }
                } else {
                    for (int ig = 1; ig <= ng; ++ig) {
                        double taufor = TAUFOR(ig);
                        TG(ig) = c->colh2o[lay] * S4B(ind0 + 1, ind1 + 1, ig) + taufor;
                        FR(ig) = FRB(ig, 1);
                    }
                }
                break;
            }
            case 3: case 4: case 5: case 7: case 9: case 12: case 13: case 15: case 16:
                if (lower) {
                    /* binary key species in the lower atmosphere */
                    double cola, colb, r0, r1, rpl, rm = 0., rm3 = 0.;
                    switch (band) {
                    case 3: cola = c->colh2o[lay]; colb = c->colco2[lay]; r0 = c->rat_h2oco2[lay]; r1 = c->rat_h2oco2_1[lay];
                        rpl = CHI(1, 9) / CHI(2, 9); rm = CHI(1, 3) / CHI(2, 3); break;
                    case 4: cola = c->colh2o[lay]; colb = c->colco2[lay]; r0 = c->rat_h2oco2[lay]; r1 = c->rat_h2oco2_1[lay];
                        rpl = CHI(1, 11) / CHI(2, 11); break;
                    case 5: cola = c->colh2o[lay]; colb = c->colco2[lay]; r0 = c->rat_h2oco2[lay]; r1 = c->rat_h2oco2_1[lay];
                        rpl = CHI(1, 5) / CHI(2, 5); rm = CHI(1, 7) / CHI(2, 7); break;
                    case 7: cola = c->colh2o[lay]; colb = c->colo3[lay]; r0 = c->rat_h2oo3[lay]; r1 = c->rat_h2oo3_1[lay];
                        rpl = CHI(1, 3) / CHI(3, 3); rm = CHI(1, 3) / CHI(3, 3); break;
                    case 9: cola = c->colh2o[lay]; colb = c->colch4[lay]; r0 = c->rat_h2och4[lay]; r1 = c->rat_h2och4_1[lay];
                        rpl = CHI(1, 9) / CHI(6, 9); rm = CHI(1, 3) / CHI(6, 3); break;
                    case 12: cola = c->colh2o[lay]; colb = c->colco2[lay]; r0 = c->rat_h2oco2[lay]; r1 = c->rat_h2oco2_1[lay];
                        rpl = CHI(1, 10) / CHI(2, 10); break;
                    case 13: cola = c->colh2o[lay]; colb = c->coln2o[lay]; r0 = c->rat_h2on2o[lay]; r1 = c->rat_h2on2o_1[lay];
                        rpl = CHI(1, 5) / CHI(4, 5); rm = CHI(1, 1) / CHI(4, 1); rm3 = CHI(1, 3) / CHI(4, 3); break;
                    case 15: cola = c->coln2o[lay]; colb = c->colco2[lay]; r0 = c->rat_n2oco2[lay]; r1 = c->rat_n2oco2_1[lay];
                        rpl = CHI(4, 1) / CHI(2, 1); rm = CHI(4, 1) / CHI(2, 1); break;
                    default: cola = c->colh2o[lay]; colb = c->colch4[lay]; r0 = c->rat_h2och4[lay]; r1 = c->rat_h2och4_1[lay];
                        rpl = CHI(1, 6) / CHI(6, 6); break;
                    }
                    Maj m0 = major_setup(cola, r0, colb, 8., c->fac00[lay], c->fac10[lay], om, 1);
                    Maj m1 = major_setup(cola, r1, colb, 8., c->fac01[lay], c->fac11[lay], om, 1);
                    int jm = 1, jm3 = 1, jpl;
                    double fm = 0., fm3 = 0., fpl;
                    if (rm != 0.) spec_idx(cola, rm, colb, 8., om, &jm, &fm);
                    if (rm3 != 0.) spec_idx(cola, rm3, colb, 8., om, &jm3, &fm3);
                    double adj = 0.;
                    if (band == 3 || band == 9) adj = adjcol(c->coln2o[lay], c->coldry[lay], CHI(4, jp + 1), 1.5, 0.5, 0.65);
                    if (band == 7) adj = adjcol(c->colco2[lay], c->coldry[lay], CHI(2, jp + 1), 3.0, 3.0, 0.79);
                    if (band == 13) adj = adjcol(c->colco2[lay], c->coldry[lay], 3.55e-4, 3.0, 2.0, 0.68);
                    spec_idx(cola, rpl, colb, 8., om, &jpl, &fpl);
                    double scalen2 = c->colbrd[lay] * c->scaleminor[lay];
                    int i0 = ind0 + m0.js, i1 = ind1 + m1.js;
                    for (int ig = 1; ig <= ng; ++ig) {
                        double tauself = TAUSELF(ig), taufor = TAUFOR(ig);
                        double tau_major = major_a(b, &m0, i0, ig), tau_major1 = major_a(b, &m1, i1, ig);
                        double v;
                        switch (band) {
                        case 3: case 9: {
                            double absn2o = minor2(b->kam[0], 9, jm, fm, indm, mf, ig);
                            v = tau_major + tau_major1 + tauself + taufor + adj * absn2o;
                            break;
                        }
                        case 5: {
                            double abso3 = minor2(b->kam[0], 9, jm, fm, indm, mf, ig);
                            v = tau_major + tau_major1 + tauself + taufor + abso3 * c->colo3[lay] +
                                c->wx[1][lay] * b->gv[0][ig - 1];
                            break;
                        }
                        case 7: {
                            double absco2 = minor2(b->kam[0], 9, jm, fm, indm, mf, ig);
                            v = tau_major + tau_major1 + tauself + taufor + adj * absco2;
                            break;
                        }
                        case 13: {
                            double absco2 = minor2(b->kam[0], 9, jm, fm, indm, mf, ig);
                            double absco = minor2(b->kam[1], 9, jm3, fm3, indm, mf, ig);
                            v = tau_major + tau_major1 + tauself + taufor + adj * absco2 + c->colco[lay] * absco;
                            break;
                        }
                        case 15: {
                            double n2m1 = KM(b->kam[0], 9, jm, indm, ig) +
                                          fm * (KM(b->kam[0], 9, jm + 1, indm, ig) - KM(b->kam[0], 9, jm, indm, ig));
                            double n2m2 = KM(b->kam[0], 9, jm, indm + 1, ig) +
                                          fm * (KM(b->kam[0], 9, jm + 1, indm + 1, ig) - KM(b->kam[0], 9, jm, indm + 1, ig));
                            double taun2 = scalen2 * (n2m1 + mf * (n2m2 - n2m1));
                            v = tau_major + tau_major1 + tauself + taufor + taun2;
                            break;
                        }
                        default: v = tau_major + tau_major1 + tauself + taufor; break;
                        }
                        TG(ig) = v;
                        FR(ig) = FRA(ig, jpl) + fpl * (FRA(ig, jpl + 1) - FRA(ig, jpl));
                    }
                } else {
                    switch (band) {
                    case 3: case 4: case 5: {   /* binary upper atmosphere */
                        double cola, colb, r0, r1, rpl;
                        if (band == 3) { cola = c->colh2o[lay]; colb = c->colco2[lay]; r0 = c->rat_h2oco2[lay];
                            r1 = c->rat_h2oco2_1[lay]; rpl = CHI(1, 13) / CHI(2, 13); }
                        else { cola = c->colo3[lay]; colb = c->colco2[lay]; r0 = c->rat_o3co2[lay]; r1 = c->rat_o3co2_1[lay];
                            rpl = band == 4 ? CHI(3, 13) / CHI(2, 13) : CHI(3, 43) / CHI(2, 43); }
                        Maj m0 = major_setup(cola, r0, colb, 4., c->fac00[lay], c->fac10[lay], om, 0);
                        Maj m1 = major_setup(cola, r1, colb, 4., c->fac01[lay], c->fac11[lay], om, 0);
                        int jm = 1, jpl;
                        double fm = 0., fpl, adj = 0.;
                        if (band == 3) {
                            spec_idx(cola, CHI(1, 13) / CHI(2, 13), colb, 4., om, &jm, &fm);
                            adj = adjcol(c->coln2o[lay], c->coldry[lay], CHI(4, jp + 1), 1.5, 0.5, 0.65);
                        }
                        spec_idx(cola, rpl, colb, 4., om, &jpl, &fpl);
                        int i0 = ind0 + m0.js, i1 = ind1 + m1.js;
                        for (int ig = 1; ig <= ng; ++ig) {
                            double v = major_b(b, &m0, i0, ig) + major_b(b, &m1, i1, ig);
                            if (band == 3) {
                                double taufor = TAUFOR(ig);
                                double absn2o = minor2(b->kbm[0], 5, jm, fm, indm, mf, ig);
                                v = v + taufor + adj * absn2o;
                            } else if (band == 5) {
                                v = v + c->wx[1][lay] * b->gv[0][ig - 1];
                            }
                            TG(ig) = v;
                            FR(ig) = FRB(ig, jpl) + fpl * (FRB(ig, jpl + 1) - FRB(ig, jpl));
                        }
                        if (band == 4) {   /* empirical scaling, LW:5687-5696 */
                            TG(8) = TG(8) * 0.92; TG(9) = TG(9) * 0.88; TG(10) = TG(10) * 1.07; TG(11) = TG(11) * 1.1;
                            TG(12) = TG(12) * 0.99; TG(13) = TG(13) * 0.88; TG(14) = TG(14) * 0.943;
                        }
                        break;
                    }
                    case 7: {   /* o3 + minor co2, LW:6290-6331 */
                        double adj = adjcol(c->colco2[lay], c->coldry[lay], CHI(2, jp + 1), 3.0, 2.0, 0.79);
                        for (int ig = 1; ig <= ng; ++ig) {
                            double absco2 = KM(b->kbm[0], 1, 1, indm, ig) +
                                            mf * (KM(b->kbm[0], 1, 1, indm + 1, ig) - KM(b->kbm[0], 1, 1, indm, ig));
                            TG(ig) = c->colo3[lay] * S4B(ind0 + 1, ind1 + 1, ig) + adj * absco2;
                            FR(ig) = FRB(ig, 1);
                        }
                        TG(6) = TG(6) * 0.92; TG(7) = TG(7) * 0.88; TG(8) = TG(8) * 1.07; TG(9) = TG(9) * 1.1;
                        TG(10) = TG(10) * 0.99; TG(11) = TG(11) * 0.855;
                        break;
                    }
                    case 9: {
                        double adj = adjcol(c->coln2o[lay], c->coldry[lay], CHI(4, jp + 1), 1.5, 0.5, 0.65);
                        for (int ig = 1; ig <= ng; ++ig) {
                            double absn2o = KM(b->kbm[0], 1, 1, indm, ig) +
                                            mf * (KM(b->kbm[0], 1, 1, indm + 1, ig) - KM(b->kbm[0], 1, 1, indm, ig));
                            TG(ig) = c->colch4[lay] * S4B(ind0 + 1, ind1 + 1, ig) + adj * absn2o;
                            FR(ig) = FRB(ig, 1);
                        }
                        break;
                    }
                    case 13:
                        for (int ig = 1; ig <= ng; ++ig) {
                            double abso3 = KM(b->kbm[0], 1, 1, indm, ig) +
                                           mf * (KM(b->kbm[0], 1, 1, indm + 1, ig) - KM(b->kbm[0], 1, 1, indm, ig));
                            TG(ig) = c->colo3[lay] * abso3;
                            FR(ig) = FRB(ig, 1);
                        }
                        break;
                    case 16:
                        for (int ig = 1; ig <= ng; ++ig) {
                            TG(ig) = c->colch4[lay] * S4B(ind0 + 1, ind1 + 1, ig);
                            FR(ig) = FRB(ig, 1);
                        }
                        break;
                    default:   /* 12, 15: nothing above the tropopause */
                        for (int ig = 1; ig <= ng; ++ig) { TG(ig) = 0.0; FR(ig) = 0.0; }
                        break;
                    }
                }
                break;
            case 6: {   /* LW:5880-5935 */
                if (lower) {
                    double adj = adjcol(c->colco2[lay], c->coldry[lay], CHI(2, jp + 1), 3.0, 2.0, 0.77);
                    for (int ig = 1; ig <= ng; ++ig) {
                        double tauself = TAUSELF(ig), taufor = TAUFOR(ig);
                        double absco2 = minor1(b->kam[0], indm, mf, ig);
                        TG(ig) = c->colh2o[lay] * S4A(ind0 + 1, ind1 + 1, ig) + tauself + taufor + adj * absco2 +
                                 c->wx[2][lay] * b->gv[0][ig - 1] + c->wx[3][lay] * b->gv[1][ig - 1];
                        FR(ig) = FRA(ig, 1);
                    }
                } else {
                    for (int ig = 1; ig <= ng; ++ig) {
                        TG(ig) = 0.0 + c->wx[2][lay] * b->gv[0][ig - 1] + c->wx[3][lay] * b->gv[1][ig - 1];
                        FR(ig) = FRA(ig, 1);
                    }
                }
                break;
            }
            case 8: {   /* LW:6390-6470 */
                double adj = adjcol(c->colco2[lay], c->coldry[lay], CHI(2, jp + 1), 3.0, 2.0, 0.65);
                if (lower) {
                    for (int ig = 1; ig <= ng; ++ig) {
                        double tauself = TAUSELF(ig), taufor = TAUFOR(ig);
                        double absco2 = minor1(b->kam[0], indm, mf, ig);
                        double abso3 = minor1(b->kam[2], indm, mf, ig);
                        double absn2o = minor1(b->kam[1], indm, mf, ig);
                        TG(ig) = c->colh2o[lay] * S4A(ind0 + 1, ind1 + 1, ig) + tauself + taufor + adj * absco2 +
                                 c->colo3[lay] * abso3 + c->coln2o[lay] * absn2o + c->wx[3][lay] * b->gv[0][ig - 1] +
                                 c->wx[4][lay] * b->gv[1][ig - 1];
                        FR(ig) = FRA(ig, 1);
                    }
                } else {
                    for (int ig = 1; ig <= ng; ++ig) {
                        double absco2 = minor1(b->kbm[0], indm, mf, ig);
                        double absn2o = minor1(b->kbm[1], indm, mf, ig);
                        TG(ig) = c->colo3[lay] * S4B(ind0 + 1, ind1 + 1, ig) + adj * absco2 + c->coln2o[lay] * absn2o +
                                 c->wx[3][lay] * b->gv[0][ig - 1] + c->wx[4][lay] * b->gv[1][ig - 1];
                        FR(ig) = FRB(ig, 1);
                    }
                }
                break;
            }
            case 10: case 14: {
                double col = band == 10 ? c->colh2o[lay] : c->colco2[lay];
                if (lower) {
                    for (int ig = 1; ig <= ng; ++ig) {
                        double tauself = TAUSELF(ig), taufor = TAUFOR(ig);
                        TG(ig) = col * S4A(ind0 + 1, ind1 + 1, ig) + tauself + taufor;
                        FR(ig) = FRA(ig, 1);
                    }
                } else {
                    for (int ig = 1; ig <= ng; ++ig) {
                        double v = col * S4B(ind0 + 1, ind1 + 1, ig);
                        if (band == 10) { double taufor = TAUFOR(ig); v = v + taufor; }
                        TG(ig) = v;
                        FR(ig) = FRB(ig, 1);
                    }
                }
                break;
            }
            case 11: {
                double scaleo2 = c->colo2[lay] * c->scaleminor[lay];
                if (lower) {
                    for (int ig = 1; ig <= ng; ++ig) {
                        double tauself = TAUSELF(ig), taufor = TAUFOR(ig);
                        double tauo2 = scaleo2 * (KM(b->kam[0], 1, 1, indm, ig) +
                                                  mf * (KM(b->kam[0], 1, 1, indm + 1, ig) - KM(b->kam[0], 1, 1, indm, ig)));
                        TG(ig) = c->colh2o[lay] * S4A(ind0 + 1, ind1 + 1, ig) + tauself + taufor + tauo2;
                        FR(ig) = FRA(ig, 1);
                    }
                } else {
                    for (int ig = 1; ig <= ng; ++ig) {
                        double taufor = TAUFOR(ig);
                        double tauo2 = scaleo2 * (KM(b->kbm[0], 1, 1, indm, ig) +
                                                  mf * (KM(b->kbm[0], 1, 1, indm + 1, ig) - KM(b->kbm[0], 1, 1, indm, ig)));
                        TG(ig) = c->colh2o[lay] * S4B(ind0 + 1, ind1 + 1, ig) + taufor + tauo2;
                        FR(ig) = FRB(ig, 1);
                    }
                }
                break;
            }
            }
        }
    }
#undef TG
#undef FR
}

/* rtrnmc for a cloud-free column, LW:3135-3410 (icldlyr = 0 everywhere, so the
 * clear stream equals the total stream: iclddn = 0) */
static void rtrnmc_lw(const LwTables *t, const LwCol *c, const double (*taut)[MAXLAY], double *totuflux,
                      double *totdflux, double *htr)
{
    static const double a0[LW_NBND] = {1.66, 1.55, 1.58, 1.66, 1.54, 1.454, 1.89, 1.33,
                                       1.668, 1.66, 1.66, 1.66, 1.66, 1.66, 1.66, 1.66};
    static const double a1[LW_NBND] = {0.00, 0.25, 0.22, 0.00, 0.13, 0.446, -0.10, 0.40,
                                       -0.006, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00};
    static const double a2[LW_NBND] = {0.00, -12.0, -11.7, 0.00, -0.72, -0.243, 0.19, -0.062,
                                       0.414, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00};
    const double wtdiff = 0.5, rec_6 = 0.166667;
    const int nlayers = c->nlayers;
    double secdiff[LW_NBND], urad[MAXLAY], drad[MAXLAY], atrans[MAXLAY], bbugas[MAXLAY], fnet[MAXLAY];
    for (int ibnd = 1; ibnd <= LW_NBND; ++ibnd) {
        if (ibnd == 1 || ibnd == 4 || ibnd >= 10) secdiff[ibnd - 1] = 1.66;
        else {
            secdiff[ibnd - 1] = a0[ibnd - 1] + a1[ibnd - 1] * exp(a2[ibnd - 1] * c->pwvcm);
            if (secdiff[ibnd - 1] > 1.80) secdiff[ibnd - 1] = 1.80;
            if (secdiff[ibnd - 1] < 1.50) secdiff[ibnd - 1] = 1.50;
        }
    }
    for (int lay = 0; lay <= nlayers; ++lay) { urad[lay] = 0.0; drad[lay] = 0.0; totuflux[lay] = 0.0; totdflux[lay] = 0.0; }
    int igc = 1;
    for (int iband = 1; iband <= LW_NBND; ++iband) {
        for (; igc <= ngs[iband - 1]; ++igc) {
            double radld = 0.;
            for (int lev = nlayers; lev >= 1; --lev) {
                double plfrac = c->fracs[igc - 1][lev];
                double blay = c->planklay[iband - 1][lev];
                double dplankup = c->planklev[iband - 1][lev] - blay;
                double dplankdn = c->planklev[iband - 1][lev - 1] - blay;
                double odepth = secdiff[iband - 1] * taut[igc - 1][lev];
                if (odepth < 0.0) odepth = 0.0;
                double bbd;
                if (odepth <= 0.06) {
                    atrans[lev] = odepth - 0.5 * odepth * odepth;
                    odepth = rec_6 * odepth;
                    bbd = plfrac * (blay + dplankdn * odepth);
                    bbugas[lev] = plfrac * (blay + dplankup * odepth);
                } else {
                    double tblind = odepth / (t->bpade + odepth);
                    int itr = (int)(10000.0 * tblind + 0.5);
                    double transc = t->exp_tbl[itr];
                    atrans[lev] = 1. - transc;
                    double tausfac = t->tfn_tbl[itr];
                    bbd = plfrac * (blay + tausfac * dplankdn);
                    bbugas[lev] = plfrac * (blay + tausfac * dplankup);
                }
                radld = radld + (bbd - radld) * atrans[lev];
                drad[lev - 1] = drad[lev - 1] + radld;
            }
            double rad0 = c->fracs[igc - 1][1] * c->plankbnd[iband - 1];
            double reflect = 1. - c->semiss[iband - 1];
            double radlu = rad0 + reflect * radld;
            urad[0] = urad[0] + radlu;
            for (int lev = 1; lev <= nlayers; ++lev) {
                radlu = radlu + (bbugas[lev] - radlu) * atrans[lev];
                urad[lev] = urad[lev] + radlu;
            }
        }
        for (int lev = nlayers; lev >= 0; --lev) {
            double uflux = urad[lev] * wtdiff, dflux = drad[lev] * wtdiff;
            urad[lev] = 0.0; drad[lev] = 0.0;
            totuflux[lev] = totuflux[lev] + uflux * delwave[iband - 1];
            totdflux[lev] = totdflux[lev] + dflux * delwave[iband - 1];
        }
    }
    totuflux[0] = totuflux[0] * t->fluxfac;
    totdflux[0] = totdflux[0] * t->fluxfac;
    fnet[0] = totuflux[0] - totdflux[0];
    for (int lev = 1; lev <= nlayers; ++lev) {
        totuflux[lev] = totuflux[lev] * t->fluxfac;
        totdflux[lev] = totdflux[lev] * t->fluxfac;
        fnet[lev] = totuflux[lev] - totdflux[lev];
        int l = lev - 1;
        htr[l] = t->heatfac * (fnet[l] - fnet[lev]) / (c->pz[l] - c->pz[lev]);
    }
    htr[nlayers] = 0.0;
}

/* rtrnmc with McICA clouds, LW:3135-3410 in full: cldfmc / taucmc are [LW_NGPT][nlayers] (0-based layers).  A layer
 * in which ANY sub-column is cloudy (icldlyr) takes the three-regime "cloudy" arithmetic for every g-point, and the
 * clear-sky stream is carried separately from the first cloudy layer downwards (iclddn). */
static void rtrnmc_lw_cloudy(const LwTables *t, const LwCol *c, const double (*taut)[MAXLAY], const double *cldfmc,
                             const double *taucmc, double *totuflux, double *totdflux, double *htr, double *totuclfl,
                             double *totdclfl, double *htrc)
{
    static const double a0[LW_NBND] = {1.66, 1.55, 1.58, 1.66, 1.54, 1.454, 1.89, 1.33,
                                       1.668, 1.66, 1.66, 1.66, 1.66, 1.66, 1.66, 1.66};
    static const double a1[LW_NBND] = {0.00, 0.25, 0.22, 0.00, 0.13, 0.446, -0.10, 0.40,
                                       -0.006, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00};
    static const double a2[LW_NBND] = {0.00, -12.0, -11.7, 0.00, -0.72, -0.243, 0.19, -0.062,
                                       0.414, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00};
    const double wtdiff = 0.5, rec_6 = 0.166667;
    const int nlayers = c->nlayers;
    double secdiff[LW_NBND], urad[MAXLAY], drad[MAXLAY], clrurad[MAXLAY], clrdrad[MAXLAY];
    double atrans[MAXLAY], atot[MAXLAY], bbugas[MAXLAY], bbutot[MAXLAY], fnet[MAXLAY], fnetc[MAXLAY];
    int icldlyr[MAXLAY];
    for (int ibnd = 1; ibnd <= LW_NBND; ++ibnd) {
        if (ibnd == 1 || ibnd == 4 || ibnd >= 10) secdiff[ibnd - 1] = 1.66;
        else {
            secdiff[ibnd - 1] = a0[ibnd - 1] + a1[ibnd - 1] * exp(a2[ibnd - 1] * c->pwvcm);
            if (secdiff[ibnd - 1] > 1.80) secdiff[ibnd - 1] = 1.80;
            if (secdiff[ibnd - 1] < 1.50) secdiff[ibnd - 1] = 1.50;
        }
    }
    for (int lay = 0; lay <= nlayers; ++lay) {
        urad[lay] = 0.0; drad[lay] = 0.0; totuflux[lay] = 0.0; totdflux[lay] = 0.0;
        clrurad[lay] = 0.0; clrdrad[lay] = 0.0; totuclfl[lay] = 0.0; totdclfl[lay] = 0.0;
    }
    for (int lay = 1; lay <= nlayers; ++lay) {
        icldlyr[lay] = 0;
        for (int ig = 0; ig < LW_NGPT; ++ig)
            if (cldfmc[(size_t)ig * nlayers + lay - 1] == 1.) icldlyr[lay] = 1;
    }
    int igc = 1;
    for (int iband = 1; iband <= LW_NBND; ++iband) {
        for (; igc <= ngs[iband - 1]; ++igc) {
            const double *cf = cldfmc + (size_t)(igc - 1) * nlayers, *tc = taucmc + (size_t)(igc - 1) * nlayers;
            double radld = 0., radclrd = 0.;
            int iclddn = 0;
            for (int lev = nlayers; lev >= 1; --lev) {
                const double cfm = cf[lev - 1];
                double odcld = 0., efclfrac = 0.;
                if (cfm == 1.) {
                    odcld = secdiff[iband - 1] * tc[lev - 1];
                    double transcld = exp(-odcld);
                    double abscld = 1. - transcld;
                    efclfrac = abscld * cfm;
                }
                double plfrac = c->fracs[igc - 1][lev];
                double blay = c->planklay[iband - 1][lev];
                double dplankup = c->planklev[iband - 1][lev] - blay;
                double dplankdn = c->planklev[iband - 1][lev - 1] - blay;
                double odepth = secdiff[iband - 1] * taut[igc - 1][lev];
                if (odepth < 0.0) odepth = 0.0;
                double bbd;
                if (icldlyr[lev] == 1) {
                    iclddn = 1;
                    double odtot = odepth + odcld;
                    double gassrc, bbdtot;
                    if (odtot < 0.06) {
                        atrans[lev] = odepth - 0.5 * odepth * odepth;
                        double odepth_rec = rec_6 * odepth;
                        gassrc = plfrac * (blay + dplankdn * odepth_rec) * atrans[lev];
                        atot[lev] = odtot - 0.5 * odtot * odtot;
                        double odtot_rec = rec_6 * odtot;
                        bbdtot = plfrac * (blay + dplankdn * odtot_rec);
                        bbd = plfrac * (blay + dplankdn * odepth_rec);
                        bbugas[lev] = plfrac * (blay + dplankup * odepth_rec);
                        bbutot[lev] = plfrac * (blay + dplankup * odtot_rec);
                    } else if (odepth <= 0.06) {
                        atrans[lev] = odepth - 0.5 * odepth * odepth;
                        double odepth_rec = rec_6 * odepth;
                        gassrc = plfrac * (blay + dplankdn * odepth_rec) * atrans[lev];
                        odtot = odepth + odcld;
                        double tblind = odtot / (t->bpade + odtot);
                        int ittot = (int)(10000.0 * tblind + 0.5);
                        double tfactot = t->tfn_tbl[ittot];
                        bbdtot = plfrac * (blay + tfactot * dplankdn);
                        bbd = plfrac * (blay + dplankdn * odepth_rec);
                        atot[lev] = 1. - t->exp_tbl[ittot];
                        bbugas[lev] = plfrac * (blay + dplankup * odepth_rec);
                        bbutot[lev] = plfrac * (blay + tfactot * dplankup);
                    } else {
                        double tblind = odepth / (t->bpade + odepth);
                        int itgas = (int)(10000.0 * tblind + 0.5);
                        odepth = t->tau_tbl[itgas];
                        atrans[lev] = 1. - t->exp_tbl[itgas];
                        double tfacgas = t->tfn_tbl[itgas];
                        gassrc = atrans[lev] * plfrac * (blay + tfacgas * dplankdn);
                        odtot = odepth + odcld;
                        tblind = odtot / (t->bpade + odtot);
                        int ittot = (int)(10000.0 * tblind + 0.5);
                        double tfactot = t->tfn_tbl[ittot];
                        bbdtot = plfrac * (blay + tfactot * dplankdn);
                        bbd = plfrac * (blay + tfacgas * dplankdn);
                        atot[lev] = 1. - t->exp_tbl[ittot];
                        bbugas[lev] = plfrac * (blay + tfacgas * dplankup);
                        bbutot[lev] = plfrac * (blay + tfactot * dplankup);
                    }
                    radld = radld - radld * (atrans[lev] + efclfrac * (1. - atrans[lev])) + gassrc +
                            cfm * (bbdtot * atot[lev] - gassrc);
                    drad[lev - 1] = drad[lev - 1] + radld;
                } else {
                    if (odepth <= 0.06) {
                        atrans[lev] = odepth - 0.5 * odepth * odepth;
                        odepth = rec_6 * odepth;
                        bbd = plfrac * (blay + dplankdn * odepth);
                        bbugas[lev] = plfrac * (blay + dplankup * odepth);
                    } else {
                        double tblind = odepth / (t->bpade + odepth);
                        int itr = (int)(10000.0 * tblind + 0.5);
                        double transc = t->exp_tbl[itr];
                        atrans[lev] = 1. - transc;
                        double tausfac = t->tfn_tbl[itr];
                        bbd = plfrac * (blay + tausfac * dplankdn);
                        bbugas[lev] = plfrac * (blay + tausfac * dplankup);
                    }
                    radld = radld + (bbd - radld) * atrans[lev];
                    drad[lev - 1] = drad[lev - 1] + radld;
                }
                if (iclddn == 1) {
                    radclrd = radclrd + (bbd - radclrd) * atrans[lev];
                    clrdrad[lev - 1] = clrdrad[lev - 1] + radclrd;
                } else {
                    radclrd = radld;
                    clrdrad[lev - 1] = drad[lev - 1];
                }
            }
            double rad0 = c->fracs[igc - 1][1] * c->plankbnd[iband - 1];
            double reflect = 1. - c->semiss[iband - 1];
            double radlu = rad0 + reflect * radld;
            double radclru = rad0 + reflect * radclrd;
            urad[0] = urad[0] + radlu;
            clrurad[0] = clrurad[0] + radclru;
            for (int lev = 1; lev <= nlayers; ++lev) {
                const double cfm = cf[lev - 1];
                if (icldlyr[lev] == 1) {
                    double efclfrac = 0.;
                    if (cfm == 1.) efclfrac = (1. - exp(-(secdiff[iband - 1] * tc[lev - 1]))) * cfm;
                    double gassrc = bbugas[lev] * atrans[lev];
                    radlu = radlu - radlu * (atrans[lev] + efclfrac * (1. - atrans[lev])) + gassrc +
                            cfm * (bbutot[lev] * atot[lev] - gassrc);
                    urad[lev] = urad[lev] + radlu;
                } else {
                    radlu = radlu + (bbugas[lev] - radlu) * atrans[lev];
                    urad[lev] = urad[lev] + radlu;
                }
                if (iclddn == 1) {
                    radclru = radclru + (bbugas[lev] - radclru) * atrans[lev];
                    clrurad[lev] = clrurad[lev] + radclru;
                } else {
                    radclru = radlu;
                    clrurad[lev] = urad[lev];
                }
            }
        }
        for (int lev = nlayers; lev >= 0; --lev) {
            double uflux = urad[lev] * wtdiff, dflux = drad[lev] * wtdiff;
            urad[lev] = 0.0; drad[lev] = 0.0;
            totuflux[lev] = totuflux[lev] + uflux * delwave[iband - 1];
            totdflux[lev] = totdflux[lev] + dflux * delwave[iband - 1];
            double uclfl = clrurad[lev] * wtdiff, dclfl = clrdrad[lev] * wtdiff;
            clrurad[lev] = 0.0; clrdrad[lev] = 0.0;
            totuclfl[lev] = totuclfl[lev] + uclfl * delwave[iband - 1];
            totdclfl[lev] = totdclfl[lev] + dclfl * delwave[iband - 1];
        }
    }
    totuflux[0] = totuflux[0] * t->fluxfac; totdflux[0] = totdflux[0] * t->fluxfac;
    fnet[0] = totuflux[0] - totdflux[0];
    totuclfl[0] = totuclfl[0] * t->fluxfac; totdclfl[0] = totdclfl[0] * t->fluxfac;
    fnetc[0] = totuclfl[0] - totdclfl[0];
    for (int lev = 1; lev <= nlayers; ++lev) {
        totuflux[lev] = totuflux[lev] * t->fluxfac; totdflux[lev] = totdflux[lev] * t->fluxfac;
        fnet[lev] = totuflux[lev] - totdflux[lev];
        totuclfl[lev] = totuclfl[lev] * t->fluxfac; totdclfl[lev] = totdclfl[lev] * t->fluxfac;
        fnetc[lev] = totuclfl[lev] - totdclfl[lev];
        int l = lev - 1;
        htr[l] = t->heatfac * (fnet[l] - fnet[lev]) / (c->pz[l] - c->pz[lev]);
        htrc[l] = t->heatfac * (fnetc[l] - fnetc[lev]) / (c->pz[l] - c->pz[lev]);
    }
    htr[nlayers] = 0.0;
    htrc[nlayers] = 0.0;
}

/* rrtmg_lw for one column, LW:10942-11062 (+ inatm LW:11248-11340) */
int lw_oracle_rrtmg_lw(const LwTables *t, const LwColumnIn *in, LwColumnOut *out)
{
    const int nlayers = in->nlay;
    if (nlayers + 2 > MAXLAY) return 1;
    LwCol *c = (LwCol *)calloc(1, sizeof(LwCol));
    static const double avogad = 6.02214199e+23, grav = 9.8066, amd = 28.9660, amw = 18.0160;
    c->nlayers = nlayers;
    double amttl = 0.0, wvttl = 0.0;
    c->tbound = in->tsfc;
    c->pz[0] = in->plev[0];
    c->tz[0] = in->tlev[0];
    for (int l = 1; l <= nlayers; ++l) {
        c->pavel[l] = in->play[l - 1];
        c->tavel[l] = in->tlay[l - 1];
        c->pz[l] = in->plev[l];
        c->tz[l] = in->tlev[l];
        c->wkl[1][l] = in->h2ovmr[l - 1];
        c->wkl[2][l] = in->co2vmr[l - 1];
        c->wkl[3][l] = in->o3vmr[l - 1];
        c->wkl[4][l] = in->n2ovmr[l - 1];
        c->wkl[5][l] = 0.0;
        c->wkl[6][l] = in->ch4vmr[l - 1];
        c->wkl[7][l] = in->o2vmr[l - 1];
        double amm = (1. - c->wkl[1][l]) * amd + c->wkl[1][l] * amw;
        c->coldry[l] = (c->pz[l - 1] - c->pz[l]) * 1.e3 * avogad / (1.e2 * grav * amm * (1. + c->wkl[1][l]));
        c->wx[1][l] = in->ccl4vmr[l - 1];
        c->wx[2][l] = in->cfc11vmr[l - 1];
        c->wx[3][l] = in->cfc12vmr[l - 1];
        c->wx[4][l] = in->cfc22vmr[l - 1];
    }
    for (int l = 1; l <= nlayers; ++l) {
        double summol = 0.0;
        for (int imol = 2; imol <= 7; ++imol) summol = summol + c->wkl[imol][l];
        c->wbrodl[l] = c->coldry[l] * (1. - summol);
        for (int imol = 1; imol <= 7; ++imol) c->wkl[imol][l] = c->coldry[l] * c->wkl[imol][l];
        amttl = amttl + c->coldry[l] + c->wkl[1][l];
        wvttl = wvttl + c->wkl[1][l];
        for (int ix = 1; ix <= 4; ++ix) c->wx[ix][l] = c->coldry[l] * c->wx[ix][l] * 1.e-20;
    }
    double wvsh = (amw * wvttl) / (amd * amttl);
    c->pwvcm = wvsh * (1.e3 * c->pz[0]) / (1.e2 * grav);
    for (int n = 0; n < LW_NBND; ++n) c->semiss[n] = in->emis[n];

    setcoef_lw(c);
    taumol_lw(t, c);

    double(*taut)[MAXLAY] = (double(*)[MAXLAY])malloc(sizeof(double) * LW_NGPT * MAXLAY);
    int igc = 0;
    for (int ib = 0; ib < LW_NBND; ++ib)
        for (int ig = 0; ig < ngc[ib]; ++ig, ++igc)
            for (int k = 1; k <= nlayers; ++k)
                taut[igc][k] = c->taug[igc][k] + in->tauaer[ib * nlayers + k - 1];   /* LW:11004-11008 */

    double fu[MAXLAY], fd[MAXLAY], hr[MAXLAY];
    if (in->cldfrac) {
        /* mcica_subcol_lw (called by RRTMG_LWRAD with permuteseed = 150, LW:12555-12563) + cldprmc (LW:10975) */
        const size_t n = (size_t)LW_NGPT * nlayers;
        double *cldfmc = (double *)malloc(2 * n * sizeof(double)), *taucmc = cldfmc + n;
        int rc = oracle_mcica(LW_NGPT, nlayers, in->play, in->cldfrac, 150, in->seed_real4, cldfmc);
        if (!rc) rc = lw_oracle_cldprmc(nlayers, in->inflglw, in->iceflglw, in->liqflglw, cldfmc, in->ciwp, in->clwp,
                                        in->cswp, in->rei, in->rel, in->res, taucmc, NULL, 0);
        if (rc) { free(cldfmc); free(taut); free(c); return 10 + rc; }
        if (in->cldfmc_out) memcpy(in->cldfmc_out, cldfmc, n * sizeof(double));
        double fuc[MAXLAY], fdc[MAXLAY], hrc[MAXLAY];
        if (in->clean_atm_diag > 0) {
            rtrnmc_lw_cloudy(t, c, (const double(*)[MAXLAY])c->taug, cldfmc, taucmc, fu, fd, hr, fuc, fdc, hrc);
            for (int k = 0; k <= nlayers; ++k) { out->uflxcln[k] = fu[k]; out->dflxcln[k] = fd[k]; }
        } else {
            for (int k = 0; k <= nlayers; ++k) { out->uflxcln[k] = 0.0; out->dflxcln[k] = 0.0; }
        }
        rtrnmc_lw_cloudy(t, c, (const double(*)[MAXLAY])taut, cldfmc, taucmc, fu, fd, hr, fuc, fdc, hrc);
        for (int k = 0; k <= nlayers; ++k) {
            out->uflx[k] = fu[k]; out->dflx[k] = fd[k];
            out->uflxc[k] = fuc[k]; out->dflxc[k] = fdc[k];
        }
        for (int k = 0; k <= nlayers - 1; ++k) { out->hr[k] = hr[k]; out->hrc[k] = hrc[k]; }
        free(cldfmc); free(taut); free(c);
        return 0;
    }
    if (in->clean_atm_diag > 0) {
        rtrnmc_lw(t, c, (const double(*)[MAXLAY])c->taug, fu, fd, hr);                 /* LW:11023-11027 */
        for (int k = 0; k <= nlayers; ++k) { out->uflxcln[k] = fu[k]; out->dflxcln[k] = fd[k]; }
    } else {
        for (int k = 0; k <= nlayers; ++k) { out->uflxcln[k] = 0.0; out->dflxcln[k] = 0.0; }
    }
    rtrnmc_lw(t, c, (const double(*)[MAXLAY])taut, fu, fd, hr);                        /* LW:11040-11044 */
    for (int k = 0; k <= nlayers; ++k) {
        out->uflx[k] = fu[k]; out->dflx[k] = fd[k];
        out->uflxc[k] = fu[k]; out->dflxc[k] = fd[k];   /* cloud-free: clrurad = urad (LW:3305-3311) */
    }
    for (int k = 0; k <= nlayers - 1; ++k) { out->hr[k] = hr[k]; out->hrc[k] = hr[k]; }
    free(taut); free(c);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* RRTMG_LWRAD over a tile, LW:11876-12692 (cloud-free, WRF-Chem aerosol)    */
/* ------------------------------------------------------------------------ */
static const double PPROF[60] = {1000.00, 855.47, 731.82, 626.05, 535.57, 458.16, 391.94, 335.29, 286.83, 245.38,
                                 209.91, 179.57, 153.62, 131.41, 112.42, 96.17, 82.27, 70.38, 60.21, 51.51,
                                 44.06, 37.69, 32.25, 27.59, 23.60, 20.19, 17.27, 14.77, 12.64, 10.81,
                                 9.25, 7.91, 6.77, 5.79, 4.95, 4.24, 3.63, 3.10, 2.65, 2.27,
                                 1.94, 1.66, 1.42, 1.22, 1.04, 0.89, 0.76, 0.65, 0.56, 0.48,
                                 0.41, 0.35, 0.30, 0.26, 0.22, 0.19, 0.16, 0.14, 0.12, 0.10};
static const double TPROF[60] = {286.96, 281.07, 275.16, 268.11, 260.56, 253.02, 245.62, 238.41, 231.57, 225.91,
                                 221.72, 217.79, 215.06, 212.74, 210.25, 210.16, 210.69, 212.14, 213.74, 215.37,
                                 216.82, 217.94, 219.03, 220.18, 221.37, 222.64, 224.16, 225.88, 227.63, 229.51,
                                 231.50, 233.73, 236.18, 238.78, 241.60, 244.44, 247.35, 250.33, 253.32, 256.30,
                                 259.22, 262.12, 264.80, 266.50, 267.59, 268.44, 268.69, 267.76, 266.13, 263.96,
                                 261.54, 258.93, 256.15, 253.23, 249.89, 246.67, 243.48, 240.25, 236.66, 233.86};

int lw_oracle_nlayers(int kme, double p_top_pa)
{
    /* NLAYERS = kme + nint(p_top*0.01/deltap) - 1, LW:12861 */
    return kme + (int)lround(p_top_pa * 0.01 / 4.) - 1;
}

int lw_oracle_lwrad_tile(const LwTables *t, const LwTileIn *in, LwTileOut *out, char *err, int errlen)
{
    const int ni = in->ime - in->ims + 1, nkm = in->kme - in->kms + 1;
    const int nk = in->kte - in->kts + 1;
    const int nlayers = in->nlayers;
    const size_t sj3 = (size_t)ni * nkm, sj3f = (size_t)ni * (nkm + 2);
#define I3(i, k, j) ((size_t)((j) - in->jms) * sj3 + (size_t)((k) - in->kms) * ni + ((i) - in->ims))
#define I3F(i, k, j) ((size_t)((j) - in->jms) * sj3f + (size_t)((k) - in->kms) * ni + ((i) - in->ims))
#define I2(i, j) ((size_t)((j) - in->jms) * ni + ((i) - in->ims))
    const double co2 = 379.e-6, ch4 = 1774.e-9, n2o = 319.e-9, cfc11 = 0.251e-9, cfc12 = 0.538e-9,
                 cfc22 = 0.169e-9, ccl4 = 0.093e-9, o2 = 0.209488;
    const double amdw = 1.607793, amdo = 0.603461, deltap = 4., thresh = 1.e-9;
    if (nlayers + 2 > MAXLAY || nlayers < nk + 1) return 1;
    double *w = (double *)malloc(sizeof(double) * (size_t)(64 * MAXLAY));
    double *play = w, *plev = play + MAXLAY, *tlay = plev + MAXLAY, *tlev = tlay + MAXLAY, *varint = tlev + MAXLAY;
    double *h2ovmr = varint + MAXLAY, *o3vmr = h2ovmr + MAXLAY, *o3mmr = o3vmr + MAXLAY;
    double *gas = o3mmr + MAXLAY;   /* 8 constant profiles */
    double *tauaer = gas + 8 * MAXLAY;          /* [16][nlayers] */
    double *o_ = tauaer + 16 * MAXLAY;
    double *uflx = o_, *dflx = o_ + MAXLAY, *hr = o_ + 2 * MAXLAY, *uflxc = o_ + 3 * MAXLAY, *dflxc = o_ + 4 * MAXLAY,
           *hrc = o_ + 5 * MAXLAY, *uflxcln = o_ + 6 * MAXLAY, *dflxcln = o_ + 7 * MAXLAY;
    int rc = 0;
    for (int j = in->jts; j <= in->jte && !rc; ++j)
        for (int i = in->its; i <= in->ite && !rc; ++i) {
            /* 1-D profile, LW:12149-12166 */
            for (int k = 0; k <= nk; ++k) {
                plev[k] = in->p8w[I3(i, in->kts + k, j)] / 100.;
                tlev[k] = in->t8w[I3(i, in->kts + k, j)];
            }
            for (int k = 0; k < nk; ++k) {
                double qv = in->qv3d[I3(i, in->kts + k, j)];
                qv = qv > 0. ? qv : 0.;
                qv = qv > 1.E-12 ? qv : 1.E-12;
                play[k] = in->p3d[I3(i, in->kts + k, j)] / 100.;
                tlay[k] = in->t3d[I3(i, in->kts + k, j)];
                h2ovmr[k] = qv * amdw;
            }
            /* buffer layers every deltap mb up to 0 mb, LW:12207-12217 (0-based: level L <-> plev[L-1]) */
            for (int L = nk + 1; L <= nlayers; ++L) {
                plev[L] = plev[L - 1] - deltap;
                play[L - 1] = 0.5 * (plev[L - 1] + plev[L]);
            }
            plev[nlayers] = 0.00;
            play[nlayers - 1] = 0.5 * (plev[nlayers - 1] + plev[nlayers]);
            /* temperatures of the buffer from the 60-level mean profile, LW:12219-12252 */
            for (int L = 1; L <= nlayers + 1; ++L) {
                int klev = 60;
                double p = plev[L - 1];
                if (PPROF[59] < p) {
                    for (int LL = 2; LL <= 60; ++LL)
                        if (PPROF[LL - 1] < p) { klev = LL - 1; break; }
                }
                double vark, vark1, wght;
                if (klev != 60) {
                    vark = TPROF[klev - 1]; vark1 = TPROF[klev];
                    wght = (p - PPROF[klev - 1]) / (PPROF[klev] - PPROF[klev - 1]);
                } else { vark = TPROF[klev - 1]; vark1 = TPROF[klev - 1]; wght = 0.0; }
                varint[L - 1] = wght * (vark1 - vark) + vark;
            }
            for (int L = nk + 1; L <= nlayers + 1; ++L) {
                tlev[L - 1] = varint[L - 1] + (tlev[nk - 1] - varint[nk - 1]);
                tlay[L - 2] = 0.5 * (tlev[L - 1] + tlev[L - 2]);
            }
            for (int L = nk + 1; L <= nlayers; ++L) h2ovmr[L - 1] = h2ovmr[nk - 1];
            for (int k = 0; k < nlayers; ++k) {
                gas[0 * MAXLAY + k] = co2; gas[1 * MAXLAY + k] = ch4; gas[2 * MAXLAY + k] = n2o; gas[3 * MAXLAY + k] = o2;
                gas[4 * MAXLAY + k] = cfc11; gas[5 * MAXLAY + k] = cfc12; gas[6 * MAXLAY + k] = cfc22; gas[7 * MAXLAY + k] = ccl4;
            }
            oracle_inirad(o3mmr, plev, nlayers - 1);
            for (int k = 0; k < nlayers; ++k) o3vmr[k] = o3mmr[k] * amdo;
            double emis[LW_NBND];
            for (int nb = 0; nb < LW_NBND; ++nb) emis[nb] = in->emiss[I2(i, j)];
            /* aerosol, LW:12576-12630 */
            for (int nb = 0; nb < LW_NBND; ++nb)
                for (int k = 0; k < nlayers; ++k) tauaer[nb * nlayers + k] = 0.;
            if (in->aer_ra_feedback == 1) {
                for (int k = 0; k < nk; ++k) {
                    size_t q = I3(i, in->kts + k, j);
                    if (in->tauaerlw[0][q] > thresh && in->tauaerlw[15][q] > thresh)
                        for (int nb = 0; nb < LW_NBND; ++nb) tauaer[nb * nlayers + k] = in->tauaerlw[nb][q];
                }
                for (int nb = 0; nb < LW_NBND && !rc; ++nb) {
                    double slope = 0.;
                    for (int k = 0; k < nk; ++k) slope = slope + tauaer[nb * nlayers + k];
                    if (slope < 0.) {
                        if (err)
                            snprintf(err, errlen,
                                     "ERROR: Negative total lw optical depth of %8.2f at point i,j,nb=%5d%5d%5d", slope,
                                     i, j, nb + 1);
                        rc = 2;
                    }
                }
                if (rc) break;
            }
            LwColumnIn ci = {nlayers, play, plev, tlay, tlev, in->tsk[I2(i, j)], h2ovmr, o3vmr, gas, gas + MAXLAY,
                             gas + 2 * MAXLAY, gas + 3 * MAXLAY, gas + 4 * MAXLAY, gas + 5 * MAXLAY, gas + 6 * MAXLAY,
                             gas + 7 * MAXLAY, emis, tauaer, in->clean_atm_diag};
            LwColumnOut co = {uflx, dflx, hr, uflxc, dflxc, hrc, uflxcln, dflxcln};
            double cst[7][MAXLAY], cin[9][MAXLAY], t1d[MAXLAY], pdel[MAXLAY];
            if (in->icloud != 0) {
                /* hydrometeors -> in-cloud state, LW:11921-12140, 12335-12553 (same block as the SW wrapper; relcalc
                 * and reicalc see tlay with the top model layer already overwritten, LW:12246-12250) */
                const double *src[9] = {in->cldfra3d, in->qc3d, in->qr3d, in->qi3d, in->qs3d, in->re_cloud,
                                        in->re_ice, in->re_snow, in->qndrop3d};
                for (int s = 0; s < 9; ++s)
                    if (src[s]) for (int k = 0; k < nk; ++k) cin[s][k] = src[s][I3(i, in->kts + k, j)];
                for (int k = 0; k < nk; ++k) {
                    pdel[k] = plev[k] - plev[k + 1];
                    t1d[k] = in->t3d[I3(i, in->kts + k, j)];
                }
                OracleCloudWrf cw = {nk, t1d, pdel, tlay,
                                     src[0] ? cin[0] : NULL, src[1] ? cin[1] : NULL, src[2] ? cin[2] : NULL,
                                     src[3] ? cin[3] : NULL, src[4] ? cin[4] : NULL, src[5] ? cin[5] : NULL,
                                     src[6] ? cin[6] : NULL, src[7] ? cin[7] : NULL, src[8] ? cin[8] : NULL,
                                     in->has_reqc, in->has_reqi, in->has_reqs, in->warm_rain, in->progn,
                                     in->xland ? in->xland[I2(i, j)] : 1., in->xice ? in->xice[I2(i, j)] : 0.,
                                     in->snow ? in->snow[I2(i, j)] : 0., in->g};
                    double crad[2][MAXLAY];
                    cw.f_ice_phy = in->f_ice_phy; cw.fer_hires = in->fer_hires;
                    cw.iradius = NULL; cw.lradius = NULL;
                    if (in->iradius && in->lradius) {
                        for (int k = 0; k < nk; ++k) {
                            crad[0][k] = in->iradius[I3(i, in->kts + k, j)];
                            crad[1][k] = in->lradius[I3(i, in->kts + k, j)];
                        }
                        cw.iradius = crad[0]; cw.lradius = crad[1];
                    }
                int fl[3];
                if (oracle_cloud_state(&cw, cst[0], cst[1], cst[2], cst[3], cst[4], cst[5], cst[6], fl)) { rc = 1; break; }
                for (int k = nk; k < nlayers; ++k) {   /* LW:12543-12553 */
                    cst[0][k] = 0.; cst[1][k] = 0.; cst[2][k] = 0.; cst[3][k] = 0.;
                    cst[4][k] = 10.; cst[5][k] = 10.; cst[6][k] = 10.;
                }
                ci.cldfrac = cst[0]; ci.clwp = cst[1]; ci.ciwp = cst[2]; ci.cswp = cst[3];
                ci.rel = cst[4]; ci.rei = cst[5]; ci.res = cst[6];
                ci.inflglw = fl[0]; ci.iceflglw = fl[1]; ci.liqflglw = fl[2];
                ci.seed_real4 = in->seed_real4;
            }
            if (lw_oracle_rrtmg_lw(t, &ci, &co)) { rc = 1; break; }
            size_t q2 = I2(i, j);
            if (out->glw) out->glw[q2] = dflx[0];
            if (out->olr) out->olr[q2] = uflx[nlayers];
            if (out->lwcf) out->lwcf[q2] = uflxc[nlayers] - uflx[nlayers];
            if (out->lwupt) {
                out->lwupt[q2] = uflx[nlayers]; out->lwuptc[q2] = uflxc[nlayers]; out->lwdnt[q2] = dflx[nlayers];
                out->lwdntc[q2] = dflxc[nlayers]; out->lwupb[q2] = uflx[0]; out->lwupbc[q2] = uflxc[0];
                out->lwdnb[q2] = dflx[0]; out->lwdnbc[q2] = dflxc[0]; out->lwuptcln[q2] = uflxcln[nlayers];
                out->lwdntcln[q2] = dflxcln[nlayers]; out->lwupbcln[q2] = uflxcln[0]; out->lwdnbcln[q2] = dflxcln[0];
            }
            if (out->lwupflx)
                for (int k = 0; k <= nk + 1; ++k) {   /* kts:kte+2 */
                    size_t q = I3F(i, in->kts + k, j);
                    out->lwupflx[q] = uflx[k]; out->lwupflxc[q] = uflxc[k]; out->lwdnflx[q] = dflx[k];
                    out->lwdnflxc[q] = dflxc[k]; out->lwupflxcln[q] = uflxcln[k]; out->lwdnflxcln[q] = dflxcln[k];
                }
            if (out->rthratenlw)
                for (int k = 0; k < nk; ++k) {
                    double tten = hr[k] / 86400.;
                    out->rthratenlw[I3(i, in->kts + k, j)] = tten / in->pi3d[I3(i, in->kts + k, j)];
                }
        }
    free(w);
    return rc;
#undef I3
#undef I3F
#undef I2
}
