/*
 * oracle/rrtmg_sw_cloud_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the shortwave McICA cloud path of the reference (phys/module_ra_rrtmg_sw.F):
 *   kissvec                         SW:1900-1932   KISS random numbers (32-bit wrap-around integer arithmetic)
 *   generate_stochastic_clouds_sw   SW:1517-1897   112 sub-columns of binary cloud, maximum-random overlap
 *   mcica_subcol_sw                 SW:1392-1513
 *   cldprmc_sw                      SW:1969-2390   cloud optical properties (inflag >= 2; liqflag 1; iceflag 3, 5)
 * Pinned against the reference Fortran itself (oracle/_ref/ref_sw ... clouds; tests/test_oracle_golden_cloud.py).
 */
#include "rrtmg_oracle.h"

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "sw_cld_tables.h"

static const int ngc_sw[SW_NBND] = {6, 12, 8, 8, 10, 10, 2, 10, 8, 6, 6, 8, 6, 12};

/* kissvec, SW:1900-1932, one column.  Fortran default integers wrap modulo 2^32 in the reference build; ishft is
 * a logical shift of the 32-bit pattern. */
typedef struct { int32_t s1, s2, s3, s4; } Kiss;

static double kiss_next(Kiss *k)
{
    uint32_t s1 = (uint32_t)k->s1, s2 = (uint32_t)k->s2, s3 = (uint32_t)k->s3, s4 = (uint32_t)k->s4;
    s1 = 69069u * s1 + 1327217885u;
    s2 ^= s2 << 13;
    s2 ^= s2 >> 17;
    s2 ^= s2 << 5;
    s3 = 18000u * (s3 & 65535u) + (s3 >> 16);
    s4 = 30903u * (s4 & 65535u) + (s4 >> 16);
    k->s1 = (int32_t)s1; k->s2 = (int32_t)s2; k->s3 = (int32_t)s3; k->s4 = (int32_t)s4;
    int32_t kiss = (int32_t)(s1 + s2 + (s3 << 16) + s4);
    return (double)kiss * 2.328306e-10 + 0.5;
}

/* seed from the fractional part of a mid-layer pressure in Pa, SW:1696-1699.
 * real4 != 0 reproduces a stock (REAL*4) WRF build, where play, pmid and the product are single precision. */
static int32_t kiss_seed(double play_mb, int real4)
{
    if (real4) {
        float pmid = (float)play_mb * 1.e2f;
        float v = (pmid - (float)(int32_t)pmid) * (float)1000000000;
        return (int32_t)v;
    }
    double pmid = play_mb * 1.e2;
    return (int32_t)((pmid - (double)(int32_t)pmid) * 1000000000.);
}

/* mcica_subcol_sw + generate_stochastic_clouds_sw for one column, icld = 2 (maximum-random), irng = 0.
 * play [nlay] in mb bottom->top, cldfrac [nlay]; cldfmc [SW_NGPT][nlay] (0 or 1).  Returns nonzero on the
 * reference's STOP conditions. */
int sw_oracle_mcica(int nlay, const double *play, const double *cldfrac, int change_seed, int real4, double *cldfmc)
{
    const double cldmin = 1.0e-20;
    if (nlay < 4 || nlay > 256) return 1;
    if (play[0] * 1.e2 < play[1] * 1.e2) return 2;   /* KISSVEC SEED GENERATOR REQUIRES PMID FROM BOTTOM FOUR LAYERS */
    double cldf[256];
    for (int l = 0; l < nlay; ++l) cldf[l] = cldfrac[l] < cldmin ? 0. : cldfrac[l];
    Kiss k = {kiss_seed(play[0], real4), kiss_seed(play[1], real4), kiss_seed(play[2], real4), kiss_seed(play[3], real4)};
    for (int i = 0; i < change_seed; ++i) (void)kiss_next(&k);
    static _Thread_local double cdf[SW_NGPT][256];
    for (int is = 0; is < SW_NGPT; ++is)
        for (int l = 0; l < nlay; ++l) cdf[is][l] = kiss_next(&k);
    for (int l = 1; l < nlay; ++l)
        for (int is = 0; is < SW_NGPT; ++is) {
            if (cdf[is][l - 1] > 1. - cldf[l - 1]) cdf[is][l] = cdf[is][l - 1];
            else cdf[is][l] = cdf[is][l] * (1. - cldf[l - 1]);
        }
    for (int is = 0; is < SW_NGPT; ++is)
        for (int l = 0; l < nlay; ++l) cldfmc[is * nlay + l] = cdf[is][l] >= 1. - cldf[l] ? 1. : 0.;
    return 0;
}

static int fail(char *err, int errlen, const char *msg)
{
    if (err && errlen > 0) snprintf(err, errlen, "%s", msg);
    return 1;
}

/* cldprmc_sw, SW:1969-2390, inflag >= 2 (the only form the WRF wrapper uses), liqflag = 1, iceflag = 3 or 5.
 * Inputs per layer (in-cloud paths g m-2, sizes um) and the sub-column mask; outputs [SW_NGPT][nlay]:
 * taucmc (delta-scaled), ssacmc, asmcmc, taormc (unscaled).  Uncloudy sub-columns keep the generator's values
 * (tau 0, ssa 1, g 0). */
int sw_oracle_cldprmc(int nlay, int inflag, int iceflag, int liqflag, const double *cldfmc, const double *ciwp,
                      const double *clwp, const double *cswp, const double *rei, const double *rel, const double *res,
                      double *taucmc, double *ssacmc, double *asmcmc, double *taormc, char *err, int errlen)
{
    const double cldmin = 1.e-20;
    if (inflag < 2 || liqflag != 1 || (iceflag != 3 && iceflag != 5))
        return fail(err, errlen, "oracle cldprmc_sw: only inflag>=2, liqflag=1, iceflag=3|5 are restated");
    int ig = 0;
    for (int ib = 0; ib < SW_NBND; ++ib)
        for (int jg = 0; jg < ngc_sw[ib]; ++jg, ++ig)
            for (int lay = 0; lay < nlay; ++lay) {
                const size_t q = (size_t)ig * nlay + lay;
                taucmc[q] = 0.; ssacmc[q] = 1.; asmcmc[q] = 0.; taormc[q] = 0.;
                const double cf = cldfmc[q];
                const double ciwpmc = cf > 0. ? ciwp[lay] : 0., clwpmc = cf > 0. ? clwp[lay] : 0.;
                const double cswpmc = (cf > 0. && iceflag == 5) ? cswp[lay] : 0.;
                const double cwp = ciwpmc + clwpmc + cswpmc;
                if (!(cf >= cldmin && (cwp >= cldmin || taucmc[q] >= cldmin))) continue;
                double extcoice = 0., ssacoice = 0., gice = 0., forwice = 0.;
                double extcosno = 0., ssacosno = 0., gsno = 0., forwsno = 0.;
                double extcoliq = 0., ssacoliq = 0., gliq = 0., forwliq = 0.;
                if ((ciwpmc + cswpmc) == 0.0) {
                    /* all zero */
                } else {
                    const double radice = rei[lay];
                    if (radice < 5.0 || radice > 140.0)
                        return fail(err, errlen, "ERROR: ICE GENERALIZED EFFECTIVE SIZE OUT OF BOUNDS");
                    double factor = (radice - 2.) / 3.;
                    int index = (int)factor;
                    if (index == 46) index = 45;
                    double fint = factor - (double)index;
                    const double *e = SWCLD_EXTICE3[ib], *s = SWCLD_SSAICE3[ib], *a = SWCLD_ASYICE3[ib], *f = SWCLD_FDLICE3[ib];
                    extcoice = e[index - 1] + fint * (e[index] - e[index - 1]);
                    ssacoice = s[index - 1] + fint * (s[index] - s[index - 1]);
                    gice = a[index - 1] + fint * (a[index] - a[index - 1]);
                    double fdelta = f[index - 1] + fint * (f[index] - f[index - 1]);
                    if (fdelta < 0.0) return fail(err, errlen, "FDELTA LESS THAN 0.0");
                    if (fdelta > 1.0) return fail(err, errlen, "FDELTA GT THAN 1.0");
                    forwice = fdelta + 0.5 / ssacoice;
                    if (forwice > gice) forwice = gice;
                }
                if (cswpmc > 0.0 && iceflag == 5) {
                    const double radsno = res[lay];
                    if (radsno < 5.0 || radsno > 140.0)
                        return fail(err, errlen, "ERROR: SNOW GENERALIZED EFFECTIVE SIZE OUT OF BOUNDS");
                    double factor = (radsno - 2.) / 3.;
                    int index = (int)factor;
                    if (index == 46) index = 45;
                    double fint = factor - (double)index;
                    const double *e = SWCLD_EXTICE3[ib], *s = SWCLD_SSAICE3[ib], *a = SWCLD_ASYICE3[ib], *f = SWCLD_FDLICE3[ib];
                    extcosno = e[index - 1] + fint * (e[index] - e[index - 1]);
                    ssacosno = s[index - 1] + fint * (s[index] - s[index - 1]);
                    gsno = a[index - 1] + fint * (a[index] - a[index - 1]);
                    double fdelta = f[index - 1] + fint * (f[index] - f[index - 1]);
                    if (fdelta < 0.0) return fail(err, errlen, "FDELTA LESS THAN 0.0");
                    if (fdelta > 1.0) return fail(err, errlen, "FDELTA GT THAN 1.0");
                    forwsno = fdelta + 0.5 / ssacosno;
                    if (forwsno > gsno) forwsno = gsno;
                }
                if (clwpmc == 0.0) {
                    /* all zero */
                } else {
                    const double radliq = rel[lay];
                    if (radliq < 1.5 || radliq > 60.) return fail(err, errlen, "liquid effective radius out of bounds");
                    int index = (int)(radliq - 1.5);
                    if (index == 0) index = 1;
                    if (index == 58) index = 57;
                    double fint = radliq - 1.5 - (double)index;
                    const double *e = SWCLD_EXTLIQ1[ib], *s = SWCLD_SSALIQ1[ib], *a = SWCLD_ASYLIQ1[ib];
                    extcoliq = e[index - 1] + fint * (e[index] - e[index - 1]);
                    ssacoliq = s[index - 1] + fint * (s[index] - s[index - 1]);
                    if (fint < 0. && ssacoliq > 1.) ssacoliq = s[index - 1];
                    gliq = a[index - 1] + fint * (a[index] - a[index - 1]);
                    forwliq = gliq * gliq;
                }
                double tauliqorig = clwpmc * extcoliq, tauiceorig = ciwpmc * extcoice, tausnoorig = 0.;
                double ssaliq = ssacoliq * (1. - forwliq) / (1. - forwliq * ssacoliq);
                double tauliq = (1. - forwliq * ssacoliq) * tauliqorig;
                double ssaice = ssacoice * (1. - forwice) / (1. - forwice * ssacoice);
                double tauice = (1. - forwice * ssacoice) * tauiceorig;
                double scatliq = ssaliq * tauliq, scatice = ssaice * tauice, scatsno = 0.0;
                if (iceflag < 5) {
                    taormc[q] = tauliqorig + tauiceorig;
                    taucmc[q] = tauliq + tauice;
                } else {
                    tausnoorig = cswpmc * extcosno;
                    taormc[q] = tauliqorig + tauiceorig + tausnoorig;
                    double ssasno = ssacosno * (1. - forwsno) / (1. - forwsno * ssacosno);
                    double tausno = (1. - forwsno * ssacosno) * tausnoorig;
                    scatsno = ssasno * tausno;
                    taucmc[q] = tauliq + tauice + tausno;
                }
                if (taucmc[q] == 0.) taucmc[q] = cldmin;
                if (scatice == 0.) scatice = cldmin;
                if (scatsno == 0.) scatsno = cldmin;
                if (iceflag < 5) ssacmc[q] = (scatliq + scatice) / taucmc[q];
                else ssacmc[q] = (scatliq + scatice + scatsno) / taucmc[q];
                if (iceflag == 3) {
                    asmcmc[q] = (1.0 / (scatliq + scatice)) *
                                (scatliq * (gliq - forwliq) / (1.0 - forwliq) + scatice * ((gice - forwice) / (1.0 - forwice)));
                } else {
                    asmcmc[q] = (1.0 / (scatliq + scatice + scatsno)) *
                                (scatliq * (gliq - forwliq) / (1.0 - forwliq) +
                                 scatice * ((gice - forwice) / (1.0 - forwice)) +
                                 scatsno * ((gsno - forwsno) / (1.0 - forwsno)));
                }
            }
    return 0;
}
