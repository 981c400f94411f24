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
 * play [nlay] in mb bottom->top, cldfrac [nlay]; cldfmc [nsub][nlay] (0 or 1).  Returns nonzero on the
 * reference's STOP conditions.  The longwave generator (generate_stochastic_clouds, LW:2208-2578) is the same
 * code with 140 sub-columns. */
int oracle_mcica(int nsub, int nlay, const double *play, const double *cldfrac, int change_seed, int real4,
                 double *cldfmc)
{
    const double cldmin = 1.0e-20;
    if (nlay < 4 || nlay > 256 || nsub > 140) return 1;
    if (play[0] * 1.e2 < play[1] * 1.e2) return 2;   /* KISSVEC SEED GENERATOR REQUIRES PMID FROM BOTTOM FOUR LAYERS */
    double cldf[256];
    for (int l = 0; l < nlay; ++l) cldf[l] = cldfrac[l] < cldmin ? 0. : cldfrac[l];
    Kiss k = {kiss_seed(play[0], real4), kiss_seed(play[1], real4), kiss_seed(play[2], real4), kiss_seed(play[3], real4)};
    for (int i = 0; i < change_seed; ++i) (void)kiss_next(&k);
    static _Thread_local double cdf[140][256];
    for (int is = 0; is < nsub; ++is)
        for (int l = 0; l < nlay; ++l) cdf[is][l] = kiss_next(&k);
    for (int l = 1; l < nlay; ++l)
        for (int is = 0; is < nsub; ++is) {
            if (cdf[is][l - 1] > 1. - cldf[l - 1]) cdf[is][l] = cdf[is][l - 1];
            else cdf[is][l] = cdf[is][l] * (1. - cldf[l - 1]);
        }
    for (int is = 0; is < nsub; ++is)
        for (int l = 0; l < nlay; ++l) cldfmc[is * nlay + l] = cdf[is][l] >= 1. - cldf[l] ? 1. : 0.;
    return 0;
}

int sw_oracle_mcica(int nlay, const double *play, const double *cldfrac, int change_seed, int real4, double *cldfmc)
{
    return oracle_mcica(SW_NGPT, nlay, play, cldfrac, change_seed, real4, cldfmc);
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
    if (inflag < 2 || liqflag != 1 || iceflag < 3 || iceflag > 5)
        return fail(err, errlen, "oracle cldprmc_sw: only inflag>=2, liqflag=1, iceflag=3|4|5 are restated");
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
                if (iceflag == 3 || iceflag == 4) {
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

/* ---------------------------------------------------------------------------------------------------------
 * In-cloud state of one column as RRTMG_SWRAD / RRTMG_LWRAD build it from the WRF hydrometeor fields
 * (SW:10371-10618 and SW:10707-10947; the LW wrapper has the same block, LW:11921-12140, 12306-12480).
 * PARITY UNPINNED: the wrapper modules cannot be compiled here; restated from the source text.
 * Arrays over the nk model layers, bottom -> top.  Outputs have nk+1 entries (extra top layer: no cloud).
 * Not restated (refused by the engine as well): is_cammgmp_used, the Ferrier/HWRF F_ICE_PHY variants.
 * --------------------------------------------------------------------------------------------------------- */
#include "cld_retab.h"

/* relcalc, LW:14403-14462: liquid effective radius (um) from temperature and surface type.  Pinned bitwise on the
 * reference routine (oracle/_ref/ref_wrapper, tests/golden/wrapper_helpers.npz). */
double oracle_relcalc(double t, double landm, double icefrac, double snowh)
{
    const double tmelt = 273.16, rliqocean = 14.0, rliqice = 14.0, rliqland = 8.0;
    double rel = rliqland + (rliqocean - rliqland) * fmin(1.0, fmax(0.0, (tmelt - t) * 0.05));
    rel = rel + (rliqocean - rel) * fmin(1.0, fmax(0.0, snowh * 10.));
    rel = rel + (rliqocean - rel) * fmin(1.0, fmax(0.0, 1.0 - landm));
    rel = rel + (rliqice - rel) * fmin(1.0, fmax(0.0, icefrac));
    return rel;
}

/* reicalc, LW:14464-14491: ice effective radius (um) from the retab table (LW:11424-11442).  Pinned likewise. */
double oracle_reicalc(double t)
{
    int index = (int)(t - 179.);
    index = index < 1 ? 1 : (index > 94 ? 94 : index);
    const double corr = t - (double)(int)t;
    return CLD_RETAB[index - 1] * (1. - corr) + CLD_RETAB[index] * corr;
}

int oracle_cloud_state(const OracleCloudWrf *w, double *cldfrac, double *clwp, double *ciwp, double *cswp,
                       double *rel, double *rei, double *res, int *flags3)
{
    const int nk = w->nk;
    double qc1d[256], qr1d[256], qi1d[256], qs1d[256], recloud1d[256], reice1d[256], resnow1d[256];
    if (nk > 255) return 1;
    for (int k = 0; k < nk; ++k) {
        qc1d[k] = w->qc ? fmax(0., w->qc[k]) : 0.;
        qr1d[k] = w->qr ? fmax(0., w->qr[k]) : 0.;
        qi1d[k] = 0.; qs1d[k] = 0.;
    }
    if (!w->qi && !w->warm_rain)   /* no ice species: split by temperature, SW:10413-10422 */
        for (int k = 0; k < nk; ++k)
            if (w->t[k] < 273.15) { qi1d[k] = qc1d[k]; qs1d[k] = qr1d[k]; qc1d[k] = 0.; qr1d[k] = 0.; }
    if (w->qi) for (int k = 0; k < nk; ++k) qi1d[k] = fmax(0., w->qi[k]);
    if (w->qs) for (int k = 0; k < nk; ++k) qs1d[k] = fmax(0., w->qs[k]);
    if (w->f_ice_phy)   /* "MP option 5", SW:10461-10472 (LW:12003-12014) */
        for (int k = 0; k < nk; ++k) {
            const double s3 = w->qs ? w->qs[k] : 0.;
            qi1d[k] = fmax(0., 0.1 * s3); qs1d[k] = 0.9 * s3;
            qc1d[k] = w->qc ? fmax(0., w->qc[k]) : 0.;   /* qc1d = qc3d again, SW:10467 / LW:12009 */
        }
    if (w->fer_hires)   /* qi3d saves all frozen water, SW:10474-10492 (LW:12016-12034) */
        for (int k = 0; k < nk; ++k) {
            qi1d[k] = w->qi ? fmax(0., w->qi[k]) : 0.; qs1d[k] = 0.0;
            qc1d[k] = w->qc ? fmax(0., w->qc[k]) : 0.;   /* SW:10487 / LW:12029 */
        }
    int inflg = 2, iceflg = 3, liqflg = 1;
    for (int k = 0; k < nk; ++k) {
        const double cf3 = w->cldfra ? w->cldfra[k] : 0.;
        if (w->has_reqc != 0) {   /* SW:10519-10533 */
            inflg = 3;
            double r = fmax(2.5, w->re_cloud[k] * 1.E6);
            if (r <= 2.5 && cf3 > 0. && (w->xland - 1.5) > 0.) r = 10.5;
            else if (r <= 2.5 && cf3 > 0. && (w->xland - 1.5) < 0.) r = 7.5;
            recloud1d[k] = r;
        } else recloud1d[k] = 5.0;
        if (w->has_reqi != 0) {   /* SW:10543-10557 */
            inflg = 4; iceflg = 4;
            double r = fmax(5., w->re_ice[k] * 1.E6);
            if (r <= 5. && cf3 > 0.) {
                int idx = (int)(w->t[k] - 179.);
                idx = idx < 1 ? 1 : (idx > 75 ? 75 : idx);
                double corr = w->t[k] - (double)(int)w->t[k];
                r = CLD_RETAB[idx - 1] * (1. - corr) + CLD_RETAB[idx] * corr;
                r = fmax(r, 5.0);
            }
            reice1d[k] = r;
        } else reice1d[k] = 10.;
        if (w->has_reqs != 0) {   /* SW:10563-10568 */
            inflg = 5; iceflg = 5;
            resnow1d[k] = fmax(10., w->re_snow[k] * 1.E6);
        } else resnow1d[k] = 10.0;
    }
    if (w->has_reqs == 0 && w->has_reqi != 0 && w->has_reqc != 0) {   /* SW:10595-10604 */
        inflg = 5; iceflg = 5;
        for (int k = 0; k < nk; ++k) {
            resnow1d[k] = fmax(10., w->re_ice[k] * 1.E6);
            qs1d[k] = w->qi ? w->qi[k] : 0.;
            qi1d[k] = 0.;
            reice1d[k] = 10.;
        }
    }
    /* SW:10743-10947 */
    const double landfrac = 2. - w->xland, landm = landfrac, snowh = 0.001 * w->snow, icefrac = w->xice;
    for (int k = 0; k < nk; ++k) {
        const double cf = w->cldfra ? w->cldfra[k] : 0.;
        cldfrac[k] = cf;
        double gicewp = (qi1d[k] + qs1d[k]) * w->pdel[k] * 100.0 / w->g * 1000.0;
        double gliqwp = qc1d[k] * w->pdel[k] * 100.0 / w->g * 1000.0;
        double cicewp = gicewp / fmax(0.01, cf), cliqwp = gliqwp / fmax(0.01, cf), csnowp = 0.;
        if (iceflg >= 4) {
            gicewp = qi1d[k] * w->pdel[k] * 100.0 / w->g * 1000.0;
            cicewp = gicewp / fmax(0.01, cf);
        }
        if (iceflg == 5) {
            double snow_mass_factor = 1.0;
            if (resnow1d[k] > 130.) {
                snow_mass_factor = (130.0 / resnow1d[k]) * (130.0 / resnow1d[k]);
                resnow1d[k] = 130.0;
            }
            double gsnowp = qs1d[k] * snow_mass_factor * w->pdel[k] * 100.0 / w->g * 1000.0;
            csnowp = gsnowp / fmax(0.01, cf);
        }
        double reliq;
        if (w->progn == 1) {   /* prognostic droplet number, SW:10797-10816 */
            const double pi = 4. * atan(1.0), third = 1. / 3., rhoh2o = 1.e3, relconst = 3 / (4. * pi * rhoh2o);
            const double lwpmin = 3.e-5;
            reliq = 10.;
            if (w->qndrop && qc1d[k] * w->pdel[k] > lwpmin && w->qndrop[k] > 1000.) {
                reliq = pow(relconst * qc1d[k] / w->qndrop[k], third);
                reliq = 1.1 * reliq;
                reliq = reliq * 1.e6;
                reliq = fmax(reliq, 4.);
                reliq = fmin(reliq, 20.);
            }
        } else {
            reliq = oracle_relcalc(w->t_rad ? w->t_rad[k] : w->t[k], landm, icefrac, snowh);
        }
        double reice = oracle_reicalc(w->t_rad ? w->t_rad[k] : w->t[k]);
        if (inflg >= 3) reliq = recloud1d[k];
        if (iceflg >= 4) reice = reice1d[k];
        if (iceflg == 3) { reice = reice * 1.0315; reice = fmin(140.0, reice); }
        if (w->iradius) {   /* CAM-MG microphysics supplies the radii, SW:10883-10898 (LW:12473-12488) */
            reice = (qi1d[k] > 1.e-20 || qs1d[k] > 1.e-20) ? w->iradius[k] : 25.;
            reice = fmax(5., fmin(140.0, reice));
            reliq = qc1d[k] > 1.e-20 ? w->lradius[k] : 10.;
            reliq = fmax(2.5, fmin(60.0, reliq));
        }
        clwp[k] = cliqwp; ciwp[k] = cicewp; rel[k] = reliq; rei[k] = reice;
        if (inflg == 5) { cswp[k] = csnowp; res[k] = resnow1d[k]; }
        else { cswp[k] = 0.0; res[k] = 10.0; }
    }
    clwp[nk] = 0.; ciwp[nk] = 0.; cswp[nk] = 0.; rel[nk] = 10.; rei[nk] = 10.; res[nk] = 10.; cldfrac[nk] = 0.;
    flags3[0] = inflg; flags3[1] = iceflg; flags3[2] = liqflg;
    return 0;
}

/* cldprmc (longwave), LW:2620-2914, inflag >= 2, liqflag = 1, iceflag = 3, 4 or 5: cloud absorption optical
 * depth per sub-column and layer, taucmc [LW_NGPT][nlay]. */
#include "lw_cld_tables.h"
static const int ngc_lw[LW_NBND] = {10, 12, 16, 14, 16, 8, 12, 8, 12, 6, 8, 8, 4, 2, 2, 2};

int lw_oracle_cldprmc(int nlay, int inflag, int iceflag, int liqflag, const double *cldfmc, const double *ciwp,
                      const double *clwp, const double *cswp, const double *rei, const double *rel, const double *res,
                      double *taucmc, char *err, int errlen)
{
    const double cldmin = 1.e-20;
    if (inflag < 2 || liqflag != 1 || iceflag < 3 || iceflag > 5)
        return fail(err, errlen, "oracle cldprmc: only inflag>=2, liqflag=1, iceflag=3|4|5 are restated");
    int ig = 0;
    for (int ib = 0; ib < LW_NBND; ++ib)
        for (int jg = 0; jg < ngc_lw[ib]; ++jg, ++ig)
            for (int lay = 0; lay < nlay; ++lay) {
                const size_t q = (size_t)ig * nlay + lay;
                taucmc[q] = 0.;
                const double cf = cldfmc[q];
                const double ciwpmc = cf > 0. ? ciwp[lay] : 0., clwpmc = cf > 0. ? clwp[lay] : 0.;
                const double cswpmc = cf > 0. ? cswp[lay] : 0.;
                const double cwp = ciwpmc + clwpmc + cswpmc;
                if (!(cf >= cldmin && (cwp >= cldmin || taucmc[q] >= cldmin))) continue;
                double abscoice = 0., abscosno = 0., abscoliq = 0.;
                if ((ciwpmc + cswpmc) == 0.0) {
                } else {
                    const double radice = rei[lay];
                    if (radice < 5.0 || radice > 140.0)
                        return fail(err, errlen, "ERROR: ICE GENERALIZED EFFECTIVE SIZE OUT OF BOUNDS");
                    double factor = (radice - 2.) / 3.;
                    int index = (int)factor;
                    if (index == 46) index = 45;
                    double fint = factor - (double)index;
                    const double *a = LWCLD_ABSICE3[ib];
                    abscoice = a[index - 1] + fint * (a[index] - (a[index - 1]));
                }
                if (cswpmc > 0.0 && iceflag == 5) {
                    const double radsno = res[lay];
                    if (radsno < 5.0 || radsno > 140.0)
                        return fail(err, errlen, "ERROR: SNOW GENERALIZED EFFECTIVE SIZE OUT OF BOUNDS");
                    double factor = (radsno - 2.) / 3.;
                    int index = (int)factor;
                    if (index == 46) index = 45;
                    double fint = factor - (double)index;
                    const double *a = LWCLD_ABSICE3[ib];
                    abscosno = a[index - 1] + fint * (a[index] - (a[index - 1]));
                }
                if (clwpmc == 0.0) {
                } else {
                    const double radliq = rel[lay];
                    if (radliq < 2.5 || radliq > 60.) return fail(err, errlen, "LIQUID EFFECTIVE RADIUS OUT OF BOUNDS");
                    int index = (int)(radliq - 1.5);
                    if (index == 0) index = 1;
                    if (index == 58) index = 57;
                    double fint = radliq - 1.5 - (double)index;
                    const double *a = LWCLD_ABSLIQ1[ib];
                    abscoliq = a[index - 1] + fint * (a[index] - (a[index - 1]));
                }
                taucmc[q] = ciwpmc * abscoice + clwpmc * abscoliq + cswpmc * abscosno;
            }
    return 0;
}
