/*
 * oracle/rrtmg_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, one column at a time, fp64) of the reference's
 * aerosol + clean-sky RRTMG double call, WRF-Chem v3.9.1 patch set:
 *   SW:  phys/module_ra_rrtmg_sw.F   (setcoef_sw, taumol_sw, reftra_sw,
 *        vrtqdr_sw, spcvmc_sw, rrtmg_sw, inatm_sw, rrtmg_sw_ini/cmbgbNN,
 *        RRTMG_SWRAD pre/post-processing)
 *   LW:  phys/module_ra_rrtmg_lw.F   (see rrtmg_lw_oracle.c)
 * Every function cites the reference lines it follows.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call this library.  The shipped HIP path never does.
 *
 * Pinning: the column solvers (rrtmg_sw / rrtmg_lw and everything below them,
 * including the g-point reduction at init) are pinned against the reference
 * Fortran itself, compiled from /root/reference by oracle/build_ref.sh into
 * oracle/_ref and run on synthetic tables + columns (tests/golden npz fixtures,
 * generator tools/make_golden.py).  The WRF-facing wrapper arithmetic
 * (RRTMG_SWRAD / RRTMG_LWRAD pre- and post-processing) cannot be compiled
 * here without stand-ins for absent WRF modules, so for those rows the
 * oracle is "parity unpinned" (restated from the source text only).
 */
#ifndef RRTMG_ORACLE_H
#define RRTMG_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define SW_NBND 14
#define SW_NGPT 112
#define SW_JPB1 16

typedef struct SwTables SwTables;

/* rrtmg_swlookuptable + rrtmg_sw_ini (SW:11236-11293, SW:4571-4698):
 * read a RRTMG_SW_DATA-layout file, reduce 224->112 g-points, build exp_tbl. */
SwTables *sw_oracle_load(const char *path, int real_bytes, double cpdair, char *err, int errlen);
void sw_oracle_free(SwTables *t);

/* Reduced table access for parity tests of the init (band = 16..29).
 * name: "absa","absb","selfref","forref","sfluxref","rayl","rayla","raylb",
 *       "abso3a","abso3b","absch4","absh2o","absco2","exp_tbl","rwgt".
 * Returns pointer to doubles in the reference's (Fortran) memory order and
 * the element count in *n; NULL if the band has no such table. */
const double *sw_oracle_table(const SwTables *t, int band, const char *name, int *n);

/* One column of rrtmg_sw (SW:8740-9490), cloud-free (icld handling: all
 * McICA cloud fractions zero), iaer=10, aerosol + clean double call.
 * Arrays are bottom->top; nlay = number of RRTMG layers (WRF K+1).
 *   play,tlay,h2ovmr,o3vmr,co2vmr,ch4vmr,n2ovmr,o2vmr : [nlay]
 *   plev,tlev : [nlay+1]
 *   tauaer,ssaaer,asmaer : [SW_NBND][nlay]  (band-major)
 * Outputs: flux arrays [nlay+1], heating [nlay] (K/day). */
typedef struct {
    int nlay;
    const double *play, *plev, *tlay, *tlev;
    double tsfc;
    const double *h2ovmr, *o3vmr, *co2vmr, *ch4vmr, *n2ovmr, *o2vmr;
    double asdir, asdif, aldir, aldif;
    double coszen, adjes, scon;
    int dyofyr;
    const double *tauaer, *ssaaer, *asmaer;
    int clean_atm_diag;
    /* clouds: cldfrac == NULL -> cloud-free column.  Otherwise the in-cloud state RRTMG_SWRAD hands to
     * mcica_subcol_sw (SW:10707-10957): per layer cloud fraction, in-cloud liquid / ice / snow water path (g m-2),
     * effective sizes (um), and inflgsw / iceflgsw / liqflgsw.  icld = 2, permuteseed = 1, irng = 0.
     * seed_real4: derive the KISS seeds in single precision (stock REAL*4 WRF build). */
    const double *cldfrac, *clwp, *ciwp, *cswp, *rel, *rei, *res;
    int inflgsw, iceflgsw, liqflgsw, seed_real4;
    double *cldfmc_out;      /* optional [SW_NGPT][nlay] dump of the sub-column mask */
} SwColumnIn;

/* WRF hydrometeor state of one column as the wrappers receive it (NULL array = argument absent / F_Q* false) */
typedef struct {
    int nk;
    const double *t, *pdel;                       /* layer temperature t3d (K), thickness (mb) */
    const double *t_rad;                          /* tlay as relcalc / reicalc see it (LW overwrites tlay(kte),
                                                     LW:12246-12250); NULL = t */
    const double *cldfra, *qc, *qr, *qi, *qs;     /* cloud fraction, mixing ratios (kg/kg) */
    const double *re_cloud, *re_ice, *re_snow;    /* effective radii (m), used with has_req* */
    const double *qndrop;                         /* droplet number (1/kg), used with progn = 1 */
    int has_reqc, has_reqi, has_reqs, warm_rain, progn;
    double xland, xice, snow, g;
    /* variants: "MP option 5" split of snow (SW:10461-10472), Ferrier-Aligo (SW:10474-10492), CAM-MG radii in um
     * (SW:10883-10898; both arrays or NULL) */
    int f_ice_phy, fer_hires;
    const double *iradius, *lradius;
} OracleCloudWrf;
int oracle_cloud_state(const OracleCloudWrf *w, double *cldfrac, double *clwp, double *ciwp, double *cswp,
                       double *rel, double *rei, double *res, int *flags3);

double oracle_relcalc(double t, double landm, double icefrac, double snowh);   /* LW:14403-14462 */
double oracle_reicalc(double t);                                               /* LW:14464-14491 */

/* McICA generator and cloud optics (rrtmg_sw_cloud_oracle.c) */
int oracle_mcica(int nsub, int nlay, const double *play, const double *cldfrac, int change_seed, int real4,
                 double *cldfmc);
int lw_oracle_cldprmc(int nlay, int inflag, int iceflag, int liqflag, const double *cldfmc, const double *ciwp,
                      const double *clwp, const double *cswp, const double *rei, const double *rel, const double *res,
                      double *taucmc, char *err, int errlen);
int sw_oracle_mcica(int nlay, const double *play, const double *cldfrac, int change_seed, int real4, double *cldfmc);
int sw_oracle_cldprmc(int nlay, int inflag, int iceflag, int liqflag, const double *cldfmc, const double *ciwp,
                      const double *clwp, const double *cswp, const double *rei, const double *rel, const double *res,
                      double *taucmc, double *ssacmc, double *asmcmc, double *taormc, char *err, int errlen);

typedef struct {
    double *swuflx, *swdflx, *swhr, *swuflxc, *swdflxc, *swhrc;
    double *swuflxcln, *swdflxcln;
    double *sibvisdir, *sibvisdif, *sibnirdir, *sibnirdif;
    double *swdkdir, *swdkdif;
} SwColumnOut;

int sw_oracle_rrtmg_sw(const SwTables *t, const SwColumnIn *in, SwColumnOut *out);

/* RRTMG_SWRAD over a tile (SW:9901-11207), cloud-free path (icloud = 0),
 * aer_ra_feedback = 1 with the WRF-Chem compact aerosol form, o3input != 2.
 * WRF memory order, fp64 (a -r8 build): 3-D (ims:ime,kms:kme,jms:jme),
 * flux profiles (ims:ime,kms:kme+2,jms:jme), 2-D (ims:ime,jms:jme).
 * Any output pointer may be NULL (== argument not PRESENT). */
typedef struct {
    int ims, ime, jms, jme, kms, kme;
    int its, ite, jts, jte, kts, kte;
    const double *xcoszen, *albedo, *tsk;                    /* 2-D */
    const double *t3d, *t8w, *p3d, *p8w, *pi3d, *qv3d;       /* 3-D */
    const double *tauaer300, *tauaer400, *tauaer600, *tauaer999;
    const double *gaer300, *gaer400, *gaer600, *gaer999;
    const double *waer300, *waer400, *waer600, *waer999;
    double solcon;
    int aer_ra_feedback, clean_atm_diag;
    /* clouds (icloud != 0; SW:10371-10618, 10707-10957).  NULL 3-D array = argument absent / F_Q* false. */
    int icloud;
    const double *cldfra3d, *qc3d, *qr3d, *qi3d, *qs3d, *re_cloud, *re_ice, *re_snow, *qndrop3d;   /* 3-D */
    const double *xland, *xice, *snow;                                                              /* 2-D */
    int has_reqc, has_reqi, has_reqs, warm_rain, progn;
    double g;
    int seed_real4;
    const double *tauaer3d_sw, *ssaaer3d_sw, *asyaer3d_sw;   /* (i,k,j,band), optional (SW:10982-10991) */
    const double *aerod;                                     /* (i,k,j,6), optional: aer_opt = 1 (SW:11083-11100) */
    const double *alswvisdir, *alswvisdif, *alswnirdir, *alswnirdif;   /* SSiB albedos (2-D), optional (SW:10692-10703) */
    int f_ice_phy, fer_hires;                                /* cloud variants, see OracleCloudWrf */
    const double *iradius, *lradius;                         /* 3-D, CAM-MG radii */
} SwTileIn;

/* iaer = 6: band optical properties of the mixture of the six ECMWF aerosol types, SW:9313-9340.
 * ecaer[type][layer] (optical depth at 0.55 um) -> tau, ssa, asy [band][layer]. */
void sw_oracle_ecaer_bands(int nlay, const double *ecaer, double *tau, double *ssa, double *asy);

typedef struct {
    double *rthratensw;                                      /* 3-D, INOUT */
    double *gsw, *swcf, *coszr;                              /* 2-D */
    double *swupt, *swuptc, *swuptcln, *swdnt, *swdntc, *swdntcln;
    double *swupb, *swupbc, *swupbcln, *swdnb, *swdnbc, *swdnbcln;
    double *swvisdir, *swvisdif, *swnirdir, *swnirdif;
    double *swddir, *swddni, *swddif;
    double *swupflx, *swupflxc, *swupflxcln, *swdnflx, *swdnflxc, *swdnflxcln; /* (i,kms:kme+2,j) */
} SwTileOut;

/* returns 0, or >0 with a message (the text wrf_error_fatal would get) */
int sw_oracle_swrad_tile(const SwTables *t, const SwTileIn *in, SwTileOut *out, char *err, int errlen);

/* O3 climatology: INIRAD/O3DATA (LW:12704-12840). plev[nlev=kte-kts+3] in mb
 * bottom->top, o3prof[kte-kts+2] out (mass mixing ratio). */
void oracle_inirad(double *o3prof, const double *plev, int nlay_model);

/* ------------------------------ longwave ------------------------------- */
#define LW_NBND 16
#define LW_NGPT 140

typedef struct LwTables LwTables;

/* rrtmg_lwlookuptable + rrtmg_lw_ini (LW:12876-12930, LW:7862-8009) */
LwTables *lw_oracle_load(const char *path, int real_bytes, double cpdair, char *err, int errlen);
void lw_oracle_free(LwTables *t);
/* name: absa, absb, selfref, forref, fracrefa, fracrefb, kam0..2, kbm0..1, gv0..1,
 *       exp_tbl, tau_tbl, tfn_tbl (band ignored) ; band = 1..16 */
const double *lw_oracle_table(const LwTables *t, int band, const char *name, int *n);

/* one cloud-free column of rrtmg_lw (LW:10578-11062): gas-only ("clean") call
 * and gas+aerosol call.  Arrays bottom->top; tauaer [LW_NBND][nlay].
 * Outputs: flux arrays [nlay+1], heating [nlay]. */
typedef struct {
    int nlay;
    const double *play, *plev, *tlay, *tlev;
    double tsfc;
    const double *h2ovmr, *o3vmr, *co2vmr, *ch4vmr, *n2ovmr, *o2vmr;
    const double *cfc11vmr, *cfc12vmr, *cfc22vmr, *ccl4vmr;
    const double *emis;      /* [LW_NBND] */
    const double *tauaer;
    int clean_atm_diag;
    /* clouds (cldfrac == NULL: cloud-free): in-cloud state on all nlay layers (zero above the model top), as
     * RRTMG_LWRAD hands it to mcica_subcol_lw (LW:12335-12563; icld = 2, permuteseed = 150, irng = 0) */
    const double *cldfrac, *clwp, *ciwp, *cswp, *rel, *rei, *res;
    int inflglw, iceflglw, liqflglw, seed_real4;
    double *cldfmc_out;      /* optional [LW_NGPT][nlay] */
} LwColumnIn;

typedef struct {
    double *uflx, *dflx, *hr, *uflxc, *dflxc, *hrc, *uflxcln, *dflxcln;
} LwColumnOut;

int lw_oracle_rrtmg_lw(const LwTables *t, const LwColumnIn *in, LwColumnOut *out);

/* NLAYERS of rrtmg_lwinit (LW:12861) */
int lw_oracle_nlayers(int kme, double p_top_pa);

/* RRTMG_LWRAD over a tile (LW:11451-12700), cloud-free path, fp64 WRF arrays */
typedef struct {
    int ims, ime, jms, jme, kms, kme;
    int its, ite, jts, jte, kts, kte;
    int nlayers;                                             /* module variable set by rrtmg_lwinit */
    const double *emiss, *tsk;                               /* 2-D */
    const double *t3d, *t8w, *p3d, *p8w, *pi3d, *qv3d;       /* 3-D */
    const double *tauaerlw[LW_NBND];                         /* tauaerlw1..16 */
    int aer_ra_feedback, clean_atm_diag;
    /* clouds (icloud != 0; LW:11921-12140, 12335-12563), as in SwTileIn */
    int icloud;
    const double *cldfra3d, *qc3d, *qr3d, *qi3d, *qs3d, *re_cloud, *re_ice, *re_snow, *qndrop3d;   /* 3-D */
    const double *xland, *xice, *snow;                                                              /* 2-D */
    int has_reqc, has_reqi, has_reqs, warm_rain, progn;
    double g;
    int seed_real4;
    int f_ice_phy, fer_hires;                                /* cloud variants, see OracleCloudWrf */
    const double *iradius, *lradius;                         /* 3-D, CAM-MG radii */
} LwTileIn;

typedef struct {
    double *rthratenlw;                                      /* 3-D */
    double *glw, *olr, *lwcf;                                /* 2-D */
    double *lwupt, *lwuptc, *lwuptcln, *lwdnt, *lwdntc, *lwdntcln;
    double *lwupb, *lwupbc, *lwupbcln, *lwdnb, *lwdnbc, *lwdnbcln;
    double *lwupflx, *lwupflxc, *lwupflxcln, *lwdnflx, *lwdnflxc, *lwdnflxcln;   /* (i,kms:kme+2,j) */
} LwTileOut;

int lw_oracle_lwrad_tile(const LwTables *t, const LwTileIn *in, LwTileOut *out, char *err, int errlen);

#ifdef __cplusplus
}
#endif
#endif
