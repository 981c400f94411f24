/*
 * include/rrtmg_hip.h -- C ABI of the MI355X column-radiation engine.
 *
 * Drop-in boundary for WRF-Chem v3.9.1's patched RRTMG double call
 * (douglowe/WRFChem-ARC-Interactions).  Each entry point replaces one Fortran
 * module procedure of the reference; the thin ISO_C_BINDING shims in
 * wrfchem-arc-interactions_amd/fortran/ keep the reference's procedure names
 * and argument lists and forward to these symbols.
 *
 *   rrtmg_hip_sw_init   <- rrtmg_swinit + rrtmg_swlookuptable + rrtmg_sw_ini
 *                          phys/module_ra_rrtmg_sw.F:11211-11293, 4571-4698
 *   rrtmg_hip_swrad     <- RRTMG_SWRAD           module_ra_rrtmg_sw.F:9901-11207
 *                          (call site phys/module_radiation_driver.F:1944-2001)
 *   rrtmg_hip_lw_init   <- rrtmg_lwinit + rrtmg_lwlookuptable + rrtmg_lw_ini
 *                          phys/module_ra_rrtmg_lw.F:12845-12930, 7862-8009
 *   rrtmg_hip_lwrad     <- RRTMG_LWRAD           module_ra_rrtmg_lw.F:11451-12700
 *                          (call site phys/module_radiation_driver.F:1528-1577)
 *   rrtmg_hip_finalize  (no reference counterpart; frees device tables)
 *
 * Conventions
 *  - plain pointers and sizes only; arrays stay in WRF memory order:
 *      3-D  (ims:ime, kms:kme, jms:jme), i fastest      (SW:9959)
 *      2-D  (ims:ime, jms:jme)
 *      flux profiles (ims:ime, kms:kme+2, jms:jme)      (SW:10099)
 *  - real_bytes selects the width of every REAL array (8 for a -r8 build,
 *    4 for stock WRF); all arithmetic is fp64 on the device either way.
 *  - mem = 0: pointers are host memory and the call is synchronous.  Only the
 *    tile its:ite x jts:jte of every array is read, staged (packed into the
 *    calling thread's pinned buffer, one copy each way) and written back, so
 *    concurrent calls on disjoint tiles of SHARED arrays do not disturb each
 *    other, and each call has its own status word: a fatal is delivered to the
 *    call that raised it.  mem = 1: pointers are device (HBM) addresses, work is
 *    enqueued on `stream` (a hipStream_t, NULL = default stream) and the call
 *    returns without synchronising.
 *  - an OPTIONAL Fortran argument that is not PRESENT is passed as NULL.
 *  - return value 0 = ok; nonzero = fatal, with the text the reference would
 *    hand to wrf_error_fatal in `err` (the Fortran shim calls wrf_error_fatal).
 *  - re-entrant after init: module tables are read-only; each call uses only
 *    its own stream and workspace (DRV:975-982 calls this from an OpenMP tile
 *    loop).
 */
#ifndef RRTMG_HIP_H
#define RRTMG_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define RRTMG_HIP_ERRLEN 256

/* Read RRTMG_SW_DATA (Fortran unformatted, one record per band, field order
 * of sw_kgb16..29, SW:11379-12366), reduce 224 -> 112 g-points (cmbgb16s..29),
 * build the exponential lookup table and upload everything to `device`.
 * data_real_bytes: word size of the REALs in the file (4: RRTMG_SW_DATA,
 * 8: RRTMG_SW_DATA_DBL).  cpdair: WRF's cp (module_model_constants). */
int rrtmg_hip_sw_init(const char *data_path, int data_real_bytes, double cpdair, int device, char *err);

typedef struct rrtmg_hip_swrad_args {
    /* dimensions (SW:9937-9939) */
    int ims, ime, jms, jme, kms, kme;
    int its, ite, jts, jte, kts, kte;
    int real_bytes;          /* 4 or 8 */
    int mem;                 /* 0 host pointers, 1 device pointers */
    void *stream;            /* hipStream_t, used when mem = 1 */
    /* scalars / flags */
    double solcon;           /* SOLCON */
    int aer_ra_feedback;     /* 0/1 */
    int clean_atm_diag;      /* <=0: CLN outputs are 0 (SW:9473-9478) */
    int icloud;              /* 0: cloud-free; otherwise McICA clouds from the cloud arguments below */
    int o3input;             /* 2 = use o33d inside the model domain */
    /* inputs, 2-D */
    const void *xcoszen, *albedo, *tsk;
    /* inputs, 3-D */
    const void *t3d, *t8w, *p3d, *p8w, *pi3d, *qv3d;
    const void *o33d;                                        /* optional */
    const void *tauaer300, *tauaer400, *tauaer600, *tauaer999;   /* optional unless aer_ra_feedback */
    const void *gaer300, *gaer400, *gaer600, *gaer999;
    const void *waer300, *waer400, *waer600, *waer999;
    /* outputs, 3-D */
    void *rthratensw;                                        /* INOUT: written for daylit columns only */
    /* outputs, 2-D */
    void *coszr, *gsw, *swcf;
    void *swupt, *swuptc, *swuptcln, *swdnt, *swdntc, *swdntcln;    /* optional as a group */
    void *swupb, *swupbc, *swupbcln, *swdnb, *swdnbc, *swdnbcln;
    void *swvisdir, *swvisdif, *swnirdir, *swnirdif;                /* optional */
    void *swddir, *swddni, *swddif;
    /* optional flux profiles (ims:ime, kms:kme+2, jms:jme) */
    void *swupflx, *swupflxc, *swupflxcln, *swdnflx, *swdnflxc, *swdnflxcln;
    /* clouds, used when icloud != 0 (RRTMG_SWRAD arguments of the same names, SW:9901-9945; McICA with
     * maximum-random overlap, KISS seeded from the fractional Pa of the four lowest layers, SW:1696-1699).
     * A hydrometeor array is NULL when the argument is not PRESENT or its F_Q* flag is false. */
    const void *cldfra3d, *qc3d, *qr3d, *qi3d, *qs3d;      /* 3-D */
    const void *re_cloud, *re_ice, *re_snow;               /* 3-D, metres; used with has_reqc / has_reqi / has_reqs */
    const void *qndrop3d;                                  /* 3-D, used with progn = 1 (F_QNDROP) */
    const void *xland, *xice, *snow;                       /* 2-D */
    int has_reqc, has_reqi, has_reqs;
    int warm_rain, progn;
    int is_cammgmp_used;                                   /* radii from lradius / iradius (below) */
    int f_ice_phy;                                         /* 1: F_ICE_PHY present with F_QC, F_QS and no F_QI ("MP option 5") */
    int mp_physics;                                        /* FER_MP_HIRES (5) / _ADVECT (15): qi3d = all frozen water */
    double g;                                              /* gravity the driver passes (G), m s-2 */
    /* band-resolved aerosol input of aer_opt = 2 / 3 (Fortran pointers tauaer3d_sw, ssaaer3d_sw, asyaer3d_sw,
     * (ims:ime, kms:kme, jms:jme, 1:14), SW:10982-10991): NULL = not associated.  Where the WRF-Chem arrays
     * qualify (aer_ra_feedback = 1, tauaer300 and tauaer999 > 1e-9) they override these, as in the reference. */
    const void *tauaer3d_sw, *ssaaer3d_sw, *asyaer3d_sw;
    /* aer_opt = 1 (SW:9201-9205, 11083-11100): optical depth at 0.55 um of the six ECMWF aerosol types,
     * aerod(ims:ime, kms:kme, jms:jme, 1:6); rrtmg_sw then runs with iaer = 6 and every other aerosol input is
     * ignored (SW:9313-9340).  NULL = aer_opt /= 1.  The clean call uses zero aerosol optical depth (the
     * reference leaves ztauacln unset on this route). */
    const void *aerod;
    /* SSiB land surface (sf_surface_physics = 8, SW:10692-10703): spectral albedos (2-D) used where xland < 1.5;
     * the four together or none.  NULL = the broadband albedo everywhere.  xland is then read whatever icloud. */
    const void *alswvisdir, *alswvisdif, *alswnirdir, *alswnirdif;
    /* is_cammgmp_used: effective radii from the CAM-MG microphysics, lradius / iradius(ims:ime,kms:kme,jms:jme) in
     * um (SW:10883-10898); required when is_cammgmp_used /= 0. */
    const void *lradius, *iradius;
} rrtmg_hip_swrad_args;

/* Returns 0, or non-zero with a message in err[RRTMG_HIP_ERRLEN].  An empty tile (ite < its or jte < jts) returns 0
 * without touching any array, like the tile loops of RRTMG_SWRAD; the same holds for rrtmg_hip_lwrad and
 * rrtmg_hip_optical_averaging.  Host threads may call concurrently (one call's kernels are enqueued at a time). */
int rrtmg_hip_swrad(const rrtmg_hip_swrad_args *args, char *err);

/* mem = 1 calls do not synchronise; this does (hipDeviceSynchronize), and
 * returns nonzero with the reference's message if any of them raised the
 * "Negative total optical depth" fatal (SW:11031-11033) on the device. */
int rrtmg_hip_sw_status(char *err);

/* Host-only helpers used by the CPU tests of the init path (no device). */
int rrtmg_hip_sw_reduce_tables(const char *data_path, int data_real_bytes, double cpdair, double *k_out,
                               long long k_cap, long long *k_len, double *exp_out, char *err);
int rrtmg_hip_sw_table_info(const char *data_path, int data_real_bytes, int g, int *fields16, double *dfields10,
                            char *err);

/* bytes of device workspace a mem=0 call stages (diagnostic) */
long long rrtmg_hip_swrad_staging_bytes(const rrtmg_hip_swrad_args *args);

/* Per-tile sums used for the multi-GPU gather: sums over the
 * tile of the 12 SW TOA/surface fluxes + the clean-minus-aerosol TOA
 * difference + daylit-column count, written to out[16] (device memory when
 * mem = 1).  Field order: swupt,swuptc,swuptcln,swdnt,swdntc,swdntcln,
 * swupb,swupbc,swupbcln,swdnb,swdnbc,swdnbcln, (swuptcln-swupt), ncol_day,
 * ncol, 0. */
int rrtmg_hip_sw_tile_sums(const rrtmg_hip_swrad_args *args, double *out16, char *err);
/* same for any 12 two-dimensional device arrays (order *upt,*uptc,*uptcln,*dnt,
 * *dntc,*dntcln,*upb,*upbc,*upbcln,*dnb,*dnbc,*dnbcln); out16 is device memory;
 * enqueued on `stream`, deterministic summation order. */
int rrtmg_hip_tile_sums(const void *const *fields12, int real_bytes, int ims, int ime, int jms, int jme,
                        int its, int ite, int jts, int jte, double *out16_dev, void *stream, char *err);

/* Domain statistics of one 2-D device field over the tile (optionally masked, mask2d nonzero = use; memory shape):
 * what calc_standard_stats of the reference's post-processing forms on the host
 * (analysis_scripts/NCL_extraction_package/misc_stats_library.ncl:405-458).  out13 (device):
 * avg, stddev (N-1), min, max, median, 25th, 75th, 5th, 95th percentile, standard error, Moran's I (neighbour /
 * Manhattan weights, :196-370), corrected SE = SE * I, N.  Percentile p = sorted[round(p/100 (N-1))].  Synchronises `stream`. */
int rrtmg_hip_domain_stats(const void *field2d, const unsigned char *mask2d, int real_bytes, int ims, int ime, int jms,
                           int jme, int its, int ite, int jts, int jte, double *out13_dev, void *stream, char *err);

/* ---------------------- driver-side neighbours ---------------------- */

/* Tile of two-dimensional (i,j) arrays in WRF memory order; mem / stream as in rrtmg_hip_swrad_args. */
typedef struct {
    int ims, ime, jms, jme, its, ite, jts, jte;
    int real_bytes;          /* 4 or 8: REAL size of every array of the call */
    int mem;                 /* 0 host pointers (staged, synchronous), 1 device pointers (enqueued on stream) */
    void *stream;
} rrtmg_hip_tile2d;

/* Energy buckets, module_radiation_driver.F:2308-2381: acc[f] += flux[f] * dtaccum for up to 16 field pairs
 * (the driver's order: LW upt,uptc,dnt,dntc,upb,upbc,dnb,dnbc then the same for SW; NULL pairs are skipped). */
int rrtmg_hip_accumulate_fluxes(const rrtmg_hip_tile2d *dims, double dtaccum, const void *const *flux16,
                                void *const *acc16, char *err);

/* swint_opt = 1, after a radiation step: update_swinterp_parameters, module_radiation_driver.F:2668-2749. */
typedef struct {
    const void *coszen, *coszen_loc, *swddir, *swdown;
    void *swddir_ref, *bb, *Bx, *swdown_ref, *gg, *Gx, *coszen_ref;
} rrtmg_hip_swinterp_update_args;
int rrtmg_hip_update_swinterp_parameters(const rrtmg_hip_tile2d *dims, const rrtmg_hip_swinterp_update_args *args,
                                         char *err);

/* swint_opt = 1, every model step: interp_sw_radiation, module_radiation_driver.F:2751-2795. */
typedef struct {
    const void *coszen_ref, *coszen_loc, *swddir_ref, *bb, *Bx, *swdown_ref, *gg, *Gx, *albedo;
    void *swdown, *swddir, *swddni, *swddif, *gsw;
} rrtmg_hip_swinterp_apply_args;
int rrtmg_hip_interp_sw_radiation(const rrtmg_hip_tile2d *dims, const rrtmg_hip_swinterp_apply_args *args, char *err);

/* ------------------------------- longwave -------------------------------- */

/* Read RRTMG_LW_DATA (record field order of lw_kgb01..16, LW:13021-14387),
 * reduce 256 -> 140 g-points (cmbgb1..16), build the exp / tfn lookup tables,
 * upload.  p_top_pa and kme fix NLAYERS = kme + nint(p_top*0.01/4) - 1, the
 * model layers plus the 4-mb buffer layers up to 0 mb (LW:12861). */
int rrtmg_hip_lw_init(const char *data_path, int data_real_bytes, double cpdair, double p_top_pa, int kme,
                      int device, char *err);
int rrtmg_hip_lw_nlayers(int kme, double p_top_pa);
/* the same with the quotient evaluated in the caller's REAL (real_bytes 4: single precision, as a stock WRF build
 * does, LW:12861); rrtmg_hip_lw_init uses data_real_bytes this way */
int rrtmg_hip_lw_nlayers_real(int kme, double p_top_pa, int real_bytes);

typedef struct rrtmg_hip_lwrad_args {
    int ims, ime, jms, jme, kms, kme;
    int its, ite, jts, jte, kts, kte;
    int real_bytes;
    int mem;
    void *stream;
    int aer_ra_feedback, clean_atm_diag, icloud, o3input;
    /* inputs, 2-D */
    const void *emiss, *tsk;
    /* inputs, 3-D */
    const void *t3d, *t8w, *p3d, *p8w, *pi3d, *qv3d;
    const void *o33d;                    /* optional */
    const void *tauaerlw[16];            /* tauaerlw1..16, optional unless aer_ra_feedback */
    /* outputs */
    void *rthratenlw;                    /* 3-D */
    void *glw, *olr, *lwcf;              /* 2-D */
    void *lwupt, *lwuptc, *lwuptcln, *lwdnt, *lwdntc, *lwdntcln;   /* optional as a group */
    void *lwupb, *lwupbc, *lwupbcln, *lwdnb, *lwdnbc, *lwdnbcln;
    void *lwupflx, *lwupflxc, *lwupflxcln, *lwdnflx, *lwdnflxc, *lwdnflxcln;   /* optional, (i,kms:kme+2,j) */
    /* clouds, used when icloud != 0 (RRTMG_LWRAD arguments of the same names, LW:11451-11486): same meaning as in
     * rrtmg_hip_swrad_args; the longwave McICA stream uses permuteseed 150 (LW:12555) */
    const void *cldfra3d, *qc3d, *qr3d, *qi3d, *qs3d;
    const void *re_cloud, *re_ice, *re_snow;
    const void *qndrop3d;
    const void *xland, *xice, *snow;
    int has_reqc, has_reqi, has_reqs;
    int warm_rain, progn;
    int is_cammgmp_used, f_ice_phy, mp_physics;            /* as in rrtmg_hip_swrad_args */
    double g;
    const void *lradius, *iradius;                         /* CAM-MG effective radii (um), LW:12473-12488 */
} rrtmg_hip_lwrad_args;

int rrtmg_hip_lwrad(const rrtmg_hip_lwrad_args *args, char *err);
int rrtmg_hip_lw_status(char *err);
int rrtmg_hip_lw_finalize(void);
int rrtmg_hip_lw_tile_sums(const rrtmg_hip_lwrad_args *args, double *out16, char *err);
/* host-only helpers for the CPU tests of the init path */
int rrtmg_hip_lw_reduce_tables(const char *data_path, int data_real_bytes, double cpdair, double *k_out,
                               long long k_cap, long long *k_len, double *exp_out, double *tfn_out, char *err);
int rrtmg_hip_lw_table_info(const char *data_path, int data_real_bytes, int g, int *fields24, double *dfields8,
                            char *err);

/* ------------------------- aerosol optical averaging ------------------------ */
/*
 * Produces the arrays the radiation path consumes (S2 / L2): tauaer1..4, waer1..4, gaer1..4 at 300/400/600/999 nm
 * (registry.chem:1367-1378, handed over at module_first_rk_step_part1.F:374-385, consumed at
 * module_ra_rrtmg_sw.F:10993-11023) and tauaerlw1..16 per RRTMG-LW band (registry.chem:1349-1364, consumed at
 * module_ra_rrtmg_lw.F:12582-12604) from MOSAIC sectional aerosol (registry.chem:2325-2447, packages
 * registry.chem:3823-3826).  Upstream WRF-Chem does this in chem/module_optical_averaging.F, which is NOT part of
 * the reference repository: there is no reference arithmetic to match, the physics is specified in DESIGN.md
 * (volume-mixed or Maxwell-Garnett refractive index per bin, homogeneous-sphere Mie theory evaluated per grid
 * cell, bin and wavelength) and parity is "unpinned".
 *
 * chem is WRF-Chem's 4-D tracer array chem(ims:ime, kms:kme, jms:jme, num_chem), i fastest; idx[b][s] is the
 * 1-based index (as in module_state_description's p_so4_a01 ...) of species s of bin b, 0 = not carried.
 */
enum {
    RRTMG_HIP_AER_SO4 = 0, RRTMG_HIP_AER_NO3, RRTMG_HIP_AER_CL, RRTMG_HIP_AER_NH4, RRTMG_HIP_AER_NA,
    RRTMG_HIP_AER_OIN, RRTMG_HIP_AER_OC, RRTMG_HIP_AER_BC, RRTMG_HIP_AER_WATER, RRTMG_HIP_AER_NUM,
    RRTMG_HIP_AER_NSPEC
};
#define RRTMG_HIP_MAXBIN 8

/* nbin sections between dlo_um and dhi_um (MOSAIC: 0.0390625 and 10.0 um), equal in log diameter.
 * refindex: NULL = built-in table, else [9 mass species][20 wavelengths][2] (real, imaginary >= 0) for the 4 SW
 * wavelengths followed by the 16 LW band centres; density: NULL = built-in, else [9] in g cm-3. */
int rrtmg_hip_optics_init(int nbin, double dlo_um, double dhi_um, const double *refindex, const double *density,
                          int device, char *err);

typedef struct rrtmg_hip_optics_args {
    int ims, ime, jms, jme, kms, kme;
    int its, ite, jts, jte, kts, kte;
    int real_bytes;
    int mem;
    void *stream;
    int aer_op_opt;           /* registry.chem:4014-4018: 1/3 volume mixing, 2/4 Maxwell-Garnett (BC inclusions);
                                 5 shell_exact (black carbon as a concentric core, coated-sphere Mie theory);
                                 every option evaluates Mie theory per cell */
    int num_chem;
    int idx[RRTMG_HIP_MAXBIN][RRTMG_HIP_AER_NSPEC];
    const void *chem;         /* (ims:ime, kms:kme, jms:jme, num_chem), ug/kg-dryair, number #/kg-dryair */
    const void *alt;          /* inverse dry-air density, m3 kg-1, 3-D */
    const void *dz8w;         /* layer thickness, m, 3-D */
    /* outputs, 3-D; each group may be NULL as a whole */
    void *tauaer[4], *waer[4], *gaer[4];        /* layer optical depth, single-scattering albedo, asymmetry */
    void *extaer[4], *bscoef[4];                /* extinction 1/km, backscatter 1/(km sr) */
    void *tauaerlw[16];                         /* layer absorption optical depth per LW band */
    void *extaerlw[16];                         /* absorption coefficient 1/km */
    /* cloud-borne aerosol (the "_cw" tracers of the aqueous packages, registry.chem:3935-3936): 1-based index of the
     * cloud-borne counterpart of species s of bin b, 0 = none (the water slot is ignored).  Where given, cloud-borne
     * mass and number are added to the interstitial ones before the particle geometry is formed.  All zero = the
     * interstitial aerosol alone. */
    int idx_cw[RRTMG_HIP_MAXBIN][RRTMG_HIP_AER_NSPEC];
    /* l2aer..l7aer (registry.chem:1382-1387), each (ims:ime, kms:kme, jms:jme, 4): Legendre moments chi_2..chi_7 of the
     * phase function at the four shortwave wavelengths, p(mu) = sum_k (2k+1) chi_k P_k(mu) with chi_0 = 1 and chi_1 =
     * gaer; scattering-weighted mean over the sections.  All six or none (NULL).  A diagnostic for schemes outside
     * this path (the radiation calls do not read it): evaluated by an extra kernel only when passed. */
    void *laer[6];
} rrtmg_hip_optics_args;

int rrtmg_hip_optical_averaging(const rrtmg_hip_optics_args *args, char *err);
/* Arithmetic of the stage: 0 = fp64 throughout (default); 1 = "mixed" (BASELINE configs[4]): the recurrences of the
 * Mie series (psi_n, the upward D_n) and the per-particle set-up in fp64, the coefficients a_n, b_n and the sums in
 * packed fp32 -- optical depths within 2e-5, omega and g within 1e-5, longwave absorption within 1e-4 of mode 0 at
 * about three quarters of its time.  The radiation calls are fp64 in both modes.  Takes effect from the next
 * rrtmg_hip_optical_averaging call; an unknown mode returns non-zero with a message in err. */
int rrtmg_hip_optics_precision(int mode, char *err);
int rrtmg_hip_optics_finalize(void);

/* Modal aerosol (MADE/SORGAM type: log-normal modes with fixed geometric standard deviations) -> the sectional
 * form above, so that rrtmg_hip_optical_averaging serves both tracer families (north_star: "MOSAIC/MADE size
 * bins").  Per mode: number (#/kg) and species masses (ug/kg, slots as in RRTMG_HIP_AER_*; idx[m][s] 1-based into
 * chem, 0 = not carried).  The mode's number-median diameter follows from number and dry volume; each section of
 * rrtmg_hip_optics_init receives the log-normal's share of number and (volume-weighted) of every species mass, the
 * end sections taking the tails (number and mass are conserved).  chem_sec(ims:ime, kms:kme, jms:jme, nbin*10):
 * species s of section b at 1-based index b*10 + s + 1 = the table rrtmg_hip_modal_section_idx fills for
 * rrtmg_hip_optics_args.idx (with num_chem = nbin*10).  The specification is the engine's own. */
#define RRTMG_HIP_MAXMODE 4
typedef struct rrtmg_hip_modal_args {
    int ims, ime, jms, jme, kms, kme;
    int its, ite, jts, jte, kts, kte;
    int real_bytes;
    int mem;
    void *stream;
    int nmode, num_chem;
    double sigma[RRTMG_HIP_MAXMODE];                          /* MADE: 1.7 (Aitken), 2.0 (accumulation), 2.2 (coarse) */
    int idx[RRTMG_HIP_MAXMODE][RRTMG_HIP_AER_NSPEC];
    const void *chem;                                         /* (ims:ime, kms:kme, jms:jme, num_chem) */
    void *chem_sec;                                           /* (ims:ime, kms:kme, jms:jme, nbin*10), output */
} rrtmg_hip_modal_args;
int rrtmg_hip_modal_to_sections(const rrtmg_hip_modal_args *args, char *err);
int rrtmg_hip_modal_section_idx(int nbin, int *idx /* [RRTMG_HIP_MAXBIN][RRTMG_HIP_AER_NSPEC] */);
/* host-only: wavelengths (um) of the 20 spectral points and the table the engine would use */
int rrtmg_hip_optics_table_info(double *wavelength_um20, double *refindex_9x20x2, double *density9);

/* Device for this process: RRTMG_HIP_DEVICE, else the MPI/SLURM local rank
 * (OMPI_COMM_WORLD_LOCAL_RANK, MV2_COMM_WORLD_LOCAL_RANK, MPI_LOCALRANKID,
 * SLURM_LOCALID, LOCAL_RANK) modulo the visible device count, else 0. */
int rrtmg_hip_pick_device(void);

/* Opt-in RRTMG_HIP_REGISTER_HOST=1 (mem = 0 calls): the caller's arrays are pinned in place (hipHostRegister) and
 * remembered by base address until the finalize calls.  A host that frees such an array earlier tells the engine
 * first; unknown addresses are ignored.  No reference counterpart (WRF's grid fields live as long as the run). */
int rrtmg_hip_unregister_host(const void *base);

int rrtmg_hip_finalize(void);

/* version / build info: "rrtmg_hip <ver> gfx950" */
const char *rrtmg_hip_version(void);

#ifdef __cplusplus
}
#endif
#endif
