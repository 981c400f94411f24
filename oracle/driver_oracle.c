/* oracle/driver_oracle.c -- TEST INFRASTRUCTURE (checker only; the product never calls it).
 *
 * CPU restatement of the driver-side neighbours of the radiation call, fp64:
 *   driver_oracle_accumulate       module_radiation_driver.F:2308-2381 (inline loops of the driver)
 *   driver_oracle_swinterp_update  update_swinterp_parameters, module_radiation_driver.F:2668-2749
 *   driver_oracle_swinterp_apply   interp_sw_radiation,        module_radiation_driver.F:2751-2795
 * Pinning: the two swint_opt routines are checked against the reference routines themselves, compiled from
 * /root/reference by oracle/build_ref.sh (-fdefault-real-8) and run by tools/make_golden_swinterp.py
 * (tests/golden/swinterp_*.npz, tests/test_driver_steps_cpu.py).  The accumulation is three lines of the driver
 * body, not a routine: restated only.
 * r4 != 0 mirrors the engine's handling of REAL*4 storage (thresholds and stored exponents rounded to float).
 */
#include <math.h>
#include <stddef.h>

static double rnd(double v, int r4) { return r4 ? (double)(float)v : v; }

void driver_oracle_accumulate(int n, double dt, const double *flux, double *acc)
{
    for (int i = 0; i < n; ++i) acc[i] = acc[i] + flux[i] * dt;   /* DRV:2330-2337, 2365-2372 */
}

/* DRV:2690-2707 (direct) and DRV:2714-2727 (global): exponent e and scale X of flux = X coszen**e */
static void fit(double flux, double flux_ref, double cz, double cz_loc, double cz_ref, double X_old, int r4, double *e,
                double *X)
{
    double f0, c0, ee;
    if (X_old <= 0.) { f0 = (cz_loc / cz) * flux; c0 = cz_loc; }
    else { f0 = flux_ref; c0 = cz_ref; }
    const double ratio = cz / c0;
    const double num = log(fmax(1., flux) / fmax(1., f0));
    if (ratio < 1.) ee = num / log(fmin(rnd(1. - 1e-4, r4), ratio));
    else if (ratio > 1.) ee = num / log(fmax(rnd(1. + 1e-4, r4), ratio));
    else ee = 0.;
    ee = rnd(fmax(-.5, fmin(2.5, ee)), r4);
    *e = ee;
    *X = flux / pow(cz, ee);
}

void driver_oracle_swinterp_update(int n, int r4, const double *coszen, const double *coszen_loc, const double *swddir,
                                   const double *swdown, double *swddir_ref, double *bb, double *Bx, double *swdown_ref,
                                   double *gg, double *Gx, double *coszen_ref)
{
    const double coszen_min = r4 ? (double)1e-4f : 1e-4;
    for (int i = 0; i < n; ++i) {
        if (coszen[i] > coszen_min && coszen_loc[i] > coszen_min) {
            fit(swddir[i], swddir_ref[i], coszen[i], coszen_loc[i], coszen_ref[i], Bx[i], r4, &bb[i], &Bx[i]);
            fit(swdown[i], swdown_ref[i], coszen[i], coszen_loc[i], coszen_ref[i], Gx[i], r4, &gg[i], &Gx[i]);
        } else {
            Bx[i] = 0.; bb[i] = 0.; Gx[i] = 0.; gg[i] = 0.;
        }
        coszen_ref[i] = coszen[i];      /* DRV:2734-2737 */
        swdown_ref[i] = swdown[i];
        swddir_ref[i] = swddir[i];
    }
}

void driver_oracle_swinterp_apply(int n, int r4, const double *coszen_ref, const double *coszen_loc,
                                  const double *swddir_ref, const double *bb, const double *Bx, const double *swdown_ref,
                                  const double *gg, const double *Gx, const double *albedo, double *swdown, double *swddir,
                                  double *swddni, double *swddif, double *gsw)
{
    const double coszen_min = r4 ? (double)1e-4f : 1e-4;
    for (int i = 0; i < n; ++i) {
        if (coszen_ref[i] > coszen_min && coszen_loc[i] > coszen_min) {   /* DRV:2772-2786 */
            double dir, glo;
            if (bb[i] == -0.5 || bb[i] == 2.5 || bb[i] == 0.0) dir = (coszen_loc[i] / coszen_ref[i]) * swddir_ref[i];
            else dir = Bx[i] * pow(coszen_loc[i], bb[i]);
            if (gg[i] == -0.5 || gg[i] == 2.5 || gg[i] == 0.0) glo = (coszen_loc[i] / coszen_ref[i]) * swdown_ref[i];
            else glo = Gx[i] * pow(coszen_loc[i], gg[i]);
            dir = rnd(dir, r4); glo = rnd(glo, r4);
            swddir[i] = dir; swdown[i] = glo;
            swddif[i] = glo - dir;
            swddni[i] = dir / coszen_loc[i];
            gsw[i] = glo * (1. - albedo[i]);
        } else {
            swddir[i] = 0.; swdown[i] = 0.; swddif[i] = 0.; swddni[i] = 0.; gsw[i] = 0.;
        }
    }
}
