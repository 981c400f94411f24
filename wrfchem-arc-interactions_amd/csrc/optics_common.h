// Aerosol optical averaging: shared host/device definitions.
//
// The reference repository carries only the two ends of this stage: the MOSAIC sectional tracers it reads
// (Registry/registry.chem:2325-2447, packages :3823-3826) and the arrays it must fill (registry.chem:1332-1387,
// zeroed in chem/chemics_init.F:486-553, handed to the radiation driver at
// dyn_em/module_first_rk_step_part1.F:374-385).  The arithmetic (upstream chem/module_optical_averaging.F) is not
// in the reference, so the physics below is this engine's own specification (DESIGN.md "Optical averaging").
#pragma once
#include <cstdint>

namespace rrtmg {

constexpr int OPT_NSPEC = 9;       // mass species: so4 no3 cl nh4 na oin oc bc water
constexpr int OPT_NSLOT = 10;      // + number
constexpr int OPT_MAXBIN = 8;
constexpr int OPT_NSW = 4;         // 300, 400, 600, 999 nm (registry.chem:1367-1378; SW:10993-11023)
constexpr int OPT_NLW = 16;        // RRTMG-LW bands (LW:78)
constexpr int OPT_NWAVE = OPT_NSW + OPT_NLW;
constexpr int OPT_NMIE = 256;      // stored logarithmic-derivative terms: size parameter capped at OPT_XMAX
constexpr double OPT_XMAX = 225.0;

enum OptSpec : int { OS_SO4 = 0, OS_NO3, OS_CL, OS_NH4, OS_NA, OS_OIN, OS_OC, OS_BC, OS_WATER, OS_NUM };

struct OpticsTables {
    double refr[OPT_NSPEC][OPT_NWAVE];
    double refi[OPT_NSPEC][OPT_NWAVE];
    double dens[OPT_NSPEC];            // g cm-3
    double rdens[OPT_NSPEC];           // 1 / dens (the kernels multiply; formed once on the host)
    double wavelength_um[OPT_NWAVE];
    double dlo_um[OPT_MAXBIN], dhi_um[OPT_MAXBIN];
    int nbin, pad_;
};

// RRTMG-LW band limits in cm-1 (LW:8030-8035 wavenum1/wavenum2); the optics are evaluated at the band centre
static const double OPT_LW_WAVENUM1[OPT_NLW] = {10., 350., 500., 630., 700., 820., 980., 1080.,
                                                1180., 1390., 1480., 1800., 2080., 2250., 2380., 2600.};
static const double OPT_LW_WAVENUM2[OPT_NLW] = {350., 500., 630., 700., 820., 980., 1080., 1180.,
                                                1390., 1480., 1800., 2080., 2250., 2380., 2600., 3250.};
static const double OPT_SW_WAVELENGTH_UM[OPT_NSW] = {0.300, 0.400, 0.600, 0.999};

// Built-in material table (a caller may replace it at init).  Values are rounded literature figures:
// liquid water after Hale & Querry (1973); mineral dust ("other inorganics") after the OPAC/Hess et al. (1998)
// mineral components; sulfate / nitrate / ammonium as ammonium sulfate, sea salt as NaCl, organic carbon and
// black carbon (Bond & Bergstrom 2006 for BC) taken wavelength-independent.
static const double OPT_DENSITY[OPT_NSPEC] = {1.8, 1.8, 2.2, 1.8, 2.2, 2.6, 1.0, 1.7, 1.0};
static const double OPT_WATER_REFR[OPT_NWAVE] = {1.349, 1.339, 1.332, 1.327,
                                                 1.650, 1.515, 1.410, 1.270, 1.160, 1.150, 1.232, 1.270,
                                                 1.297, 1.317, 1.320, 1.325, 1.330, 1.337, 1.350, 1.410};
static const double OPT_WATER_REFI[OPT_NWAVE] = {1.6e-8, 1.86e-9, 1.09e-8, 2.89e-6,
                                                 0.550, 0.370, 0.428, 0.402, 0.320, 0.105, 0.047, 0.037,
                                                 0.034, 0.0325, 0.125, 0.0105, 0.0147, 0.0087, 0.0047, 0.017};
static const double OPT_DUST_REFR[OPT_NWAVE] = {1.550, 1.550, 1.550, 1.550,
                                                2.340, 2.904, 1.748, 1.508, 1.911, 1.822, 2.917, 1.557,
                                                1.242, 1.447, 1.432, 1.473, 1.495, 1.500, 1.500, 1.510};
static const double OPT_DUST_REFI[OPT_NWAVE] = {0.006, 0.006, 0.006, 0.006,
                                                0.700, 0.857, 0.462, 0.263, 0.319, 0.260, 0.650, 0.373,
                                                0.093, 0.105, 0.061, 0.0245, 0.011, 0.008, 0.0068, 0.018};
// wavelength-independent species: real, imaginary
static const double OPT_CONST_REF[OPT_NSPEC][2] = {
    {1.52, 1.0e-9},   // so4
    {1.52, 1.0e-9},   // no3
    {1.51, 8.66e-7},  // cl   (sea salt)
    {1.52, 1.0e-9},   // nh4
    {1.51, 8.66e-7},  // na   (sea salt)
    {0., 0.},         // oin  -> OPT_DUST_*
    {1.45, 0.0},      // oc
    {1.85, 0.71},     // bc
    {0., 0.},         // water -> OPT_WATER_*
};

inline void optics_default_tables(OpticsTables &t, int nbin, double dlo_um, double dhi_um)
{
    for (int w = 0; w < OPT_NSW; ++w) t.wavelength_um[w] = OPT_SW_WAVELENGTH_UM[w];
    for (int b = 0; b < OPT_NLW; ++b) t.wavelength_um[OPT_NSW + b] = 1.0e4 / (0.5 * (OPT_LW_WAVENUM1[b] + OPT_LW_WAVENUM2[b]));
    for (int s = 0; s < OPT_NSPEC; ++s) {
        t.dens[s] = OPT_DENSITY[s];
        for (int w = 0; w < OPT_NWAVE; ++w) {
            double re = OPT_CONST_REF[s][0], im = OPT_CONST_REF[s][1];
            if (s == OS_OIN) { re = OPT_DUST_REFR[w]; im = OPT_DUST_REFI[w]; }
            if (s == OS_WATER) { re = OPT_WATER_REFR[w]; im = OPT_WATER_REFI[w]; }
            t.refr[s][w] = re; t.refi[s][w] = im;
        }
    }
    t.nbin = nbin; t.pad_ = 0;
    // sections equal in ln(diameter), as MOSAIC lays them out between dlo_sect(1) and dhi_sect(nbin)
    for (int b = 0; b < OPT_MAXBIN; ++b) { t.dlo_um[b] = 0.; t.dhi_um[b] = 0.; }
    double lo = dlo_um;
    for (int b = 0; b < nbin; ++b) {
        // dhi(b) = dlo * (dhi/dlo)^((b+1)/nbin), built by repeated multiplication-free evaluation
        double frac = (double)(b + 1) / (double)nbin;
        double hi = dlo_um * __builtin_pow(dhi_um / dlo_um, frac);
        if (b == nbin - 1) hi = dhi_um;
        t.dlo_um[b] = lo; t.dhi_um[b] = hi;
        lo = hi;
    }
}

struct OpticsKernelArgs {
    int ni_m, nk_m, nj_m;          // memory dimensions
    int i0, k0, j0;                // tile origin inside memory (0-based)
    int ni_t, nk_t, nj_t;          // tile dimensions
    int real4;
    int mix;                       // 0 volume mixing, 1 Maxwell-Garnett with BC inclusions, 2 BC core + shell
    int nbin;
    int idx[OPT_MAXBIN][OPT_NSLOT];   // 0-based species index into chem, -1 = absent
    const void *chem, *alt, *dz8w;
    void *tauaer[OPT_NSW], *waer[OPT_NSW], *gaer[OPT_NSW], *extaer[OPT_NSW], *bscoef[OPT_NSW];
    void *tauaerlw[OPT_NLW], *extaerlw[OPT_NLW];
    void *laer[6];                 // l2aer..l7aer, each [4 wavelengths][memory shape]; all or none
};

}  // namespace rrtmg
