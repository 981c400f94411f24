// Shared host/device definitions of the MI355X RRTMG engine.
//
// Spectral structure (reference: phys/module_ra_rrtmg_sw.F):
//   14 shortwave bands jb = 16..29, 112 reduced g-points; per-band g counts
//   ngc (SW:4812) and the role of each band's key species (taumol16..29,
//   SW:3293-4540) are encoded once, on the host, into a per-g-point descriptor
//   so that every lane of a wavefront runs the same instruction stream.
#pragma once
#include <cstdint>

namespace rrtmg {

constexpr int SW_NBND = 14;
constexpr int SW_NGPT = 112;
constexpr int SW_JPB1 = 16;
constexpr int MG = 16;
constexpr int NTBL = 10000;

// species slots of the per-layer column-amount array col[slot][k]
enum Species : int { SP_H2O = 0, SP_CO2 = 1, SP_O3 = 2, SP_CH4 = 3, SP_O2 = 4, SP_ZERO = 5, SP_COUNT = 6 };

// cfg bit layout of one regime (lower / upper atmosphere) of a g-point
constexpr uint32_t CFG_SPA_SHIFT = 0;    // 3 bits: species A (major, numerator of specparm)
constexpr uint32_t CFG_SPB_SHIFT = 3;    // 3 bits: species B of a binary band
constexpr uint32_t CFG_SPM_SHIFT = 6;    // 3 bits: minor species (x minorabs)
constexpr uint32_t CFG_NSP_SHIFT = 9;    // 4 bits: reference-atmosphere count (T stride), 1/5/9
constexpr uint32_t CFG_BINARY = 1u << 13;   // binary-species interpolation
constexpr uint32_t CFG_MULT4 = 1u << 14;    // specmult = 4*specparm (upper) instead of 8
constexpr uint32_t CFG_SELF = 1u << 15;     // water-vapour self continuum
constexpr uint32_t CFG_FOR = 1u << 16;      // foreign continuum
constexpr uint32_t CFG_O2CONT = 1u << 17;   // band 22 O2 continuum
constexpr uint32_t CFG_RAYL24 = 1u << 18;   // band 24 lower: Rayleigh interpolated in js
constexpr uint32_t CFG_MAJOR = 1u << 19;    // has a k-table in this regime

// solar-source selection kinds (which loop picks laysolfr, SW:3363-3380 vs SW:3533-3538)
enum SolarKind : int { SOL_UPPER = 0, SOL_LOWER = 1, SOL_LAYTROP = 2 };

struct GRegime {
    uint32_t cfg;
    int32_t off_major;   // offset (doubles) of K[row 0][ig] in the packed table
    double strrat;       // binary species ratio (band 22 lower: o2adj*strrat)
    double mfac;         // multiplies the major amount (givfac band 23, o2adj band 22 upper)
    double minorabs;     // per-g minor absorber coefficient (absch4, abso3a/b, absco2, absh2o)
    double rayl;         // per-g (or band scalar) Rayleigh coefficient
};

struct GDesc {
    GRegime lo, up;
    int32_t band;        // 0..13
    int32_t ig;          // 0-based g within band
    int32_t ng;          // g-points in band == row stride of every table of the band
    int32_t off_self;    // selfref[row][ig]  (10 rows)
    int32_t off_for;     // forref[row][ig]   (4 rows, zero padded)
    int32_t off_sflux;   // sfluxref[js][ig]
    int32_t off_rayla;   // rayla[js][ig] (band 24)
    int32_t nsf;         // second dim of sfluxref: 1, 5 or 9
    int32_t layreffr;
    int32_t solar_kind;
    int32_t uvband;      // 1 if ibm in 10..13 (SW:8637), else near-IR
    int32_t pad_;
    double scalekur;     // band 27 solar scaling, 1 elsewhere
};

struct SwDeviceTables {
    const double *K;         // packed k-tables (zero pad first)
    const double *exp_tbl;   // [NTBL+1]
    const GDesc *gdesc;      // [112]
    double bpade, heatfac, oneminus;
};

}  // namespace rrtmg
