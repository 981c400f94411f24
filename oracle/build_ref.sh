#!/bin/bash
# oracle/build_ref.sh -- TEST INFRASTRUCTURE.
#
# Builds oracle/_ref/ from the reference sources WHERE THEY LIE under
# /root/reference (nothing is copied into the repo; oracle/_ref is git-ignored).
# Only the self-contained inner modules of each file are compiled -- the line
# ranges that end just before the WRF-facing wrapper modules
# (MODULE module_ra_rrtmg_lw at LW:11408, MODULE module_ra_rrtmg_sw at SW:9878),
# because those wrappers USE WRF modules that do not exist here and the round
# rules forbid stand-ins for them.  The inner modules need no external module;
# their only external symbol is the error hook wrf_error_fatal, which the
# harness provides as an abort.
#
# -fdefault-real-8 makes kind_rb = kind(1.0) (LW:42) fp64, i.e. a WRF "-r8" build.
set -euo pipefail
REF=${REF_ROOT:-/root/reference}/WRF-Chem_code/v3.9.1/phys
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
FC=${FC:-/opt/rocm/lib/llvm/bin/flang}
if [ ! -d "$REF" ]; then echo "build_ref: $REF not present (GPU box) -- using prebuilt $OUT" >&2; exit 0; fi
mkdir -p "$OUT"
FFLAGS="-cpp -ffree-form -fdefault-real-8 -DWRF_CHEM=1 -DEM_CORE=1 -DRWORDSIZE=8 -DIWORDSIZE=4 -O2 -fPIC"
LW_END=11407   # last line before "MODULE module_ra_rrtmg_lw"
SW_END=9876    # last line before "MODULE module_ra_rrtmg_sw"
grep -n -m1 -i '^MODULE module_ra_rrtmg_lw' "$REF/module_ra_rrtmg_lw.F" | grep -q "^$((LW_END+1)):" || { echo "LW line range moved" >&2; exit 1; }
grep -n -m1 -i '^MODULE module_ra_rrtmg_sw' "$REF/module_ra_rrtmg_sw.F" | grep -q "^$((SW_END+2)):" || { echo "SW line range moved" >&2; exit 1; }
cd "$OUT"
sed -n "1,${LW_END}p" "$REF/module_ra_rrtmg_lw.F" | $FC $FFLAGS -c -x f95-cpp-input - -o lw_inner.o -module-dir "$OUT"
sed -n "1,${SW_END}p" "$REF/module_ra_rrtmg_sw.F" | $FC $FFLAGS -c -x f95-cpp-input - -o sw_inner.o -module-dir "$OUT"
$FC -fdefault-real-8 -O2 -I"$OUT" "$HERE/ref_harness_sw.F90" sw_inner.o lw_inner.o -o ref_sw -module-dir "$OUT"
if [ -f "$HERE/ref_harness_lw.F90" ]; then
  $FC -fdefault-real-8 -O2 -I"$OUT" "$HERE/ref_harness_lw.F90" lw_inner.o -o ref_lw -module-dir "$OUT"
fi
# swint_opt routines of the radiation driver: two self-contained subroutines (DRV:2668-2795), wrapped into a module
# on the fly (the driver module itself USEs most of WRF)
DRV="$REF/module_radiation_driver.F"
DRV_BEG=2668; DRV_END=2795
sed -n "${DRV_BEG}p" "$DRV" | grep -qi 'subroutine update_swinterp_parameters' || { echo "driver line range moved" >&2; exit 1; }
sed -n "${DRV_END}p" "$DRV" | grep -qi 'end subroutine interp_sw_radiation' || { echo "driver line range moved" >&2; exit 1; }
if [ -f "$HERE/ref_harness_swinterp.F90" ]; then
  { echo "module swinterp_ref"; echo "contains"; sed -n "${DRV_BEG},${DRV_END}p" "$DRV"; echo "end module swinterp_ref"; } \
    | $FC $FFLAGS -c -x f95-cpp-input - -o swinterp_inner.o -module-dir "$OUT"
  $FC -fdefault-real-8 -O2 -I"$OUT" "$HERE/ref_harness_swinterp.F90" swinterp_inner.o -o ref_swinterp -module-dir "$OUT"
fi
# helper routines of the LW wrapper module that need nothing from WRF: retab, INIRAD/O3DATA, relcalc/reicalc
LWF="$REF/module_ra_rrtmg_lw.F"
sed -n "11424p" "$LWF" | grep -qi 'real retab(95)' || { echo "retab line range moved" >&2; exit 1; }
sed -n "12704p" "$LWF" | grep -qi 'SUBROUTINE INIRAD' || { echo "INIRAD line range moved" >&2; exit 1; }
sed -n "12840p" "$LWF" | grep -qi 'END SUBROUTINE O3DATA' || { echo "O3DATA line range moved" >&2; exit 1; }
sed -n "14403p" "$LWF" | grep -qi 'subroutine relcalc' || { echo "relcalc line range moved" >&2; exit 1; }
sed -n "14491p" "$LWF" | grep -qi 'end subroutine reicalc' || { echo "reicalc line range moved" >&2; exit 1; }
if [ -f "$HERE/ref_harness_wrapper.F90" ]; then
  { echo "module wrapper_ref"; sed -n "11424,11444p" "$LWF"; echo "contains"; sed -n "12704,12840p" "$LWF";
    sed -n "14403,14491p" "$LWF"; echo "end module wrapper_ref"; } \
    | $FC $FFLAGS -c -x f95-cpp-input - -o wrapper_inner.o -module-dir "$OUT"
  $FC -fdefault-real-8 -O2 -I"$OUT" "$HERE/ref_harness_wrapper.F90" wrapper_inner.o -o ref_wrapper -module-dir "$OUT"
fi
echo "build_ref: built $(ls $OUT/ref_* | tr '\n' ' ')"
