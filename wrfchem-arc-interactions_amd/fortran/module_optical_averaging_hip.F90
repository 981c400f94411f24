!  module_optical_averaging_hip.F90 -- Fortran face of the HIP aerosol optical-averaging stage.
!
!  Upstream WRF-Chem v3.9.1 does this work in chem/module_optical_averaging.F (subroutine
!  optical_averaging, driven from chem/optical_driver.F).  That file is NOT part of the reference repository
!  (douglowe/WRFChem-ARC-Interactions carries only the patched radiation / registry files), so its dummy-argument
!  list cannot be mirrored from source here.  What the reference does fix is both ends of the stage:
!     inputs   MOSAIC sectional tracers chem(:,:,:,p_so4_a01 ...)      Registry/registry.chem:2325-2447
!     outputs  tauaer1..4, gaer1..4, waer1..4, tauaerlw1..16, ...       Registry/registry.chem:1332-1387
!              (zeroed in chem/chemics_init.F:486-553, handed to the radiation driver at
!               dyn_em/module_first_rk_step_part1.F:374-385)
!  and this module is written against exactly those: optical_driver (or chem_driver) calls
!  optical_averaging_hip with the chem array, the p_* species numbers of module_state_description and the
!  registry output arrays.  No arithmetic here: the bodies marshal to librrtmg_hip.so (include/rrtmg_hip.h).

MODULE module_optical_averaging_hip

   USE, INTRINSIC :: iso_c_binding

   IMPLICIT NONE
   PRIVATE
   PUBLIC :: optical_averaging_hip, optical_averaging_hip_init, modal_to_sections_hip, optical_averaging_hip_precision
   PUBLIC :: aer_so4, aer_no3, aer_cl, aer_nh4, aer_na, aer_oin, aer_oc, aer_bc, aer_water, aer_num, aer_nspec, aer_maxbin

   INTEGER, PARAMETER :: errlen = 256
   ! slots of idx(:, bin) -- order of enum RRTMG_HIP_AER_* in include/rrtmg_hip.h
   INTEGER, PARAMETER :: aer_so4 = 1, aer_no3 = 2, aer_cl = 3, aer_nh4 = 4, aer_na = 5, aer_oin = 6, aer_oc = 7, &
                         aer_bc = 8, aer_water = 9, aer_num = 10, aer_nspec = 10, aer_maxbin = 8

   ! mirrors struct rrtmg_hip_optics_args of include/rrtmg_hip.h
   TYPE, BIND(C) :: optics_args
      INTEGER(c_int) :: ims, ime, jms, jme, kms, kme
      INTEGER(c_int) :: its, ite, jts, jte, kts, kte
      INTEGER(c_int) :: real_bytes, mem
      TYPE(c_ptr)    :: stream
      INTEGER(c_int) :: aer_op_opt, num_chem
      INTEGER(c_int) :: idx(aer_nspec, aer_maxbin)        ! C: idx[bin][species]
      TYPE(c_ptr) :: chem, alt, dz8w
      TYPE(c_ptr) :: tauaer(4), waer(4), gaer(4), extaer(4), bscoef(4)
      TYPE(c_ptr) :: tauaerlw(16), extaerlw(16)
      INTEGER(c_int) :: idx_cw(aer_nspec, aer_maxbin)     ! C: idx_cw[bin][species]
      TYPE(c_ptr) :: laer(6)
   END TYPE optics_args

   ! mirrors struct rrtmg_hip_modal_args of include/rrtmg_hip.h
   INTEGER, PARAMETER :: aer_maxmode = 4
   TYPE, BIND(C) :: modal_args
      INTEGER(c_int) :: ims, ime, jms, jme, kms, kme
      INTEGER(c_int) :: its, ite, jts, jte, kts, kte
      INTEGER(c_int) :: real_bytes, mem
      TYPE(c_ptr)    :: stream
      INTEGER(c_int) :: nmode, num_chem
      REAL(c_double) :: sigma(aer_maxmode)
      INTEGER(c_int) :: idx(aer_nspec, aer_maxmode)       ! C: idx[mode][species]
      TYPE(c_ptr) :: chem, chem_sec
   END TYPE modal_args

   INTERFACE
      INTEGER(c_int) FUNCTION rrtmg_hip_optics_init(nbin, dlo_um, dhi_um, refindex, density, device, err) &
            BIND(C, name='rrtmg_hip_optics_init')
         IMPORT :: c_int, c_double, c_char, c_ptr
         INTEGER(c_int), VALUE :: nbin, device
         REAL(c_double), VALUE :: dlo_um, dhi_um
         TYPE(c_ptr), VALUE :: refindex, density
         CHARACTER(kind=c_char), INTENT(INOUT) :: err(*)
      END FUNCTION rrtmg_hip_optics_init
      INTEGER(c_int) FUNCTION rrtmg_hip_optical_averaging(args, err) BIND(C, name='rrtmg_hip_optical_averaging')
         IMPORT :: c_int, c_char, optics_args
         TYPE(optics_args), INTENT(IN) :: args
         CHARACTER(kind=c_char), INTENT(INOUT) :: err(*)
      END FUNCTION rrtmg_hip_optical_averaging
      INTEGER(c_int) FUNCTION rrtmg_hip_optics_precision(mode, err) BIND(C, name='rrtmg_hip_optics_precision')
         IMPORT :: c_int, c_char
         INTEGER(c_int), VALUE :: mode
         CHARACTER(kind=c_char), INTENT(INOUT) :: err(*)
      END FUNCTION rrtmg_hip_optics_precision
      INTEGER(c_int) FUNCTION rrtmg_hip_modal_to_sections(args, err) BIND(C, name='rrtmg_hip_modal_to_sections')
         IMPORT :: c_int, c_char, modal_args
         TYPE(modal_args), INTENT(IN) :: args
         CHARACTER(kind=c_char), INTENT(INOUT) :: err(*)
      END FUNCTION rrtmg_hip_modal_to_sections
      INTEGER(c_int) FUNCTION rrtmg_hip_pick_device() BIND(C, name='rrtmg_hip_pick_device')
         IMPORT :: c_int
      END FUNCTION rrtmg_hip_pick_device
   END INTERFACE

CONTAINS

   SUBROUTINE fatal_from_c(err)
      CHARACTER(kind=c_char), INTENT(IN) :: err(errlen)
      CHARACTER(len=errlen) :: msg
      INTEGER :: n
      msg = ' '
      DO n = 1, errlen
         IF (err(n) == c_null_char) EXIT
         msg(n:n) = err(n)
      END DO
      CALL wrf_error_fatal(TRIM(msg))
   END SUBROUTINE fatal_from_c

!------------------------------------------------------------------
!  nbin_o sections between dlo_sect(1) and dhi_sect(nbin_o) in um (MOSAIC: 0.0390625, 10.0), equal in
!  log diameter; built-in material table.  Call once per process, e.g. from chem_init.
   SUBROUTINE optical_averaging_hip_init(nbin_o, dlo_um, dhi_um)
      INTEGER, INTENT(IN) :: nbin_o
      REAL, INTENT(IN) :: dlo_um, dhi_um
      CHARACTER(kind=c_char) :: err(errlen)
      INTEGER(c_int) :: rc
      err = c_null_char
      rc = rrtmg_hip_optics_init(INT(nbin_o, c_int), REAL(dlo_um, c_double), REAL(dhi_um, c_double), &
                                 c_null_ptr, c_null_ptr, rrtmg_hip_pick_device(), err)
      IF (rc /= 0) CALL fatal_from_c(err)
   END SUBROUTINE optical_averaging_hip_init

!------------------------------------------------------------------
!  mode 0: fp64 throughout (default); 1: "mixed" -- the optical averaging in fp32
   SUBROUTINE optical_averaging_hip_precision(mode)
      INTEGER, INTENT(IN) :: mode
      CHARACTER(kind=c_char) :: err(errlen)
      err = c_null_char
      IF (rrtmg_hip_optics_precision(INT(mode, c_int), err) /= 0) CALL fatal_from_c(err)
   END SUBROUTINE optical_averaging_hip_precision

!------------------------------------------------------------------
!  Modal aerosol (MADE/SORGAM type) -> sections: idx(species, mode) as below but per mode (e.g. idx(aer_so4,2) =
!  p_so4aj), sigma(mode) the geometric standard deviations (MADE: 1.7, 2.0, 2.2).  chem_sec(i,k,j, nbin*10) and
!  idx_sec(species, bin) are what optical_averaging_hip then takes (with num_chem = nbin*10).
   SUBROUTINE modal_to_sections_hip(nbin_o, nmode, sigma, num_chem, idx, chem, chem_sec, idx_sec, &
                                    ims,ime, jms,jme, kms,kme, its,ite, jts,jte, kts,kte)
      INTEGER, INTENT(IN) :: nbin_o, nmode, num_chem
      INTEGER, INTENT(IN) :: ims,ime, jms,jme, kms,kme, its,ite, jts,jte, kts,kte
      REAL, INTENT(IN) :: sigma(nmode)
      INTEGER, INTENT(IN) :: idx(aer_nspec, nmode)
      REAL, DIMENSION(ims:ime,kms:kme,jms:jme,num_chem), INTENT(IN), TARGET :: chem
      REAL, DIMENSION(ims:ime,kms:kme,jms:jme,nbin_o*aer_nspec), INTENT(INOUT), TARGET :: chem_sec
      INTEGER, INTENT(OUT) :: idx_sec(aer_nspec, nbin_o)
      TYPE(modal_args) :: a
      CHARACTER(kind=c_char) :: err(errlen)
      INTEGER :: m, b, s
      EXTERNAL wrf_error_fatal
      IF (nmode > aer_maxmode) CALL wrf_error_fatal('modal_to_sections_hip: at most 4 modes')
      a%ims = ims; a%ime = ime; a%jms = jms; a%jme = jme; a%kms = kms; a%kme = kme
      a%its = its; a%ite = ite; a%jts = jts; a%jte = jte; a%kts = kts; a%kte = kte
      a%real_bytes = INT(STORAGE_SIZE(sigma(1)) / 8, c_int)
      a%mem = 0
      a%stream = c_null_ptr
      a%nmode = nmode; a%num_chem = num_chem
      a%sigma = 2.0_c_double; a%idx = 0
      DO m = 1, nmode
         a%sigma(m) = REAL(sigma(m), c_double)
         a%idx(:, m) = idx(:, m)
      END DO
      a%chem = c_loc(chem); a%chem_sec = c_loc(chem_sec)
      err = c_null_char
      IF (rrtmg_hip_modal_to_sections(a, err) /= 0) CALL fatal_from_c(err)
      DO b = 1, nbin_o
         DO s = 1, aer_nspec
            idx_sec(s, b) = (b - 1) * aer_nspec + s
         END DO
      END DO
   END SUBROUTINE modal_to_sections_hip

!------------------------------------------------------------------
!  idx(species, bin): species number in chem of aer_so4 .. aer_num for each section (0 = not carried), e.g.
!  idx(aer_so4,1) = p_so4_a01.  aer_op_opt: registry.chem:4014-4018.  idx_cw (optional): the same for the cloud-borne
!  counterparts of the aqueous packages (idx_cw(aer_so4,1) = p_so4_cw01, registry.chem:3935-3936), whose mass and
!  number are added to the interstitial ones; absent = interstitial aerosol alone.
   SUBROUTINE optical_averaging_hip(nbin_o, aer_op_opt, num_chem, idx, chem, dz8w, alt,          &
                       tauaer1, tauaer2, tauaer3, tauaer4,                                       &
                       gaer1, gaer2, gaer3, gaer4,                                               &
                       waer1, waer2, waer3, waer4,                                               &
                       tauaerlw1, tauaerlw2, tauaerlw3, tauaerlw4, tauaerlw5, tauaerlw6,         &
                       tauaerlw7, tauaerlw8, tauaerlw9, tauaerlw10, tauaerlw11, tauaerlw12,      &
                       tauaerlw13, tauaerlw14, tauaerlw15, tauaerlw16,                           &
                       ids,ide, jds,jde, kds,kde,                                                &
                       ims,ime, jms,jme, kms,kme,                                                &
                       its,ite, jts,jte, kts,kte,                                                &
                       extaer1, extaer2, extaer3, extaer4,                                       &
                       bscoef1, bscoef2, bscoef3, bscoef4,                                       &
                       extaerlw, idx_cw, l2aer, l3aer, l4aer, l5aer, l6aer, l7aer)
      INTEGER, INTENT(IN) :: nbin_o, aer_op_opt, num_chem
      INTEGER, INTENT(IN) :: idx(aer_nspec, nbin_o)
      INTEGER, INTENT(IN), OPTIONAL :: idx_cw(aer_nspec, nbin_o)
      INTEGER, INTENT(IN) :: ids,ide, jds,jde, kds,kde, ims,ime, jms,jme, kms,kme, its,ite, jts,jte, kts,kte
      ! Legendre moments of the phase function at the four wavelengths (registry.chem:1382-1387): all six or none
      REAL, DIMENSION(ims:ime, kms:kme, jms:jme, 4), INTENT(INOUT), OPTIONAL, TARGET ::           &
            l2aer, l3aer, l4aer, l5aer, l6aer, l7aer
      REAL, DIMENSION(ims:ime, kms:kme, jms:jme, num_chem), INTENT(IN), TARGET :: chem
      REAL, DIMENSION(ims:ime, kms:kme, jms:jme), INTENT(IN), TARGET :: dz8w, alt
      REAL, DIMENSION(ims:ime, kms:kme, jms:jme), INTENT(INOUT), TARGET ::                       &
            tauaer1, tauaer2, tauaer3, tauaer4, gaer1, gaer2, gaer3, gaer4, waer1, waer2, waer3, waer4
      REAL, DIMENSION(ims:ime, kms:kme, jms:jme), INTENT(INOUT), TARGET ::                       &
            tauaerlw1, tauaerlw2, tauaerlw3, tauaerlw4, tauaerlw5, tauaerlw6, tauaerlw7, tauaerlw8, &
            tauaerlw9, tauaerlw10, tauaerlw11, tauaerlw12, tauaerlw13, tauaerlw14, tauaerlw15, tauaerlw16
      REAL, DIMENSION(ims:ime, kms:kme, jms:jme), INTENT(INOUT), OPTIONAL, TARGET ::             &
            extaer1, extaer2, extaer3, extaer4, bscoef1, bscoef2, bscoef3, bscoef4
      REAL, DIMENSION(ims:ime, kms:kme, jms:jme, 16), INTENT(INOUT), OPTIONAL, TARGET :: extaerlw

      TYPE(optics_args) :: a
      CHARACTER(kind=c_char) :: err(errlen)
      INTEGER(c_int) :: rc
      INTEGER :: b, s

      IF (nbin_o < 1 .OR. nbin_o > aer_maxbin) CALL wrf_error_fatal('optical_averaging_hip: nbin_o must be 1..8')
      a%ims = ims; a%ime = ime; a%jms = jms; a%jme = jme; a%kms = kms; a%kme = kme
      a%its = its; a%ite = ite; a%jts = jts; a%jte = jte; a%kts = kts; a%kte = kte
      a%real_bytes = STORAGE_SIZE(dz8w) / 8            ! 4: stock WRF, 8: -r8
      a%mem = 0
      a%stream = c_null_ptr
      a%aer_op_opt = aer_op_opt
      a%num_chem = num_chem
      a%idx = 0
      a%idx_cw = 0
      DO b = 1, nbin_o
         DO s = 1, aer_nspec
            a%idx(s, b) = idx(s, b)
            IF (PRESENT(idx_cw)) a%idx_cw(s, b) = idx_cw(s, b)
         END DO
      END DO
      a%chem = c_loc(chem); a%alt = c_loc(alt); a%dz8w = c_loc(dz8w)
      a%tauaer(1) = c_loc(tauaer1); a%tauaer(2) = c_loc(tauaer2); a%tauaer(3) = c_loc(tauaer3); a%tauaer(4) = c_loc(tauaer4)
      a%waer(1) = c_loc(waer1); a%waer(2) = c_loc(waer2); a%waer(3) = c_loc(waer3); a%waer(4) = c_loc(waer4)
      a%gaer(1) = c_loc(gaer1); a%gaer(2) = c_loc(gaer2); a%gaer(3) = c_loc(gaer3); a%gaer(4) = c_loc(gaer4)
      a%extaer(1) = opt3(extaer1); a%extaer(2) = opt3(extaer2); a%extaer(3) = opt3(extaer3); a%extaer(4) = opt3(extaer4)
      a%bscoef(1) = opt3(bscoef1); a%bscoef(2) = opt3(bscoef2); a%bscoef(3) = opt3(bscoef3); a%bscoef(4) = opt3(bscoef4)
      a%tauaerlw(1) = c_loc(tauaerlw1);   a%tauaerlw(2) = c_loc(tauaerlw2);   a%tauaerlw(3) = c_loc(tauaerlw3)
      a%tauaerlw(4) = c_loc(tauaerlw4);   a%tauaerlw(5) = c_loc(tauaerlw5);   a%tauaerlw(6) = c_loc(tauaerlw6)
      a%tauaerlw(7) = c_loc(tauaerlw7);   a%tauaerlw(8) = c_loc(tauaerlw8);   a%tauaerlw(9) = c_loc(tauaerlw9)
      a%tauaerlw(10) = c_loc(tauaerlw10); a%tauaerlw(11) = c_loc(tauaerlw11); a%tauaerlw(12) = c_loc(tauaerlw12)
      a%tauaerlw(13) = c_loc(tauaerlw13); a%tauaerlw(14) = c_loc(tauaerlw14); a%tauaerlw(15) = c_loc(tauaerlw15)
      a%tauaerlw(16) = c_loc(tauaerlw16)
      DO b = 1, 16
         a%extaerlw(b) = c_null_ptr
         IF (PRESENT(extaerlw)) a%extaerlw(b) = c_loc(extaerlw(ims, kms, jms, b))
      END DO
      a%laer = c_null_ptr
      IF (PRESENT(l2aer)) a%laer(1) = c_loc(l2aer)
      IF (PRESENT(l3aer)) a%laer(2) = c_loc(l3aer)
      IF (PRESENT(l4aer)) a%laer(3) = c_loc(l4aer)
      IF (PRESENT(l5aer)) a%laer(4) = c_loc(l5aer)
      IF (PRESENT(l6aer)) a%laer(5) = c_loc(l6aer)
      IF (PRESENT(l7aer)) a%laer(6) = c_loc(l7aer)
      err = c_null_char
      rc = rrtmg_hip_optical_averaging(a, err)
      IF (rc /= 0) CALL fatal_from_c(err)

   CONTAINS
      FUNCTION opt3(x) RESULT(p)
         REAL, DIMENSION(ims:ime, kms:kme, jms:jme), INTENT(IN), OPTIONAL, TARGET :: x
         TYPE(c_ptr) :: p
         p = c_null_ptr
         IF (PRESENT(x)) p = c_loc(x)
      END FUNCTION opt3
   END SUBROUTINE optical_averaging_hip

END MODULE module_optical_averaging_hip
