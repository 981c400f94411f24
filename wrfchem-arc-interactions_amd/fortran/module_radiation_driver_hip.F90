!  module_radiation_driver_hip.F90 -- Fortran face of the driver-side neighbours of the radiation call.
!
!  Same names and dummy-argument lists as the two swint_opt routines of phys/module_radiation_driver.F
!  (update_swinterp_parameters :2668-2749, interp_sw_radiation :2751-2795), so that radiation_driver can USE
!  this module for them, plus accumulate_rad_fluxes for the AC* bucket loops of the driver body (:2308-2381).
!  No arithmetic here: the bodies marshal to librrtmg_hip.so (include/rrtmg_hip.h); host arrays, synchronous.

MODULE module_radiation_driver_hip

   USE, INTRINSIC :: iso_c_binding

   IMPLICIT NONE
   PRIVATE
   PUBLIC :: update_swinterp_parameters, interp_sw_radiation, accumulate_rad_fluxes

   INTEGER, PARAMETER :: errlen = 256

   ! mirrors rrtmg_hip_tile2d, rrtmg_hip_swinterp_update_args, rrtmg_hip_swinterp_apply_args of include/rrtmg_hip.h
   TYPE, BIND(C) :: tile2d
      INTEGER(c_int) :: ims, ime, jms, jme, its, ite, jts, jte
      INTEGER(c_int) :: real_bytes, mem
      TYPE(c_ptr)    :: stream
   END TYPE tile2d
   TYPE, BIND(C) :: update_args
      TYPE(c_ptr) :: coszen, coszen_loc, swddir, swdown
      TYPE(c_ptr) :: swddir_ref, bb, Bx, swdown_ref, gg, Gx, coszen_ref
   END TYPE update_args
   TYPE, BIND(C) :: apply_args
      TYPE(c_ptr) :: coszen_ref, coszen_loc, swddir_ref, bb, Bx, swdown_ref, gg, Gx, albedo
      TYPE(c_ptr) :: swdown, swddir, swddni, swddif, gsw
   END TYPE apply_args

   INTERFACE
      INTEGER(c_int) FUNCTION rrtmg_hip_update_swinterp_parameters(dims, args, err) &
            BIND(C, name='rrtmg_hip_update_swinterp_parameters')
         IMPORT :: c_int, c_char, tile2d, update_args
         TYPE(tile2d), INTENT(IN) :: dims
         TYPE(update_args), INTENT(IN) :: args
         CHARACTER(kind=c_char), INTENT(INOUT) :: err(*)
      END FUNCTION rrtmg_hip_update_swinterp_parameters
      INTEGER(c_int) FUNCTION rrtmg_hip_interp_sw_radiation(dims, args, err) BIND(C, name='rrtmg_hip_interp_sw_radiation')
         IMPORT :: c_int, c_char, tile2d, apply_args
         TYPE(tile2d), INTENT(IN) :: dims
         TYPE(apply_args), INTENT(IN) :: args
         CHARACTER(kind=c_char), INTENT(INOUT) :: err(*)
      END FUNCTION rrtmg_hip_interp_sw_radiation
      INTEGER(c_int) FUNCTION rrtmg_hip_accumulate_fluxes(dims, dtaccum, flux16, acc16, err) &
            BIND(C, name='rrtmg_hip_accumulate_fluxes')
         IMPORT :: c_int, c_double, c_char, c_ptr, tile2d
         TYPE(tile2d), INTENT(IN) :: dims
         REAL(c_double), VALUE :: dtaccum
         TYPE(c_ptr), INTENT(IN) :: flux16(16), acc16(16)
         CHARACTER(kind=c_char), INTENT(INOUT) :: err(*)
      END FUNCTION rrtmg_hip_accumulate_fluxes
   END INTERFACE

CONTAINS

   SUBROUTINE fatal(err)
      CHARACTER(kind=c_char), INTENT(IN) :: err(errlen)
      CHARACTER(len=errlen) :: msg
      INTEGER :: i
      EXTERNAL wrf_error_fatal
      msg = ' '
      DO i = 1, errlen
         IF (err(i) == c_null_char) EXIT
         msg(i:i) = err(i)
      END DO
      CALL wrf_error_fatal(TRIM(msg))
   END SUBROUTINE fatal

   FUNCTION tile_of(ims,ime,jms,jme,its,ite,jts,jte, sample) RESULT(t)
      INTEGER, INTENT(IN) :: ims,ime,jms,jme,its,ite,jts,jte
      REAL, INTENT(IN) :: sample
      TYPE(tile2d) :: t
      t%ims = ims; t%ime = ime; t%jms = jms; t%jme = jme
      t%its = its; t%ite = ite; t%jts = jts; t%jte = jte
      t%real_bytes = INT(STORAGE_SIZE(sample) / 8, c_int)
      t%mem = 0
      t%stream = c_null_ptr
   END FUNCTION tile_of

   SUBROUTINE update_swinterp_parameters(ims,ime,jms,jme,its,ite,jts,jte, &
                                         coszen,coszen_loc,swddir,swdown, &
                                         swddir_ref,bb,Bx,                &
                                         swdown_ref,gg,Gx,                &
                                         coszen_ref                       )
      INTEGER, INTENT(IN) :: ims,ime,jms,jme,its,ite,jts,jte
      REAL, DIMENSION(ims:ime,jms:jme), INTENT(IN), TARGET    :: coszen,coszen_loc,swddir,swdown
      REAL, DIMENSION(ims:ime,jms:jme), INTENT(INOUT), TARGET :: swddir_ref,bb,Bx,swdown_ref,gg,Gx,coszen_ref
      TYPE(tile2d) :: t
      TYPE(update_args) :: a
      CHARACTER(kind=c_char) :: err(errlen)
      t = tile_of(ims,ime,jms,jme,its,ite,jts,jte, coszen(ims,jms))
      a%coszen = c_loc(coszen); a%coszen_loc = c_loc(coszen_loc); a%swddir = c_loc(swddir); a%swdown = c_loc(swdown)
      a%swddir_ref = c_loc(swddir_ref); a%bb = c_loc(bb); a%Bx = c_loc(Bx); a%swdown_ref = c_loc(swdown_ref)
      a%gg = c_loc(gg); a%Gx = c_loc(Gx); a%coszen_ref = c_loc(coszen_ref)
      IF (rrtmg_hip_update_swinterp_parameters(t, a, err) /= 0) CALL fatal(err)
   END SUBROUTINE update_swinterp_parameters

   SUBROUTINE interp_sw_radiation(ims,ime,jms,jme,its,ite,jts,jte,  &
                                  coszen_ref,coszen_loc,swddir_ref, &
                                  bb,Bx,swdown_ref,gg,Gx,albedo,    &
                                  swdown,swddir,swddni,swddif,gsw   )
      INTEGER, INTENT(IN) :: ims,ime,jms,jme,its,ite,jts,jte
      REAL, DIMENSION(ims:ime,jms:jme), INTENT(IN), TARGET :: coszen_ref,coszen_loc,swddir_ref,Bx,bb,swdown_ref,Gx,gg,albedo
      REAL, DIMENSION(ims:ime,jms:jme), INTENT(INOUT), TARGET :: swddir,swdown,swddif,swddni,gsw
      TYPE(tile2d) :: t
      TYPE(apply_args) :: a
      CHARACTER(kind=c_char) :: err(errlen)
      t = tile_of(ims,ime,jms,jme,its,ite,jts,jte, albedo(ims,jms))
      a%coszen_ref = c_loc(coszen_ref); a%coszen_loc = c_loc(coszen_loc); a%swddir_ref = c_loc(swddir_ref)
      a%bb = c_loc(bb); a%Bx = c_loc(Bx); a%swdown_ref = c_loc(swdown_ref); a%gg = c_loc(gg); a%Gx = c_loc(Gx)
      a%albedo = c_loc(albedo)
      a%swdown = c_loc(swdown); a%swddir = c_loc(swddir); a%swddni = c_loc(swddni); a%swddif = c_loc(swddif)
      a%gsw = c_loc(gsw)
      IF (rrtmg_hip_interp_sw_radiation(t, a, err) /= 0) CALL fatal(err)
   END SUBROUTINE interp_sw_radiation

   ! ac(:,:,f) = ac(:,:,f) + flux(:,:,f) * dtaccum for the nf field pairs handed over (the loops at DRV:2326-2339
   ! and DRV:2361-2374 take eight each: upt, uptc, dnt, dntc, upb, upbc, dnb, dnbc)
   SUBROUTINE accumulate_rad_fluxes(ims,ime,jms,jme,its,ite,jts,jte, dtaccum, nf, flux, ac)
      INTEGER, INTENT(IN) :: ims,ime,jms,jme,its,ite,jts,jte, nf
      REAL, INTENT(IN) :: dtaccum
      REAL, DIMENSION(ims:ime,jms:jme,nf), INTENT(IN), TARGET    :: flux
      REAL, DIMENSION(ims:ime,jms:jme,nf), INTENT(INOUT), TARGET :: ac
      TYPE(tile2d) :: t
      TYPE(c_ptr) :: pf(16), pa(16)
      CHARACTER(kind=c_char) :: err(errlen)
      INTEGER :: f
      EXTERNAL wrf_error_fatal
      IF (nf > 16) CALL wrf_error_fatal('accumulate_rad_fluxes: at most 16 field pairs per call')
      t = tile_of(ims,ime,jms,jme,its,ite,jts,jte, dtaccum)
      pf = c_null_ptr; pa = c_null_ptr
      DO f = 1, nf
         pf(f) = c_loc(flux(ims,jms,f)); pa(f) = c_loc(ac(ims,jms,f))
      END DO
      IF (rrtmg_hip_accumulate_fluxes(t, REAL(dtaccum, c_double), pf, pa, err) /= 0) CALL fatal(err)
   END SUBROUTINE accumulate_rad_fluxes

END MODULE module_radiation_driver_hip
