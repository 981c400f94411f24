!  module_ra_rrtmg_lw_hip.F90 -- drop-in replacement for WRF-Chem v3.9.1's
!  phys/module_ra_rrtmg_lw.F (patched for the clean-sky double call by
!  douglowe/WRFChem-ARC-Interactions).
!
!  Keeps the module name and the two procedures radiation_driver uses --
!     rrtmg_lwinit   (reference: module_ra_rrtmg_lw.F:12845-12874)
!     RRTMG_LWRAD    (reference: module_ra_rrtmg_lw.F:11451-11486; called by
!                     keyword from module_radiation_driver.F:1528-1577)
!  -- with the reference's dummy-argument names, types, shapes, INTENTs and
!  OPTIONAL attributes.  The bodies only marshal to the C ABI of
!  librrtmg_hip.so (include/rrtmg_hip.h); no radiation arithmetic here.
!
!  Not provided: inirad, o3data, relcalc, reicalc, retab, which the reference
!  module also exports for the rrtmg_*f ("fast") schemes (see INTEGRATION.md).

MODULE module_ra_rrtmg_lw

   USE, INTRINSIC :: iso_c_binding
   USE module_model_constants, ONLY : cp

   IMPLICIT NONE
   PRIVATE
   PUBLIC :: rrtmg_lwrad, rrtmg_lwinit

   INTEGER, PARAMETER :: errlen = 256

   ! mirrors struct rrtmg_hip_lwrad_args of include/rrtmg_hip.h
   TYPE, BIND(C) :: lwrad_args
      INTEGER(c_int) :: ims, ime, jms, jme, kms, kme
      INTEGER(c_int) :: its, ite, jts, jte, kts, kte
      INTEGER(c_int) :: real_bytes, mem
      TYPE(c_ptr)    :: stream
      INTEGER(c_int) :: aer_ra_feedback, clean_atm_diag, icloud, o3input
      TYPE(c_ptr) :: emiss, tsk
      TYPE(c_ptr) :: t3d, t8w, p3d, p8w, pi3d, qv3d, o33d
      TYPE(c_ptr) :: tauaerlw(16)
      TYPE(c_ptr) :: rthratenlw
      TYPE(c_ptr) :: glw, olr, lwcf
      TYPE(c_ptr) :: lwupt, lwuptc, lwuptcln, lwdnt, lwdntc, lwdntcln
      TYPE(c_ptr) :: lwupb, lwupbc, lwupbcln, lwdnb, lwdnbc, lwdnbcln
      TYPE(c_ptr) :: lwupflx, lwupflxc, lwupflxcln, lwdnflx, lwdnflxc, lwdnflxcln
      TYPE(c_ptr) :: cldfra3d, qc3d, qr3d, qi3d, qs3d
      TYPE(c_ptr) :: re_cloud, re_ice, re_snow
      TYPE(c_ptr) :: qndrop3d
      TYPE(c_ptr) :: xland, xice, snow
      INTEGER(c_int) :: has_reqc, has_reqi, has_reqs
      INTEGER(c_int) :: warm_rain, progn
      INTEGER(c_int) :: is_cammgmp_used, f_ice_phy, mp_physics
      REAL(c_double) :: g
   END TYPE lwrad_args

   INTERFACE
      INTEGER(c_int) FUNCTION rrtmg_hip_lw_init(path, real_bytes, cpdair, p_top, kme, device, err) &
            BIND(C, name='rrtmg_hip_lw_init')
         IMPORT :: c_int, c_double, c_char
         CHARACTER(kind=c_char), INTENT(IN) :: path(*)
         INTEGER(c_int), VALUE :: real_bytes, kme, device
         REAL(c_double), VALUE :: cpdair, p_top
         CHARACTER(kind=c_char), INTENT(INOUT) :: err(*)
      END FUNCTION rrtmg_hip_lw_init
      INTEGER(c_int) FUNCTION rrtmg_hip_lwrad(args, err) BIND(C, name='rrtmg_hip_lwrad')
         IMPORT :: c_int, c_char, lwrad_args
         TYPE(lwrad_args), INTENT(IN) :: args
         CHARACTER(kind=c_char), INTENT(INOUT) :: err(*)
      END FUNCTION rrtmg_hip_lwrad
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
   SUBROUTINE RRTMG_LWRAD(                                        &
                       rthratenlw,                                &
                       lwupt, lwuptc, lwuptcln, lwdnt, lwdntc, lwdntcln, &
                       lwupb, lwupbc, lwupbcln, lwdnb, lwdnbc, lwdnbcln, &
                       glw, olr, lwcf, emiss,                     &
                       p8w, p3d, pi3d,                            &
                       dz8w, tsk, t3d, t8w, rho3d, r, g,          &
                       icloud, warm_rain, cldfra3d,               &
                       lradius,iradius,                           &
                       is_cammgmp_used,                           &
                       f_ice_phy, f_rain_phy,                     &
                       xland, xice, snow,                         &
                       qv3d, qc3d, qr3d,                          &
                       qi3d, qs3d, qg3d,                          &
                       o3input, o33d,                             &
                       f_qv, f_qc, f_qr, f_qi, f_qs, f_qg,        &
                       re_cloud, re_ice, re_snow,                 &
                       has_reqc, has_reqi, has_reqs,              &
                       tauaerlw1,tauaerlw2,tauaerlw3,tauaerlw4,   &
                       tauaerlw5,tauaerlw6,tauaerlw7,tauaerlw8,   &
                       tauaerlw9,tauaerlw10,tauaerlw11,tauaerlw12,   &
                       tauaerlw13,tauaerlw14,tauaerlw15,tauaerlw16,   &
                       aer_ra_feedback,                           &
                       progn,clean_atm_diag,                      &
                       qndrop3d,f_qndrop,                         &
                       yr,julian,                                 &
                       mp_physics,                                &
                       ids,ide, jds,jde, kds,kde,                 &
                       ims,ime, jms,jme, kms,kme,                 &
                       its,ite, jts,jte, kts,kte,                 &
                       lwupflx, lwupflxc, lwupflxcln, lwdnflx, lwdnflxc, lwdnflxcln &
                                                                  )
      LOGICAL, INTENT(IN) :: warm_rain, is_cammgmp_used
      INTEGER, INTENT(IN) :: ids,ide, jds,jde, kds,kde, ims,ime, jms,jme, kms,kme, its,ite, jts,jte, kts,kte
      INTEGER, INTENT(IN) :: icloud, mp_physics
      REAL, DIMENSION(ims:ime,kms:kme,jms:jme), INTENT(IN), TARGET :: dz8w, t3d, t8w, p8w, p3d, pi3d, rho3d
      REAL, DIMENSION(ims:ime,kms:kme,jms:jme), INTENT(INOUT), TARGET :: rthratenlw
      REAL, DIMENSION(ims:ime,jms:jme), INTENT(INOUT), TARGET :: glw, olr, lwcf
      REAL, DIMENSION(ims:ime,jms:jme), INTENT(IN), TARGET :: emiss, tsk
      REAL, INTENT(IN) :: r, g
      REAL, DIMENSION(ims:ime,jms:jme), INTENT(IN), TARGET :: xland, xice, snow
      INTEGER, INTENT(IN) :: yr
      REAL, INTENT(IN) :: julian
      REAL, DIMENSION(ims:ime,kms:kme,jms:jme), OPTIONAL, INTENT(IN), TARGET :: &
           cldfra3d, lradius, iradius, qv3d, qc3d, qr3d, qi3d, qs3d, qg3d, qndrop3d
      REAL, DIMENSION(ims:ime,kms:kme,jms:jme), INTENT(IN), TARGET :: re_cloud, re_ice, re_snow
      INTEGER, INTENT(IN) :: has_reqc, has_reqi, has_reqs
      REAL, DIMENSION(ims:ime,kms:kme,jms:jme), OPTIONAL, INTENT(IN) :: f_ice_phy, f_rain_phy
      LOGICAL, OPTIONAL, INTENT(IN) :: f_qv, f_qc, f_qr, f_qi, f_qs, f_qg, f_qndrop
      REAL, DIMENSION(ims:ime,kms:kme,jms:jme), OPTIONAL, INTENT(IN), TARGET :: &
           tauaerlw1, tauaerlw2, tauaerlw3, tauaerlw4, tauaerlw5, tauaerlw6, tauaerlw7, tauaerlw8, &
           tauaerlw9, tauaerlw10, tauaerlw11, tauaerlw12, tauaerlw13, tauaerlw14, tauaerlw15, tauaerlw16
      INTEGER, INTENT(IN), OPTIONAL :: aer_ra_feedback, progn, clean_atm_diag
      REAL, DIMENSION(ims:ime,kms:kme,jms:jme), OPTIONAL, INTENT(IN), TARGET :: o33d
      INTEGER, OPTIONAL, INTENT(IN) :: o3input
      REAL, DIMENSION(ims:ime,jms:jme), OPTIONAL, INTENT(INOUT), TARGET :: &
           lwupt, lwuptc, lwuptcln, lwdnt, lwdntc, lwdntcln, lwupb, lwupbc, lwupbcln, lwdnb, lwdnbc, lwdnbcln
      REAL, DIMENSION(ims:ime,kms:kme+2,jms:jme), OPTIONAL, INTENT(OUT), TARGET :: &
           lwupflx, lwupflxc, lwupflxcln, lwdnflx, lwdnflxc, lwdnflxcln

      TYPE(lwrad_args) :: a
      CHARACTER(kind=c_char) :: err(errlen)
      INTEGER(c_int) :: rc

      a%ims = ims; a%ime = ime; a%jms = jms; a%jme = jme; a%kms = kms; a%kme = kme
      a%its = its; a%ite = ite; a%jts = jts; a%jte = jte; a%kts = kts; a%kte = kte
      a%real_bytes = STORAGE_SIZE(r) / 8
      a%mem = 0
      a%stream = c_null_ptr
      a%aer_ra_feedback = 0
      IF (PRESENT(aer_ra_feedback)) a%aer_ra_feedback = aer_ra_feedback
      a%clean_atm_diag = 0
      IF (PRESENT(clean_atm_diag)) a%clean_atm_diag = clean_atm_diag
      a%icloud = icloud
      a%o3input = 0
      IF (PRESENT(o3input)) a%o3input = o3input
      IF (.NOT. PRESENT(qv3d)) CALL wrf_error_fatal('rrtmg_lwrad (HIP): qv3d is required')

      a%emiss = c_loc(emiss); a%tsk = c_loc(tsk)
      a%t3d = c_loc(t3d); a%t8w = c_loc(t8w); a%p3d = c_loc(p3d); a%p8w = c_loc(p8w); a%pi3d = c_loc(pi3d)
      a%qv3d = c_loc(qv3d)
      a%o33d = opt3(o33d)
      a%tauaerlw(1) = opt3(tauaerlw1);   a%tauaerlw(2) = opt3(tauaerlw2);   a%tauaerlw(3) = opt3(tauaerlw3)
      a%tauaerlw(4) = opt3(tauaerlw4);   a%tauaerlw(5) = opt3(tauaerlw5);   a%tauaerlw(6) = opt3(tauaerlw6)
      a%tauaerlw(7) = opt3(tauaerlw7);   a%tauaerlw(8) = opt3(tauaerlw8);   a%tauaerlw(9) = opt3(tauaerlw9)
      a%tauaerlw(10) = opt3(tauaerlw10); a%tauaerlw(11) = opt3(tauaerlw11); a%tauaerlw(12) = opt3(tauaerlw12)
      a%tauaerlw(13) = opt3(tauaerlw13); a%tauaerlw(14) = opt3(tauaerlw14); a%tauaerlw(15) = opt3(tauaerlw15)
      a%tauaerlw(16) = opt3(tauaerlw16)
      a%rthratenlw = c_loc(rthratenlw)
      a%glw = c_loc(glw); a%olr = c_loc(olr); a%lwcf = c_loc(lwcf)
      a%lwupt = opt2(lwupt); a%lwuptc = opt2(lwuptc); a%lwuptcln = opt2(lwuptcln)
      a%lwdnt = opt2(lwdnt); a%lwdntc = opt2(lwdntc); a%lwdntcln = opt2(lwdntcln)
      a%lwupb = opt2(lwupb); a%lwupbc = opt2(lwupbc); a%lwupbcln = opt2(lwupbcln)
      a%lwdnb = opt2(lwdnb); a%lwdnbc = opt2(lwdnbc); a%lwdnbcln = opt2(lwdnbcln)
      a%lwupflx = optf(lwupflx); a%lwupflxc = optf(lwupflxc); a%lwupflxcln = optf(lwupflxcln)
      a%lwdnflx = optf(lwdnflx); a%lwdnflxc = optf(lwdnflxc); a%lwdnflxcln = optf(lwdnflxcln)

      ! clouds (icloud /= 0): a hydrometeor array is handed over only when it is PRESENT and its F_Q* flag is set,
      ! which is exactly when the reference reads it (LW:11921-11990)
      a%cldfra3d = opt3(cldfra3d)
      a%qc3d = c_null_ptr; a%qr3d = c_null_ptr; a%qi3d = c_null_ptr; a%qs3d = c_null_ptr; a%qndrop3d = c_null_ptr
      IF (PRESENT(f_qc) .AND. PRESENT(qc3d)) THEN
         IF (f_qc) a%qc3d = c_loc(qc3d)
      END IF
      IF (PRESENT(f_qr) .AND. PRESENT(qr3d)) THEN
         IF (f_qr) a%qr3d = c_loc(qr3d)
      END IF
      IF (PRESENT(f_qi) .AND. PRESENT(qi3d)) THEN
         IF (f_qi) a%qi3d = c_loc(qi3d)
      END IF
      IF (PRESENT(f_qs) .AND. PRESENT(qs3d)) THEN
         IF (f_qs) a%qs3d = c_loc(qs3d)
      END IF
      IF (PRESENT(f_qndrop) .AND. PRESENT(qndrop3d)) THEN
         IF (f_qndrop) a%qndrop3d = c_loc(qndrop3d)
      END IF
      a%re_cloud = c_loc(re_cloud); a%re_ice = c_loc(re_ice); a%re_snow = c_loc(re_snow)
      a%xland = c_loc(xland); a%xice = c_loc(xice); a%snow = c_loc(snow)
      a%has_reqc = has_reqc; a%has_reqi = has_reqi; a%has_reqs = has_reqs
      a%warm_rain = MERGE(1, 0, warm_rain)
      a%progn = 0
      IF (PRESENT(progn)) a%progn = progn
      a%is_cammgmp_used = MERGE(1, 0, is_cammgmp_used)
      a%f_ice_phy = 0
      IF (PRESENT(f_ice_phy) .AND. PRESENT(f_qc) .AND. PRESENT(f_qi) .AND. PRESENT(f_qs)) THEN
         IF (f_qc .AND. .NOT. f_qi .AND. f_qs) a%f_ice_phy = 1
      END IF
      a%mp_physics = mp_physics
      a%g = REAL(g, c_double)

      rc = rrtmg_hip_lwrad(a, err)
      IF (rc /= 0) CALL fatal_from_c(err)

   CONTAINS
      FUNCTION opt3(x) RESULT(p)
         REAL, DIMENSION(ims:ime,kms:kme,jms:jme), OPTIONAL, INTENT(IN), TARGET :: x
         TYPE(c_ptr) :: p
         p = c_null_ptr
         IF (PRESENT(x)) p = c_loc(x)
      END FUNCTION opt3
      FUNCTION opt2(x) RESULT(p)
         REAL, DIMENSION(ims:ime,jms:jme), OPTIONAL, INTENT(IN), TARGET :: x
         TYPE(c_ptr) :: p
         p = c_null_ptr
         IF (PRESENT(x)) p = c_loc(x)
      END FUNCTION opt2
      FUNCTION optf(x) RESULT(p)
         REAL, DIMENSION(ims:ime,kms:kme+2,jms:jme), OPTIONAL, INTENT(IN), TARGET :: x
         TYPE(c_ptr) :: p
         p = c_null_ptr
         IF (PRESENT(x)) p = c_loc(x)
      END FUNCTION optf
   END SUBROUTINE RRTMG_LWRAD

!------------------------------------------------------------------
   SUBROUTINE rrtmg_lwinit(p_top, allowed_to_read, ids,ide, jds,jde, kds,kde, ims,ime, jms,jme, kms,kme, &
                           its,ite, jts,jte, kts,kte)
      REAL, INTENT(IN) :: p_top
      LOGICAL, INTENT(IN) :: allowed_to_read
      INTEGER, INTENT(IN) :: ids,ide, jds,jde, kds,kde, ims,ime, jms,jme, kms,kme, its,ite, jts,jte, kts,kte
      CHARACTER(kind=c_char) :: err(errlen)
      CHARACTER(len=32) :: fname
      INTEGER(c_int) :: rc
      ! NLAYERS = kme + nint(p_top*0.01/deltap) - 1 (reference: LW:12861) is fixed inside the library
      ! from p_top and kme; every rank reads RRTMG_LW_DATA itself and uploads to its own GPU.
      IF (.NOT. allowed_to_read) RETURN
      fname = 'RRTMG_LW_DATA'
      IF (STORAGE_SIZE(p_top) == 64) fname = 'RRTMG_LW_DATA_DBL'
      rc = rrtmg_hip_lw_init(TRIM(fname)//c_null_char, INT(STORAGE_SIZE(p_top)/8, c_int), REAL(cp, c_double), &
                             REAL(p_top, c_double), INT(kme, c_int), rrtmg_hip_pick_device(), err)
      IF (rc /= 0) CALL fatal_from_c(err)
   END SUBROUTINE rrtmg_lwinit

END MODULE module_ra_rrtmg_lw
