!  module_ra_rrtmg_sw_hip.F90 -- drop-in replacement for WRF-Chem v3.9.1's
!  phys/module_ra_rrtmg_sw.F (patched for the clean-sky double call by
!  douglowe/WRFChem-ARC-Interactions).
!
!  Compile this file INSTEAD of phys/module_ra_rrtmg_sw.F and link
!  librrtmg_hip.so.  It keeps the module name and the two procedures the rest
!  of WRF uses --
!     rrtmg_swinit   (reference: module_ra_rrtmg_sw.F:11211-11234)
!     RRTMG_SWRAD    (reference: module_ra_rrtmg_sw.F:9901-9945; called by
!                     keyword from module_radiation_driver.F:1944-2001)
!  -- with the reference's dummy-argument names, types, shapes, INTENTs and
!  OPTIONAL attributes, so module_radiation_driver.F, first_rk_step_part1 and
!  the Registry stay byte-identical.  The bodies only marshal: PRESENT() is
!  resolved here (it cannot cross a C ABI), the whole tile is handed to the
!  device in one call, and a non-zero status becomes wrf_error_fatal.
!
!  This file contains no radiation arithmetic.

MODULE module_ra_rrtmg_sw

   USE, INTRINSIC :: iso_c_binding
   USE module_model_constants, ONLY : cp

   IMPLICIT NONE
   PRIVATE
   PUBLIC :: rrtmg_swrad, rrtmg_swinit

   INTEGER, PARAMETER :: errlen = 256

   ! mirrors struct rrtmg_hip_swrad_args of include/rrtmg_hip.h
   TYPE, BIND(C) :: swrad_args
      INTEGER(c_int) :: ims, ime, jms, jme, kms, kme
      INTEGER(c_int) :: its, ite, jts, jte, kts, kte
      INTEGER(c_int) :: real_bytes, mem
      TYPE(c_ptr)    :: stream
      REAL(c_double) :: solcon
      INTEGER(c_int) :: aer_ra_feedback, clean_atm_diag, icloud, o3input
      TYPE(c_ptr) :: xcoszen, albedo, tsk
      TYPE(c_ptr) :: t3d, t8w, p3d, p8w, pi3d, qv3d, o33d
      TYPE(c_ptr) :: tauaer300, tauaer400, tauaer600, tauaer999
      TYPE(c_ptr) :: gaer300, gaer400, gaer600, gaer999
      TYPE(c_ptr) :: waer300, waer400, waer600, waer999
      TYPE(c_ptr) :: rthratensw
      TYPE(c_ptr) :: coszr, gsw, swcf
      TYPE(c_ptr) :: swupt, swuptc, swuptcln, swdnt, swdntc, swdntcln
      TYPE(c_ptr) :: swupb, swupbc, swupbcln, swdnb, swdnbc, swdnbcln
      TYPE(c_ptr) :: swvisdir, swvisdif, swnirdir, swnirdif
      TYPE(c_ptr) :: swddir, swddni, swddif
      TYPE(c_ptr) :: swupflx, swupflxc, swupflxcln, swdnflx, swdnflxc, swdnflxcln
      TYPE(c_ptr) :: cldfra3d, qc3d, qr3d, qi3d, qs3d
      TYPE(c_ptr) :: re_cloud, re_ice, re_snow
      TYPE(c_ptr) :: qndrop3d
      TYPE(c_ptr) :: xland, xice, snow
      INTEGER(c_int) :: has_reqc, has_reqi, has_reqs
      INTEGER(c_int) :: warm_rain, progn
      INTEGER(c_int) :: is_cammgmp_used, f_ice_phy, mp_physics
      REAL(c_double) :: g
      TYPE(c_ptr) :: tauaer3d_sw, ssaaer3d_sw, asyaer3d_sw
      TYPE(c_ptr) :: aerod
      TYPE(c_ptr) :: alswvisdir, alswvisdif, alswnirdir, alswnirdif
      TYPE(c_ptr) :: lradius, iradius
   END TYPE swrad_args

   INTERFACE
      INTEGER(c_int) FUNCTION rrtmg_hip_sw_init(path, real_bytes, cpdair, device, err) BIND(C, name='rrtmg_hip_sw_init')
         IMPORT :: c_int, c_double, c_char
         CHARACTER(kind=c_char), INTENT(IN) :: path(*)
         INTEGER(c_int), VALUE :: real_bytes, device
         REAL(c_double), VALUE :: cpdair
         CHARACTER(kind=c_char), INTENT(INOUT) :: err(*)
      END FUNCTION rrtmg_hip_sw_init
      INTEGER(c_int) FUNCTION rrtmg_hip_swrad(args, err) BIND(C, name='rrtmg_hip_swrad')
         IMPORT :: c_int, c_char, swrad_args
         TYPE(swrad_args), INTENT(IN) :: args
         CHARACTER(kind=c_char), INTENT(INOUT) :: err(*)
      END FUNCTION rrtmg_hip_swrad
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
   SUBROUTINE RRTMG_SWRAD(                                        &
                       rthratensw,                                &
                       swupt, swuptc, swuptcln, swdnt, swdntc, swdntcln, &
                       swupb, swupbc, swupbcln, swdnb, swdnbc, swdnbcln, &
                       swcf, gsw,                                 &
                       xtime, gmt, xlat, xlong,                   &
                       radt, degrad, declin,                      &
                       coszr, julday, solcon,                     &
                       albedo, t3d, t8w, tsk,                     &
                       p3d, p8w, pi3d, rho3d,                     &
                       dz8w, cldfra3d, lradius, iradius,          &
                       is_cammgmp_used, r, g,                     &
                       re_cloud,re_ice,re_snow,                   &
                       has_reqc,has_reqi,has_reqs,                &
                       icloud, warm_rain,                         &
                       f_ice_phy, f_rain_phy,                     &
                       xland, xice, snow,                         &
                       qv3d, qc3d, qr3d,                          &
                       qi3d, qs3d, qg3d,                          &
                       o3input, o33d,                             &
                       aer_opt, aerod, no_src,                    &
                       alswvisdir, alswvisdif,                    &
                       alswnirdir, alswnirdif,                    &
                       swvisdir, swvisdif,                        &
                       swnirdir, swnirdif,                        &
                       sf_surface_physics,                        &
                       f_qv, f_qc, f_qr, f_qi, f_qs, f_qg,        &
                       tauaer300,tauaer400,tauaer600,tauaer999,   &
                       gaer300,gaer400,gaer600,gaer999,           &
                       waer300,waer400,waer600,waer999,           &
                       aer_ra_feedback,                           &
                       progn,clean_atm_diag,                      &
                       qndrop3d,f_qndrop,                         &
                       mp_physics,                                &
                       ids,ide, jds,jde, kds,kde,                 &
                       ims,ime, jms,jme, kms,kme,                 &
                       its,ite, jts,jte, kts,kte,                 &
                       swupflx, swupflxc, swupflxcln,             &
                       swdnflx, swdnflxc, swdnflxcln,             &
                       tauaer3d_sw,ssaaer3d_sw,asyaer3d_sw,       &
                       swddir, swddni, swddif,                    &
                       xcoszen,julian                             &
                                                                  )
      LOGICAL, INTENT(IN) :: warm_rain, is_cammgmp_used
      INTEGER, INTENT(IN) :: ids,ide, jds,jde, kds,kde, ims,ime, jms,jme, kms,kme, its,ite, jts,jte, kts,kte
      INTEGER, INTENT(IN) :: icloud, mp_physics
      REAL, DIMENSION(ims:ime,kms:kme,jms:jme), INTENT(IN), TARGET :: dz8w, t3d, t8w, p3d, p8w, pi3d, rho3d
      REAL, DIMENSION(ims:ime,kms:kme,jms:jme), INTENT(INOUT), TARGET :: rthratensw
      REAL, DIMENSION(ims:ime,jms:jme), INTENT(INOUT), TARGET :: gsw, swcf, coszr
      INTEGER, INTENT(IN) :: julday
      REAL, INTENT(IN) :: radt, degrad, xtime, declin, solcon, gmt
      REAL, DIMENSION(ims:ime,jms:jme), INTENT(IN), TARGET :: xlat, xlong, xland, xice, snow, tsk, albedo
      REAL, DIMENSION(ims:ime,jms:jme), OPTIONAL, INTENT(IN), TARGET :: alswvisdir, alswvisdif, alswnirdir, alswnirdif
      REAL, DIMENSION(ims:ime,jms:jme), OPTIONAL, INTENT(OUT), TARGET :: swvisdir, swvisdif, swnirdir, swnirdif
      INTEGER, INTENT(IN) :: sf_surface_physics
      REAL, DIMENSION(ims:ime,jms:jme), INTENT(OUT), TARGET :: swddir, swddni, swddif
      REAL, OPTIONAL, INTENT(IN) :: julian
      REAL, DIMENSION(ims:ime,jms:jme), INTENT(IN), TARGET :: xcoszen
      REAL, DIMENSION(:,:,:,:), POINTER :: tauaer3d_sw, ssaaer3d_sw, asyaer3d_sw
      REAL, INTENT(IN) :: r, g
      REAL, DIMENSION(ims:ime,kms:kme,jms:jme), OPTIONAL, INTENT(IN), TARGET :: &
           cldfra3d, lradius, iradius, qv3d, qc3d, qr3d, qi3d, qs3d, qg3d, qndrop3d
      REAL, DIMENSION(ims:ime,kms:kme,jms:jme), INTENT(IN), TARGET :: re_cloud, re_ice, re_snow
      INTEGER, INTENT(IN) :: has_reqc, has_reqi, has_reqs
      REAL, DIMENSION(ims:ime,kms:kme,jms:jme), OPTIONAL, INTENT(IN) :: f_ice_phy, f_rain_phy
      LOGICAL, OPTIONAL, INTENT(IN) :: f_qv, f_qc, f_qr, f_qi, f_qs, f_qg, f_qndrop
      REAL, DIMENSION(ims:ime,kms:kme,jms:jme), OPTIONAL, INTENT(IN), TARGET :: &
           tauaer300, tauaer400, tauaer600, tauaer999, gaer300, gaer400, gaer600, gaer999, &
           waer300, waer400, waer600, waer999
      INTEGER, INTENT(IN), OPTIONAL :: aer_ra_feedback, progn, clean_atm_diag
      REAL, DIMENSION(ims:ime,kms:kme,jms:jme), OPTIONAL, INTENT(IN), TARGET :: o33d
      INTEGER, OPTIONAL, INTENT(IN) :: o3input
      INTEGER, INTENT(IN) :: no_src
      REAL, DIMENSION(ims:ime,kms:kme,jms:jme,1:no_src), OPTIONAL, INTENT(IN), TARGET :: aerod
      INTEGER, OPTIONAL, INTENT(IN) :: aer_opt
      REAL, DIMENSION(ims:ime,jms:jme), OPTIONAL, INTENT(INOUT), TARGET :: &
           swupt, swuptc, swuptcln, swdnt, swdntc, swdntcln, swupb, swupbc, swupbcln, swdnb, swdnbc, swdnbcln
      REAL, DIMENSION(ims:ime,kms:kme+2,jms:jme), OPTIONAL, INTENT(OUT), TARGET :: &
           swupflx, swupflxc, swupflxcln, swdnflx, swdnflxc, swdnflxcln

      TYPE(swrad_args) :: a
      CHARACTER(kind=c_char) :: err(errlen)
      INTEGER(c_int) :: rc

      a%ims = ims; a%ime = ime; a%jms = jms; a%jme = jme; a%kms = kms; a%kme = kme
      a%its = its; a%ite = ite; a%jts = jts; a%jte = jte; a%kts = kts; a%kte = kte
      a%real_bytes = STORAGE_SIZE(solcon) / 8
      a%mem = 0
      a%stream = c_null_ptr
      a%solcon = REAL(solcon, c_double)
      a%aer_ra_feedback = 0
      IF (PRESENT(aer_ra_feedback)) a%aer_ra_feedback = aer_ra_feedback
      a%clean_atm_diag = 0                       ! absent => 0 (reference: SW:10308-10312)
      IF (PRESENT(clean_atm_diag)) a%clean_atm_diag = clean_atm_diag
      a%icloud = icloud
      a%o3input = 0
      IF (PRESENT(o3input)) a%o3input = o3input

      ! band-resolved aerosol input of aer_opt = 2/3 (SW:10982-10991): (ims:ime,kms:kme,jms:jme,1:nbndsw), contiguous
      a%tauaer3d_sw = c_null_ptr; a%ssaaer3d_sw = c_null_ptr; a%asyaer3d_sw = c_null_ptr
      IF (ASSOCIATED(tauaer3d_sw)) THEN
         IF (.NOT. (ASSOCIATED(ssaaer3d_sw) .AND. ASSOCIATED(asyaer3d_sw))) CALL wrf_error_fatal( &
            'rrtmg_swrad (HIP): tauaer3d_sw is associated but ssaaer3d_sw / asyaer3d_sw are not')
         IF (SIZE(tauaer3d_sw,1) /= ime-ims+1 .OR. SIZE(tauaer3d_sw,2) /= kme-kms+1 .OR. &
             SIZE(tauaer3d_sw,3) /= jme-jms+1 .OR. SIZE(tauaer3d_sw,4) /= 14 .OR. &
             .NOT. IS_CONTIGUOUS(tauaer3d_sw) .OR. .NOT. IS_CONTIGUOUS(ssaaer3d_sw) .OR. &
             .NOT. IS_CONTIGUOUS(asyaer3d_sw)) CALL wrf_error_fatal( &
            'rrtmg_swrad (HIP): tauaer3d_sw must be a contiguous (ims:ime,kms:kme,jms:jme,14) array')
         a%tauaer3d_sw = c_loc(tauaer3d_sw); a%ssaaer3d_sw = c_loc(ssaaer3d_sw); a%asyaer3d_sw = c_loc(asyaer3d_sw)
      END IF
      ! aer_opt = 1: optical depth of the six ECMWF aerosol types (SW:11083-11100), rrtmg_sw then runs with iaer = 6
      a%aerod = c_null_ptr
      IF (PRESENT(aer_opt)) THEN
         IF (aer_opt == 1) THEN
            IF (.NOT. PRESENT(aerod)) CALL wrf_error_fatal('rrtmg_swrad (HIP): aer_opt=1 needs aerod')
            IF (no_src < 6) CALL wrf_error_fatal('rrtmg_swrad (HIP): aer_opt=1 needs six aerosol types in aerod')
            a%aerod = c_loc(aerod)
         END IF
      END IF
      ! SSiB land surface: spectral albedos where xland < 1.5 (SW:10692-10703)
      a%alswvisdir = c_null_ptr; a%alswvisdif = c_null_ptr; a%alswnirdir = c_null_ptr; a%alswnirdif = c_null_ptr
      IF (sf_surface_physics == 8) THEN
         IF (.NOT. (PRESENT(alswvisdir) .AND. PRESENT(alswvisdif) .AND. PRESENT(alswnirdir) .AND. PRESENT(alswnirdif))) &
            CALL wrf_error_fatal('rrtmg_swrad (HIP): sf_surface_physics=8 needs alswvisdir, alswvisdif, alswnirdir, alswnirdif')
         a%alswvisdir = c_loc(alswvisdir); a%alswvisdif = c_loc(alswvisdif)
         a%alswnirdir = c_loc(alswnirdir); a%alswnirdif = c_loc(alswnirdif)
      END IF
      IF (.NOT. PRESENT(qv3d)) CALL wrf_error_fatal('rrtmg_swrad (HIP): qv3d is required')

      a%xcoszen = c_loc(xcoszen); a%albedo = c_loc(albedo); a%tsk = c_loc(tsk)
      a%t3d = c_loc(t3d); a%t8w = c_loc(t8w); a%p3d = c_loc(p3d); a%p8w = c_loc(p8w); a%pi3d = c_loc(pi3d)
      a%qv3d = c_loc(qv3d)
      a%o33d = c_null_ptr
      IF (PRESENT(o33d)) a%o33d = c_loc(o33d)
      a%tauaer300 = opt3(tauaer300); a%tauaer400 = opt3(tauaer400); a%tauaer600 = opt3(tauaer600)
      a%tauaer999 = opt3(tauaer999)
      a%gaer300 = opt3(gaer300); a%gaer400 = opt3(gaer400); a%gaer600 = opt3(gaer600); a%gaer999 = opt3(gaer999)
      a%waer300 = opt3(waer300); a%waer400 = opt3(waer400); a%waer600 = opt3(waer600); a%waer999 = opt3(waer999)
      a%rthratensw = c_loc(rthratensw)
      a%coszr = c_loc(coszr); a%gsw = c_loc(gsw); a%swcf = c_loc(swcf)
      a%swupt = opt2(swupt); a%swuptc = opt2(swuptc); a%swuptcln = opt2(swuptcln)
      a%swdnt = opt2(swdnt); a%swdntc = opt2(swdntc); a%swdntcln = opt2(swdntcln)
      a%swupb = opt2(swupb); a%swupbc = opt2(swupbc); a%swupbcln = opt2(swupbcln)
      a%swdnb = opt2(swdnb); a%swdnbc = opt2(swdnbc); a%swdnbcln = opt2(swdnbcln)
      a%swvisdir = opt2(swvisdir); a%swvisdif = opt2(swvisdif)
      a%swnirdir = opt2(swnirdir); a%swnirdif = opt2(swnirdif)
      a%swddir = c_loc(swddir); a%swddni = c_loc(swddni); a%swddif = c_loc(swddif)
      a%swupflx = optf(swupflx); a%swupflxc = optf(swupflxc); a%swupflxcln = optf(swupflxcln)
      a%swdnflx = optf(swdnflx); a%swdnflxc = optf(swdnflxc); a%swdnflxcln = optf(swdnflxcln)

      ! clouds (icloud /= 0): a hydrometeor array is handed over only when it is PRESENT and its F_Q* flag is set,
      ! which is exactly when the reference reads it (SW:10371-10435)
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
      a%lradius = c_null_ptr; a%iradius = c_null_ptr
      IF (is_cammgmp_used) THEN   ! effective radii from the CAM-MG microphysics
         IF (.NOT. (PRESENT(lradius) .AND. PRESENT(iradius))) CALL wrf_error_fatal( &
            'rrtmg (HIP): is_cammgmp_used needs lradius and iradius')
         a%lradius = c_loc(lradius); a%iradius = c_loc(iradius)
      END IF
      ! the Ferrier-type split of snow into ice (SW:10438-10450) needs F_ICE_PHY with F_QC, F_QS and no F_QI
      a%f_ice_phy = 0
      IF (PRESENT(f_ice_phy) .AND. PRESENT(f_qc) .AND. PRESENT(f_qi) .AND. PRESENT(f_qs)) THEN
         IF (f_qc .AND. .NOT. f_qi .AND. f_qs) a%f_ice_phy = 1
      END IF
      a%mp_physics = mp_physics
      a%g = REAL(g, c_double)

      rc = rrtmg_hip_swrad(a, err)
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
   END SUBROUTINE RRTMG_SWRAD

!------------------------------------------------------------------
   SUBROUTINE rrtmg_swinit(allowed_to_read, ids,ide, jds,jde, kds,kde, ims,ime, jms,jme, kms,kme, its,ite, jts,jte, kts,kte)
      LOGICAL, INTENT(IN) :: allowed_to_read
      INTEGER, INTENT(IN) :: ids,ide, jds,jde, kds,kde, ims,ime, jms,jme, kms,kme, its,ite, jts,jte, kts,kte
      CHARACTER(kind=c_char) :: err(errlen)
      CHARACTER(len=32) :: fname
      INTEGER(c_int) :: rc
      REAL :: one
      ! The reference reads RRTMG_SW_DATA on the monitor rank and broadcasts (SW:11246-11288);
      ! here every rank reads the file itself and uploads to its own GPU.
      IF (.NOT. allowed_to_read) RETURN
      one = 1.0
      fname = 'RRTMG_SW_DATA'
      IF (STORAGE_SIZE(one) == 64) fname = 'RRTMG_SW_DATA_DBL'
      rc = rrtmg_hip_sw_init(TRIM(fname)//c_null_char, INT(STORAGE_SIZE(one)/8, c_int), REAL(cp, c_double), &
                             rrtmg_hip_pick_device(), err)
      IF (rc /= 0) CALL fatal_from_c(err)
   END SUBROUTINE rrtmg_swinit

END MODULE module_ra_rrtmg_sw
