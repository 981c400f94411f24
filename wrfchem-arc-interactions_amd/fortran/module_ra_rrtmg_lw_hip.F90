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
   PUBLIC :: inirad, o3data, relcalc, reicalc, retab

   ! ice effective radius (um) versus temperature, 180..274 K (numeric table of the reference, LW:11424-11442)
   REAL, SAVE :: retab(95) = (/ &
         5.92779, 6.26422, 6.61973, 6.99539, 7.39234, 7.81177,  &
         8.25496, 8.72323, 9.21800, 9.74075, 10.2930, 10.8765,  &
         11.4929, 12.1440, 12.8317, 13.5581, 14.2319, 15.0351,  &
         15.8799, 16.7674, 17.6986, 18.6744, 19.6955, 20.7623,  &
         21.8757, 23.0364, 24.2452, 25.5034, 26.8125, 27.7895,  &
         28.6450, 29.4167, 30.1088, 30.7306, 31.2943, 31.8151,  &
         32.3077, 32.7870, 33.2657, 33.7540, 34.2601, 34.7892,  &
         35.3442, 35.9255, 36.5316, 37.1602, 37.8078, 38.4720,  &
         39.1508, 39.8442, 40.5552, 41.2912, 42.0635, 42.8876,  &
         43.7863, 44.7853, 45.9170, 47.2165, 48.7221, 50.4710,  &
         52.4980, 54.8315, 57.4898, 60.4785, 63.7898, 65.5604,  &
         71.2885, 75.4113, 79.7368, 84.2351, 88.8833, 93.6658,  &
         98.5739, 103.603, 108.752, 114.025, 119.424, 124.954,  &
         130.630, 136.457, 142.446, 148.608, 154.956, 161.503,  &
         168.262, 175.248, 182.473, 189.952, 197.699, 205.728,  &
         214.055, 222.694, 231.661, 240.971, 250.639 /)

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
      TYPE(c_ptr) :: lradius, iradius
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
      a%lradius = c_null_ptr; a%iradius = c_null_ptr
      IF (is_cammgmp_used) THEN   ! effective radii from the CAM-MG microphysics
         IF (.NOT. (PRESENT(lradius) .AND. PRESENT(iradius))) CALL wrf_error_fatal( &
            'rrtmg (HIP): is_cammgmp_used needs lradius and iradius')
         a%lradius = c_loc(lradius); a%iradius = c_loc(iradius)
      END IF
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

!------------------------------------------------------------------
!  Host-side helpers other radiation schemes import from this module (the reference's module_ra_rrtmg_sw does,
!  SW:9894): ozone climatology INIRAD / O3DATA (LW:12704-12840), effective radii relcalc / reicalc (LW:14403-14491)
!  and the table retab (LW:11424-11442).  Plain Fortran on the host -- their callers are CPU code -- checked bit for
!  bit against the reference routines (tests/test_fortran_lw_helpers.py, tests/golden/wrapper_helpers.npz).
!------------------------------------------------------------------
   SUBROUTINE inirad(o3prof, plev, kts, kte)
      INTEGER, INTENT(IN) :: kts, kte
      REAL, DIMENSION(kts:kte+1), INTENT(INOUT) :: o3prof
      REAL, DIMENSION(kts:kte+2), INTENT(IN) :: plev
      o3prof(kts:kte+1) = 0.
      CALL o3data(o3prof, plev, kts, kte)
   END SUBROUTINE inirad

   SUBROUTINE o3data(o3prof, plev, kts, kte)
      INTEGER, INTENT(IN) :: kts, kte
      REAL, DIMENSION(kts:kte+1), INTENT(INOUT) :: o3prof
      REAL, DIMENSION(kts:kte+2), INTENT(IN) :: plev
      ! summer / winter mixing-ratio profiles and their pressure levels (mb), 31 levels
      REAL, PARAMETER :: o3sum(31) = (/ &
         5.297E-8, 5.852E-8, 6.579E-8, 7.505E-8, 8.577E-8, 9.895E-8,  &
         1.175E-7, 1.399E-7, 1.677E-7, 2.003E-7, 2.571E-7, 3.325E-7,  &
         4.438E-7, 6.255E-7, 8.168E-7, 1.036E-6, 1.366E-6, 1.855E-6,  &
         2.514E-6, 3.240E-6, 4.033E-6, 4.854E-6, 5.517E-6, 6.089E-6,  &
         6.689E-6, 1.106E-5, 1.462E-5, 1.321E-5, 9.856E-6, 5.960E-6,  &
         5.960E-6 /)
      REAL, PARAMETER :: ppsum(31) = (/ &
         955.890, 850.532, 754.599, 667.742, 589.841, 519.421,  &
         455.480, 398.085, 347.171, 301.735, 261.310, 225.360,  &
         193.419, 165.490, 141.032, 120.125, 102.689, 87.829,  &
         75.123, 64.306, 55.086, 47.209, 40.535, 34.795,  &
         29.865, 19.122, 9.277, 4.660, 2.421, 1.294,  &
         0.647 /)
      REAL, PARAMETER :: o3win(31) = (/ &
         4.629E-8, 4.686E-8, 5.017E-8, 5.613E-8, 6.871E-8, 8.751E-8,  &
         1.138E-7, 1.516E-7, 2.161E-7, 3.264E-7, 4.968E-7, 7.338E-7,  &
         1.017E-6, 1.308E-6, 1.625E-6, 2.011E-6, 2.516E-6, 3.130E-6,  &
         3.840E-6, 4.703E-6, 5.486E-6, 6.289E-6, 6.993E-6, 7.494E-6,  &
         8.197E-6, 9.632E-6, 1.113E-5, 1.146E-5, 9.389E-6, 6.135E-6,  &
         6.135E-6 /)
      REAL, PARAMETER :: ppwin(31) = (/ &
         955.747, 841.783, 740.199, 649.538, 568.404, 495.815,  &
         431.069, 373.464, 322.354, 277.190, 237.635, 203.433,  &
         174.070, 148.949, 127.408, 108.915, 93.114, 79.551,  &
         67.940, 58.072, 49.593, 42.318, 36.138, 30.907,  &
         26.362, 16.423, 7.583, 3.620, 1.807, 0.938,  &
         0.469 /)
      REAL :: o3ann(31), ppwrkh(32), pb1, pb2, pt1, pt2
      INTEGER :: k, jj
      ! annual mean on the summer pressure levels
      o3ann(1) = 0.5 * (o3sum(1) + o3win(1))
      DO k = 2, 31
         o3ann(k) = o3win(k-1) + (o3win(k) - o3win(k-1)) / (ppwin(k) - ppwin(k-1)) * (ppsum(k) - ppwin(k-1))
      END DO
      DO k = 2, 31
         o3ann(k) = 0.5 * (o3ann(k) + o3sum(k))
      END DO
      ! half levels of the climatology, then the pressure-weighted mean over each model layer
      ppwrkh(1) = 1100.
      DO k = 2, 31
         ppwrkh(k) = (ppsum(k) + ppsum(k-1)) / 2.
      END DO
      ppwrkh(32) = 0.
      DO k = kts, kte + 1
         DO jj = 1, 31
            pb1 = plev(k) - ppwrkh(jj);     IF (-pb1 >= 0.) pb1 = 0.
            pb2 = plev(k) - ppwrkh(jj+1);   IF (-pb2 >= 0.) pb2 = 0.
            pt1 = plev(k+1) - ppwrkh(jj);   IF (-pt1 >= 0.) pt1 = 0.
            pt2 = plev(k+1) - ppwrkh(jj+1); IF (-pt2 >= 0.) pt2 = 0.
            o3prof(k) = o3prof(k) + (pb2 - pb1 - pt2 + pt1) * o3ann(jj)
         END DO
         o3prof(k) = o3prof(k) / (plev(k) - plev(k+1))
      END DO
   END SUBROUTINE o3data

   SUBROUTINE relcalc(ncol, pcols, pver, t, landfrac, landm, icefrac, rel, snowh)
      INTEGER, INTENT(IN) :: ncol, pcols, pver
      REAL, INTENT(IN) :: landfrac(pcols), icefrac(pcols), snowh(pcols), landm(pcols), t(pcols,pver)
      REAL, INTENT(OUT) :: rel(pcols,pver)
      REAL, PARAMETER :: tmelt = 273.16, rliqocean = 14.0, rliqice = 14.0, rliqland = 8.0
      INTEGER :: i, k
      DO k = 1, pver
         DO i = 1, ncol
            rel(i,k) = rliqland + (rliqocean - rliqland) * MIN(1.0, MAX(0.0, (tmelt - t(i,k)) * 0.05))
            rel(i,k) = rel(i,k) + (rliqocean - rel(i,k)) * MIN(1.0, MAX(0.0, snowh(i) * 10.))
            rel(i,k) = rel(i,k) + (rliqocean - rel(i,k)) * MIN(1.0, MAX(0.0, 1.0 - landm(i)))
            rel(i,k) = rel(i,k) + (rliqice - rel(i,k)) * MIN(1.0, MAX(0.0, icefrac(i)))
         END DO
      END DO
   END SUBROUTINE relcalc

   SUBROUTINE reicalc(ncol, pcols, pver, t, re)
      INTEGER, INTENT(IN) :: ncol, pcols, pver
      REAL, INTENT(IN) :: t(pcols,pver)
      REAL, INTENT(OUT) :: re(pcols,pver)
      REAL :: corr
      INTEGER :: i, k, ix
      DO k = 1, pver
         DO i = 1, ncol
            ix = MIN(MAX(INT(t(i,k) - 179.), 1), 94)
            corr = t(i,k) - INT(t(i,k))
            re(i,k) = retab(ix) * (1. - corr) + retab(ix+1) * corr
         END DO
      END DO
   END SUBROUTINE reicalc

END MODULE module_ra_rrtmg_lw
