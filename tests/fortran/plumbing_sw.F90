! Plumbing driver (BASELINE config 0, 4x4x40): calls rrtmg_swinit and RRTMG_SWRAD
! by keyword the way module_radiation_driver.F:1944-2001 does, on a state read
! from a file written by tests/test_fortran_shim.py, and dumps the outputs.
PROGRAM plumbing_sw
   USE module_ra_rrtmg_sw, ONLY : rrtmg_swrad, rrtmg_swinit
   IMPLICIT NONE
   INTEGER :: ni, nj, nk, clean, icloud, aeropt, ssib, sfphys
   REAL, ALLOCATABLE, DIMENSION(:,:) :: alswvisdir, alswvisdif, alswnirdir, alswnirdif
   REAL, ALLOCATABLE, DIMENSION(:,:,:,:) :: aerod
   REAL, ALLOCATABLE, DIMENSION(:,:,:) :: cldfra3d, qc3d, qr3d, qi3d, qs3d
   INTEGER :: ids,ide,jds,jde,kds,kde, ims,ime,jms,jme,kms,kme, its,ite,jts,jte,kts,kte
   REAL, ALLOCATABLE, DIMENSION(:,:,:) :: t3d,t8w,p3d,p8w,pi3d,rho3d,dz8w,qv3d,rthratensw, re_c,re_i,re_s
   REAL, ALLOCATABLE, DIMENSION(:,:,:) :: tauaer300,tauaer400,tauaer600,tauaer999,gaer300,gaer400,gaer600,gaer999
   REAL, ALLOCATABLE, DIMENSION(:,:,:) :: waer300,waer400,waer600,waer999
   REAL, ALLOCATABLE, DIMENSION(:,:) :: xcoszen,albedo,tsk,xlat,xlong,xland,xice,snow,coszr,gsw,swcf
   REAL, ALLOCATABLE, DIMENSION(:,:) :: swupt,swuptc,swuptcln,swdnt,swdntc,swdntcln,swupb,swupbc,swupbcln
   REAL, ALLOCATABLE, DIMENSION(:,:) :: swdnb,swdnbc,swdnbcln,swddir,swddni,swddif
   REAL, DIMENSION(:,:,:,:), POINTER :: tauaer3d_sw, ssaaer3d_sw, asyaer3d_sw
   CHARACTER(len=256) :: fin, fout
   CALL get_command_argument(1, fin)
   CALL get_command_argument(2, fout)
   OPEN(11, file=TRIM(fin), access='stream', form='unformatted', status='old')
   READ(11) ni, nj, nk, clean
   ssib = clean / 1000            ! header word: clean_atm_diag + 10*icloud + 100*aer_opt + 1000*ssib
   clean = MOD(clean, 1000)
   aeropt = clean / 100
   clean = MOD(clean, 100)
   icloud = clean / 10
   clean = MOD(clean, 10)
   ids=1; ide=ni+1; jds=1; jde=nj+1; kds=1; kde=nk+1
   ims=1; ime=ni; jms=1; jme=nj; kms=1; kme=nk+1
   its=1; ite=ni; jts=1; jte=nj; kts=1; kte=nk
   ALLOCATE(t3d(ims:ime,kms:kme,jms:jme), t8w(ims:ime,kms:kme,jms:jme), p3d(ims:ime,kms:kme,jms:jme))
   ALLOCATE(p8w(ims:ime,kms:kme,jms:jme), pi3d(ims:ime,kms:kme,jms:jme), rho3d(ims:ime,kms:kme,jms:jme))
   ALLOCATE(dz8w(ims:ime,kms:kme,jms:jme), qv3d(ims:ime,kms:kme,jms:jme), rthratensw(ims:ime,kms:kme,jms:jme))
   ALLOCATE(re_c(ims:ime,kms:kme,jms:jme), re_i(ims:ime,kms:kme,jms:jme), re_s(ims:ime,kms:kme,jms:jme))
   ALLOCATE(tauaer300(ims:ime,kms:kme,jms:jme), tauaer400(ims:ime,kms:kme,jms:jme), tauaer600(ims:ime,kms:kme,jms:jme))
   ALLOCATE(tauaer999(ims:ime,kms:kme,jms:jme), gaer300(ims:ime,kms:kme,jms:jme), gaer400(ims:ime,kms:kme,jms:jme))
   ALLOCATE(gaer600(ims:ime,kms:kme,jms:jme), gaer999(ims:ime,kms:kme,jms:jme), waer300(ims:ime,kms:kme,jms:jme))
   ALLOCATE(waer400(ims:ime,kms:kme,jms:jme), waer600(ims:ime,kms:kme,jms:jme), waer999(ims:ime,kms:kme,jms:jme))
   ALLOCATE(xcoszen(ims:ime,jms:jme), albedo(ims:ime,jms:jme), tsk(ims:ime,jms:jme), xlat(ims:ime,jms:jme))
   ALLOCATE(xlong(ims:ime,jms:jme), xland(ims:ime,jms:jme), xice(ims:ime,jms:jme), snow(ims:ime,jms:jme))
   ALLOCATE(coszr(ims:ime,jms:jme), gsw(ims:ime,jms:jme), swcf(ims:ime,jms:jme))
   ALLOCATE(swupt(ims:ime,jms:jme), swuptc(ims:ime,jms:jme), swuptcln(ims:ime,jms:jme), swdnt(ims:ime,jms:jme))
   ALLOCATE(swdntc(ims:ime,jms:jme), swdntcln(ims:ime,jms:jme), swupb(ims:ime,jms:jme), swupbc(ims:ime,jms:jme))
   ALLOCATE(swupbcln(ims:ime,jms:jme), swdnb(ims:ime,jms:jme), swdnbc(ims:ime,jms:jme), swdnbcln(ims:ime,jms:jme))
   ALLOCATE(swddir(ims:ime,jms:jme), swddni(ims:ime,jms:jme), swddif(ims:ime,jms:jme))
   READ(11) xcoszen, albedo, tsk
   READ(11) t3d, t8w, p3d, p8w, pi3d, qv3d
   READ(11) tauaer300, tauaer400, tauaer600, tauaer999, gaer300, gaer400, gaer600, gaer999
   READ(11) waer300, waer400, waer600, waer999
   ALLOCATE(cldfra3d(ims:ime,kms:kme,jms:jme), qc3d(ims:ime,kms:kme,jms:jme), qr3d(ims:ime,kms:kme,jms:jme))
   ALLOCATE(qi3d(ims:ime,kms:kme,jms:jme), qs3d(ims:ime,kms:kme,jms:jme))
   cldfra3d = 0.; qc3d = 0.; qr3d = 0.; qi3d = 0.; qs3d = 0.
   IF (icloud /= 0) READ(11) cldfra3d, qc3d, qr3d, qi3d, qs3d
   ALLOCATE(aerod(ims:ime,kms:kme,jms:jme,6))
   aerod = 0.
   IF (aeropt == 1) READ(11) aerod
   ALLOCATE(alswvisdir(ims:ime,jms:jme), alswvisdif(ims:ime,jms:jme), alswnirdir(ims:ime,jms:jme), alswnirdif(ims:ime,jms:jme))
   alswvisdir = 0.; alswvisdif = 0.; alswnirdir = 0.; alswnirdif = 0.
   sfphys = 2
   CLOSE(11)
   rho3d = 1.; dz8w = 100.; xlat = 0.; xlong = 0.; xland = 1.; xice = 0.; snow = 0.
   IF (ssib == 1) THEN   ! SSiB land surface: spectral albedos derived from the broadband one, water in every 3rd column
      sfphys = 8
      alswvisdir = 0.5 * albedo; alswvisdif = 0.6 * albedo; alswnirdir = 1.4 * albedo; alswnirdif = 1.5 * albedo
      xland(ims:ime:3, :) = 2.
   END IF
   re_c = 0.; re_i = 0.; re_s = 0.
   rthratensw = -7777.; gsw = -7777.; swcf = -7777.; coszr = -7777.
   NULLIFY(tauaer3d_sw, ssaaer3d_sw, asyaer3d_sw)

   CALL rrtmg_swinit(.TRUE., ids,ide, jds,jde, kds,kde, ims,ime, jms,jme, kms,kme, its,ite, jts,jte, kts,kte)

   CALL RRTMG_SWRAD(                                                  &
        RTHRATENSW=rthratensw,                                        &
        SWUPT=swupt, SWUPTC=swuptc, SWUPTCLN=swuptcln,                &
        SWDNT=swdnt, SWDNTC=swdntc, SWDNTCLN=swdntcln,                &
        SWUPB=swupb, SWUPBC=swupbc, SWUPBCLN=swupbcln,                &
        SWDNB=swdnb, SWDNBC=swdnbc, SWDNBCLN=swdnbcln,                &
        SWCF=swcf, GSW=gsw,                                           &
        XTIME=0., GMT=0., XLAT=xlat, XLONG=xlong,                     &
        RADT=30., DEGRAD=0.0174533, DECLIN=0.,                        &
        COSZR=coszr, JULDAY=180, SOLCON=1370.,                        &
        ALBEDO=albedo, t3d=t3d, t8w=t8w, TSK=tsk,                     &
        p3d=p3d, p8w=p8w, pi3d=pi3d, rho3d=rho3d,                     &
        dz8w=dz8w, IS_CAMMGMP_USED=.FALSE., R=287., G=9.81,           &
        RE_CLOUD=re_c, RE_ICE=re_i, RE_SNOW=re_s,                     &
        has_reqc=0, has_reqi=0, has_reqs=0,                           &
        ICLOUD=icloud, WARM_RAIN=.FALSE.,                             &
        CLDFRA3D=cldfra3d, QC3D=qc3d, QR3D=qr3d, QI3D=qi3d, QS3D=qs3d, &
        F_QV=.TRUE., F_QC=.TRUE., F_QR=.TRUE., F_QI=.TRUE., F_QS=.TRUE., F_QG=.FALSE., &
        XLAND=xland, XICE=xice, SNOW=snow,                            &
        QV3D=qv3d, AER_OPT=aeropt, AEROD=aerod, NO_SRC=6, sf_surface_physics=sfphys, &
        ALSWVISDIR=alswvisdir, ALSWVISDIF=alswvisdif, ALSWNIRDIR=alswnirdir, ALSWNIRDIF=alswnirdif, &
        TAUAER300=tauaer300, TAUAER400=tauaer400,                     &
        TAUAER600=tauaer600, TAUAER999=tauaer999,                     &
        GAER300=gaer300, GAER400=gaer400, GAER600=gaer600, GAER999=gaer999, &
        WAER300=waer300, WAER400=waer400, WAER600=waer600, WAER999=waer999, &
        aer_ra_feedback=1, progn=0, clean_atm_diag=clean,             &
        mp_physics=0,                                                 &
        IDS=ids,IDE=ide, JDS=jds,JDE=jde, KDS=kds,KDE=kde,            &
        IMS=ims,IME=ime, JMS=jms,JME=jme, KMS=kms,KME=kme,            &
        its=its,ite=ite, jts=jts,jte=jte, kts=kts,kte=kte,            &
        tauaer3d_sw=tauaer3d_sw, ssaaer3d_sw=ssaaer3d_sw, asyaer3d_sw=asyaer3d_sw, &
        swddir=swddir, swddni=swddni, swddif=swddif,                  &
        xcoszen=xcoszen, julian=180.)

   OPEN(12, file=TRIM(fout), access='stream', form='unformatted', status='replace')
   WRITE(12) rthratensw, coszr, gsw, swcf
   WRITE(12) swupt, swuptc, swuptcln, swdnt, swdntc, swdntcln, swupb, swupbc, swupbcln, swdnb, swdnbc, swdnbcln
   WRITE(12) swddir, swddni, swddif
   CLOSE(12)
   PRINT '(a,3i5)', 'plumbing_sw ok', ni, nj, nk
END PROGRAM plumbing_sw
