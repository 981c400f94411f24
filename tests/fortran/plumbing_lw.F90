! Plumbing driver for the longwave shim (BASELINE config 0, 4x4x40): calls rrtmg_lwinit and
! RRTMG_LWRAD by keyword the way module_radiation_driver.F:1528-1577 does.
PROGRAM plumbing_lw
   USE module_ra_rrtmg_lw, ONLY : rrtmg_lwrad, rrtmg_lwinit
   IMPLICIT NONE
   INTEGER :: ni, nj, nk, clean, icloud
   REAL, ALLOCATABLE, DIMENSION(:,:,:) :: cldfra3d, qc3d, qr3d, qi3d, qs3d
   INTEGER :: ids,ide,jds,jde,kds,kde, ims,ime,jms,jme,kms,kme, its,ite,jts,jte,kts,kte
   REAL :: p_top
   REAL, ALLOCATABLE, DIMENSION(:,:,:) :: t3d,t8w,p3d,p8w,pi3d,rho3d,dz8w,qv3d,rthratenlw, re_c,re_i,re_s
   REAL, ALLOCATABLE, DIMENSION(:,:,:,:) :: tl
   REAL, ALLOCATABLE, DIMENSION(:,:) :: emiss,tsk,xland,xice,snow,glw,olr,lwcf
   REAL, ALLOCATABLE, DIMENSION(:,:) :: lwupt,lwuptc,lwuptcln,lwdnt,lwdntc,lwdntcln,lwupb,lwupbc,lwupbcln
   REAL, ALLOCATABLE, DIMENSION(:,:) :: lwdnb,lwdnbc,lwdnbcln
   CHARACTER(len=256) :: fin, fout
   CALL get_command_argument(1, fin)
   CALL get_command_argument(2, fout)
   OPEN(11, file=TRIM(fin), access='stream', form='unformatted', status='old')
   READ(11) ni, nj, nk, clean
   icloud = clean / 10            ! header word: clean_atm_diag + 10*icloud
   clean = MOD(clean, 10)
   READ(11) p_top
   ids=1; ide=ni+1; jds=1; jde=nj+1; kds=1; kde=nk+1
   ims=1; ime=ni; jms=1; jme=nj; kms=1; kme=nk+1
   its=1; ite=ni; jts=1; jte=nj; kts=1; kte=nk
   ALLOCATE(t3d(ims:ime,kms:kme,jms:jme), t8w(ims:ime,kms:kme,jms:jme), p3d(ims:ime,kms:kme,jms:jme))
   ALLOCATE(p8w(ims:ime,kms:kme,jms:jme), pi3d(ims:ime,kms:kme,jms:jme), rho3d(ims:ime,kms:kme,jms:jme))
   ALLOCATE(dz8w(ims:ime,kms:kme,jms:jme), qv3d(ims:ime,kms:kme,jms:jme), rthratenlw(ims:ime,kms:kme,jms:jme))
   ALLOCATE(re_c(ims:ime,kms:kme,jms:jme), re_i(ims:ime,kms:kme,jms:jme), re_s(ims:ime,kms:kme,jms:jme))
   ALLOCATE(tl(ims:ime,kms:kme,jms:jme,16))
   ALLOCATE(emiss(ims:ime,jms:jme), tsk(ims:ime,jms:jme), xland(ims:ime,jms:jme), xice(ims:ime,jms:jme))
   ALLOCATE(snow(ims:ime,jms:jme), glw(ims:ime,jms:jme), olr(ims:ime,jms:jme), lwcf(ims:ime,jms:jme))
   ALLOCATE(lwupt(ims:ime,jms:jme), lwuptc(ims:ime,jms:jme), lwuptcln(ims:ime,jms:jme), lwdnt(ims:ime,jms:jme))
   ALLOCATE(lwdntc(ims:ime,jms:jme), lwdntcln(ims:ime,jms:jme), lwupb(ims:ime,jms:jme), lwupbc(ims:ime,jms:jme))
   ALLOCATE(lwupbcln(ims:ime,jms:jme), lwdnb(ims:ime,jms:jme), lwdnbc(ims:ime,jms:jme), lwdnbcln(ims:ime,jms:jme))
   READ(11) emiss, tsk
   READ(11) t3d, t8w, p3d, p8w, pi3d, qv3d
   READ(11) tl
   ALLOCATE(cldfra3d(ims:ime,kms:kme,jms:jme), qc3d(ims:ime,kms:kme,jms:jme), qr3d(ims:ime,kms:kme,jms:jme))
   ALLOCATE(qi3d(ims:ime,kms:kme,jms:jme), qs3d(ims:ime,kms:kme,jms:jme))
   cldfra3d = 0.; qc3d = 0.; qr3d = 0.; qi3d = 0.; qs3d = 0.
   IF (icloud /= 0) READ(11) cldfra3d, qc3d, qr3d, qi3d, qs3d
   CLOSE(11)
   rho3d = 1.; dz8w = 100.; xland = 1.; xice = 0.; snow = 0.; re_c = 0.; re_i = 0.; re_s = 0.
   rthratenlw = -7777.; glw = -7777.; olr = -7777.; lwcf = -7777.

   CALL rrtmg_lwinit(p_top, .TRUE., ids,ide, jds,jde, kds,kde, ims,ime, jms,jme, kms,kme, its,ite, jts,jte, kts,kte)

   CALL RRTMG_LWRAD(                                                  &
        RTHRATENLW=rthratenlw,                                        &
        LWUPT=lwupt, LWUPTC=lwuptc, LWUPTCLN=lwuptcln,                &
        LWDNT=lwdnt, LWDNTC=lwdntc, LWDNTCLN=lwdntcln,                &
        LWUPB=lwupb, LWUPBC=lwupbc, LWUPBCLN=lwupbcln,                &
        LWDNB=lwdnb, LWDNBC=lwdnbc, LWDNBCLN=lwdnbcln,                &
        GLW=glw, OLR=olr, LWCF=lwcf, EMISS=emiss,                     &
        p8w=p8w, p3d=p3d, pi3d=pi3d,                                  &
        dz8w=dz8w, tsk=tsk, t3d=t3d, t8w=t8w, rho3d=rho3d, R=287., G=9.81, &
        ICLOUD=icloud, WARM_RAIN=.FALSE.,                             &
        CLDFRA3D=cldfra3d, QC3D=qc3d, QR3D=qr3d, QI3D=qi3d, QS3D=qs3d, &
        F_QV=.TRUE., F_QC=.TRUE., F_QR=.TRUE., F_QI=.TRUE., F_QS=.TRUE., F_QG=.FALSE., &
        IS_CAMMGMP_USED=.FALSE.,                                      &
        XLAND=xland, XICE=xice, SNOW=snow,                            &
        QV3D=qv3d,                                                    &
        RE_CLOUD=re_c, RE_ICE=re_i, RE_SNOW=re_s,                     &
        has_reqc=0, has_reqi=0, has_reqs=0,                           &
        tauaerlw1=tl(:,:,:,1), tauaerlw2=tl(:,:,:,2), tauaerlw3=tl(:,:,:,3), tauaerlw4=tl(:,:,:,4),     &
        tauaerlw5=tl(:,:,:,5), tauaerlw6=tl(:,:,:,6), tauaerlw7=tl(:,:,:,7), tauaerlw8=tl(:,:,:,8),     &
        tauaerlw9=tl(:,:,:,9), tauaerlw10=tl(:,:,:,10), tauaerlw11=tl(:,:,:,11), tauaerlw12=tl(:,:,:,12), &
        tauaerlw13=tl(:,:,:,13), tauaerlw14=tl(:,:,:,14), tauaerlw15=tl(:,:,:,15), tauaerlw16=tl(:,:,:,16), &
        aer_ra_feedback=1, progn=0, clean_atm_diag=clean,             &
        YR=2026, JULIAN=180., mp_physics=0,                           &
        IDS=ids,IDE=ide, JDS=jds,JDE=jde, KDS=kds,KDE=kde,            &
        IMS=ims,IME=ime, JMS=jms,JME=jme, KMS=kms,KME=kme,            &
        its=its,ite=ite, jts=jts,jte=jte, kts=kts,kte=kte)

   OPEN(12, file=TRIM(fout), access='stream', form='unformatted', status='replace')
   WRITE(12) rthratenlw, glw, olr, lwcf
   WRITE(12) lwupt, lwuptc, lwuptcln, lwdnt, lwdntc, lwdntcln, lwupb, lwupbc, lwupbcln, lwdnb, lwdnbc, lwdnbcln
   CLOSE(12)
   PRINT '(a,3i5)', 'plumbing_lw ok', ni, nj, nk
END PROGRAM plumbing_lw
