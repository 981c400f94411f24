! Plumbing driver for the optical-averaging shim: MOSAIC tracers in, the registry's aerosol optical arrays out,
! called the way an optical_driver would (chem array + p_* species numbers).
PROGRAM plumbing_optics
   USE module_optical_averaging_hip
   IMPLICIT NONE
   INTEGER :: ni, nj, nk, nbin, num_chem, opt, b
   INTEGER :: ids,ide,jds,jde,kds,kde, ims,ime,jms,jme,kms,kme, its,ite,jts,jte,kts,kte
   INTEGER, ALLOCATABLE :: idx(:,:), idx_cw(:,:)
   REAL, ALLOCATABLE, DIMENSION(:,:,:,:) :: chem, sw, lw
   REAL, ALLOCATABLE, DIMENSION(:,:,:,:,:) :: leg      ! (i,k,j,4 wavelengths, l2..l7)
   REAL, ALLOCATABLE, DIMENSION(:,:,:) :: alt, dz8w
   CHARACTER(len=256) :: fin, fout
   CALL get_command_argument(1, fin)
   CALL get_command_argument(2, fout)
   OPEN(11, file=TRIM(fin), access='stream', form='unformatted', status='old')
   READ(11) ni, nj, nk, nbin, num_chem, opt
   ids=1; ide=ni+1; jds=1; jde=nj+1; kds=1; kde=nk+1
   ims=1; ime=ni; jms=1; jme=nj; kms=1; kme=nk+1
   its=1; ite=ni; jts=1; jte=nj; kts=1; kte=nk
   ALLOCATE(idx(aer_nspec, nbin), idx_cw(aer_nspec, nbin))
   ALLOCATE(chem(ims:ime,kms:kme,jms:jme,num_chem), alt(ims:ime,kms:kme,jms:jme), dz8w(ims:ime,kms:kme,jms:jme))
   ALLOCATE(sw(ims:ime,kms:kme,jms:jme,12), lw(ims:ime,kms:kme,jms:jme,16), leg(ims:ime,kms:kme,jms:jme,4,6))
   READ(11) idx
   READ(11) chem
   READ(11) alt, dz8w
   CLOSE(11)
   sw = -7777.; lw = -7777.; leg = -7777.
   idx_cw = 0          ! no cloud-borne tracers: the optional argument is passed and must change nothing

   CALL optical_averaging_hip_init(nbin, 0.0390625, 10.0)
   CALL optical_averaging_hip(nbin, opt, num_chem, idx, chem, dz8w, alt,                                  &
        tauaer1=sw(:,:,:,1), tauaer2=sw(:,:,:,2), tauaer3=sw(:,:,:,3), tauaer4=sw(:,:,:,4),               &
        gaer1=sw(:,:,:,5), gaer2=sw(:,:,:,6), gaer3=sw(:,:,:,7), gaer4=sw(:,:,:,8),                       &
        waer1=sw(:,:,:,9), waer2=sw(:,:,:,10), waer3=sw(:,:,:,11), waer4=sw(:,:,:,12),                    &
        tauaerlw1=lw(:,:,:,1), tauaerlw2=lw(:,:,:,2), tauaerlw3=lw(:,:,:,3), tauaerlw4=lw(:,:,:,4),       &
        tauaerlw5=lw(:,:,:,5), tauaerlw6=lw(:,:,:,6), tauaerlw7=lw(:,:,:,7), tauaerlw8=lw(:,:,:,8),       &
        tauaerlw9=lw(:,:,:,9), tauaerlw10=lw(:,:,:,10), tauaerlw11=lw(:,:,:,11), tauaerlw12=lw(:,:,:,12), &
        tauaerlw13=lw(:,:,:,13), tauaerlw14=lw(:,:,:,14), tauaerlw15=lw(:,:,:,15), tauaerlw16=lw(:,:,:,16), &
        ids=ids,ide=ide, jds=jds,jde=jde, kds=kds,kde=kde,                                                &
        ims=ims,ime=ime, jms=jms,jme=jme, kms=kms,kme=kme,                                                &
        its=its,ite=ite, jts=jts,jte=jte, kts=kts,kte=kte, idx_cw=idx_cw,                                 &
        l2aer=leg(:,:,:,:,1), l3aer=leg(:,:,:,:,2), l4aer=leg(:,:,:,:,3), l5aer=leg(:,:,:,:,4),           &
        l6aer=leg(:,:,:,:,5), l7aer=leg(:,:,:,:,6))

   OPEN(12, file=TRIM(fout), access='stream', form='unformatted', status='replace')
   WRITE(12) sw, lw, leg
   CLOSE(12)
   PRINT '(a,4i5)', 'plumbing_optics ok', ni, nj, nk, nbin
END PROGRAM plumbing_optics
