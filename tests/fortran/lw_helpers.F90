! Calls the host-side helpers the LW module exports to other radiation schemes (inirad, relcalc, reicalc) on the
! inputs of tests/golden/wrapper_helpers.npz; same file formats as oracle/ref_harness_wrapper.F90.
PROGRAM lw_helpers
   USE module_ra_rrtmg_lw, ONLY : inirad, relcalc, reicalc, retab
   IMPLICIT NONE
   INTEGER :: n, nk, nprof, nlev, ip
   REAL, ALLOCATABLE :: t(:,:), landfrac(:), landm(:), icefrac(:), snowh(:), rel(:,:), rei(:,:), plev(:,:), o3(:,:)
   CHARACTER(len=256) :: fin, fout
   CALL get_command_argument(1, fin)
   CALL get_command_argument(2, fout)
   OPEN(11, file=TRIM(fin), access='stream', form='unformatted', status='old')
   READ(11) n, nk, nprof, nlev
   ALLOCATE(t(n,nk), landfrac(n), landm(n), icefrac(n), snowh(n), rel(n,nk), rei(n,nk), plev(nlev,nprof), o3(nlev-1,nprof))
   READ(11) t, landfrac, landm, icefrac, snowh
   READ(11) plev
   CLOSE(11)
   CALL relcalc(n, n, nk, t, landfrac, landm, icefrac, rel, snowh)
   CALL reicalc(n, n, nk, t, rei)
   DO ip = 1, nprof
      CALL inirad(o3(:,ip), plev(:,ip), 1, nlev-2)
   END DO
   OPEN(12, file=TRIM(fout), access='stream', form='unformatted', status='replace')
   WRITE(12) rel, rei, o3, retab
   CLOSE(12)
END PROGRAM lw_helpers
