! oracle/ref_harness_wrapper.F90 -- TEST INFRASTRUCTURE.
! Drives three self-contained helper routines of the reference's LW wrapper module (compiled from
! module_ra_rrtmg_lw.F: retab :11424-11442, INIRAD / O3DATA :12704-12840, relcalc / reicalc :14403-14491 by
! build_ref.sh into module wrapper_ref): effective radii on n x nk points, ozone climatology on nprof profiles.
! in : n, nk, nprof, nlev ; t(n,nk), landfrac(n), landm(n), icefrac(n), snowh(n) ; plev(nlev, nprof)
! out: rel(n,nk), rei(n,nk), o3prof(nlev-1, nprof)          (stream, fp64; nlev = kte-kts+3)
program ref_harness_wrapper
  use wrapper_ref
  implicit none
  integer :: n, nk, nprof, nlev, u, v, ip
  real(8), allocatable :: t(:,:), landfrac(:), landm(:), icefrac(:), snowh(:), rel(:,:), rei(:,:)
  real(8), allocatable :: plev(:,:), o3(:,:)
  character(len=512) :: fin, fout
  call get_command_argument(1, fin)
  call get_command_argument(2, fout)
  open(newunit=u, file=trim(fin), access='stream', form='unformatted', status='old')
  read(u) n, nk, nprof, nlev
  allocate(t(n,nk), landfrac(n), landm(n), icefrac(n), snowh(n), rel(n,nk), rei(n,nk))
  allocate(plev(nlev,nprof), o3(nlev-1,nprof))
  read(u) t, landfrac, landm, icefrac, snowh
  read(u) plev
  close(u)
  call relcalc(n, n, nk, t, landfrac, landm, icefrac, rel, snowh)
  call reicalc(n, n, nk, t, rei)
  do ip = 1, nprof
     call inirad(o3(:,ip), plev(:,ip), 1, nlev-2)
  end do
  open(newunit=v, file=trim(fout), access='stream', form='unformatted', status='replace')
  write(v) rel, rei, o3
  close(v)
end program ref_harness_wrapper
