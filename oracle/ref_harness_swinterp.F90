! oracle/ref_harness_swinterp.F90 -- TEST INFRASTRUCTURE.
! Drives the reference's update_swinterp_parameters / interp_sw_radiation (compiled from
! module_radiation_driver.F:2668-2795 by build_ref.sh into module swinterp_ref) on a stream of points:
!   in : n ; then per radiation step s = 1..nstep: coszen(n) coszen_loc(n) swddir(n) swdown(n)
!        and per interpolation step: coszen_loc(n) albedo(n)
! File format (unformatted stream, fp64): n, nstep, then for every step the four update inputs followed by the
! two inputs of one interpolation.  Output: per step bb Bx gg Gx coszen_ref swddir_ref swdown_ref, then
! swdown swddir swddni swddif gsw of the interpolation.
program ref_harness_swinterp
  use swinterp_ref
  implicit none
  integer :: n, nstep, s, u, v
  real(8), allocatable :: coszen(:,:), coszen_loc(:,:), swddir(:,:), swdown(:,:), albedo(:,:), cl2(:,:)
  real(8), allocatable :: swddir_ref(:,:), bb(:,:), Bx(:,:), swdown_ref(:,:), gg(:,:), Gx(:,:), coszen_ref(:,:)
  real(8), allocatable :: o_down(:,:), o_dir(:,:), o_dni(:,:), o_dif(:,:), o_gsw(:,:)
  character(len=512) :: fin, fout
  call get_command_argument(1, fin)
  call get_command_argument(2, fout)
  open(newunit=u, file=trim(fin), access='stream', form='unformatted', status='old')
  open(newunit=v, file=trim(fout), access='stream', form='unformatted', status='replace')
  read(u) n, nstep
  allocate(coszen(n,1), coszen_loc(n,1), swddir(n,1), swdown(n,1), albedo(n,1), cl2(n,1))
  allocate(swddir_ref(n,1), bb(n,1), Bx(n,1), swdown_ref(n,1), gg(n,1), Gx(n,1), coszen_ref(n,1))
  allocate(o_down(n,1), o_dir(n,1), o_dni(n,1), o_dif(n,1), o_gsw(n,1))
  swddir_ref = 0; bb = 0; Bx = 0; swdown_ref = 0; gg = 0; Gx = 0; coszen_ref = 0
  do s = 1, nstep
     read(u) coszen, coszen_loc, swddir, swdown, cl2, albedo
     call update_swinterp_parameters(1,n,1,1,1,n,1,1, coszen,coszen_loc,swddir,swdown, &
                                     swddir_ref,bb,Bx,swdown_ref,gg,Gx,coszen_ref)
     write(v) bb, Bx, gg, Gx, coszen_ref, swddir_ref, swdown_ref
     o_down = swdown; o_dir = swddir; o_dni = 0; o_dif = 0; o_gsw = 0
     call interp_sw_radiation(1,n,1,1,1,n,1,1, coszen_ref,cl2,swddir_ref,bb,Bx,swdown_ref,gg,Gx,albedo, &
                              o_down,o_dir,o_dni,o_dif,o_gsw)
     write(v) o_down, o_dir, o_dni, o_dif, o_gsw
  end do
  close(u); close(v)
end program ref_harness_swinterp
