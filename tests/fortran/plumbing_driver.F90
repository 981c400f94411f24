! Plumbing driver for module_radiation_driver_hip: one radiation step (parameter update), one interpolation and one
! bucket accumulation on n x 1 points, called exactly as radiation_driver calls the reference routines.
PROGRAM plumbing_driver
   USE module_radiation_driver_hip
   IMPLICIT NONE
   INTEGER :: n
   REAL, ALLOCATABLE, DIMENSION(:,:) :: coszen, coszen_loc, swddir, swdown, cl2, albedo
   REAL, ALLOCATABLE, DIMENSION(:,:) :: swddir_ref, bb, Bx, swdown_ref, gg, Gx, coszen_ref
   REAL, ALLOCATABLE, DIMENSION(:,:) :: o_down, o_dir, o_dni, o_dif, o_gsw
   REAL, ALLOCATABLE, DIMENSION(:,:,:) :: flux, ac
   CHARACTER(len=256) :: fin, fout
   CALL get_command_argument(1, fin)
   CALL get_command_argument(2, fout)
   OPEN(11, file=TRIM(fin), access='stream', form='unformatted', status='old')
   READ(11) n
   ALLOCATE(coszen(n,1), coszen_loc(n,1), swddir(n,1), swdown(n,1), cl2(n,1), albedo(n,1))
   ALLOCATE(swddir_ref(n,1), bb(n,1), Bx(n,1), swdown_ref(n,1), gg(n,1), Gx(n,1), coszen_ref(n,1))
   ALLOCATE(o_down(n,1), o_dir(n,1), o_dni(n,1), o_dif(n,1), o_gsw(n,1), flux(n,1,2), ac(n,1,2))
   READ(11) coszen, coszen_loc, swddir, swdown, cl2, albedo
   READ(11) swddir_ref, bb, Bx, swdown_ref, gg, Gx, coszen_ref
   READ(11) flux, ac
   CLOSE(11)
   CALL update_swinterp_parameters(1,n,1,1,1,n,1,1, coszen,coszen_loc,swddir,swdown, &
                                   swddir_ref,bb,Bx,swdown_ref,gg,Gx,coszen_ref)
   o_down = swdown; o_dir = swddir; o_dni = -1.; o_dif = -1.; o_gsw = -1.
   CALL interp_sw_radiation(1,n,1,1,1,n,1,1, coszen_ref,cl2,swddir_ref,bb,Bx,swdown_ref,gg,Gx,albedo, &
                            o_down,o_dir,o_dni,o_dif,o_gsw)
   CALL accumulate_rad_fluxes(1,n,1,1,1,n,1,1, 18.0, 2, flux, ac)
   OPEN(12, file=TRIM(fout), access='stream', form='unformatted', status='replace')
   WRITE(12) swddir_ref, bb, Bx, swdown_ref, gg, Gx, coszen_ref
   WRITE(12) o_down, o_dir, o_dni, o_dif, o_gsw
   WRITE(12) ac
   CLOSE(12)
END PROGRAM plumbing_driver
