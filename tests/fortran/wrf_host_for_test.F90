! Test-only host pieces for the Fortran shim (NOT used to build any reference
! code): the one constant and the error hook the shim takes from WRF.
MODULE module_model_constants
   REAL, PARAMETER :: r_d = 287.
   REAL, PARAMETER :: cp = 7.*r_d/2.
END MODULE module_model_constants

SUBROUTINE wrf_error_fatal(msg)
   CHARACTER(len=*), INTENT(IN) :: msg
   WRITE(0,'(a)') 'FATAL CALLED FROM RRTMG SHIM: '//TRIM(msg)
   STOP 3
END SUBROUTINE wrf_error_fatal
