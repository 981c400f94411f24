! oracle/ref_harness_lw.F90 -- TEST INFRASTRUCTURE.
!
! Host program for the reference's own longwave column solver (see
! ref_harness_sw.F90 for the build rules: object code comes straight from
! /root/reference, this file contains no reference code).  Fills the rrlw_kgNN
! tables from a RRTMG_LW_DATA-layout file (READ field order of lw_kgb01..16,
! LW:13021-14387), calls rrtmg_lw_ini and rrtmg_lw (clean + aerosol calls).
!
! usage: ref_lw <tables> <columns.bin> <out.bin> [clouds]
!
! With the fourth argument "clouds" every column record is followed by the in-cloud state RRTMG_LWRAD hands to the
! McICA generator (cldfrac, clwp, ciwp, cswp, rel, rei, res per layer and inflglw/iceflglw/liqflglw); the harness
! then calls the reference's mcica_subcol_lw (icld = 2, permuteseed = 150, irng = 0 as at LW:12553-12563) before
! rrtmg_lw, and also dumps the sub-column cloud mask.

subroutine wrf_error_fatal(msg)
  character(len=*), intent(in) :: msg
  write(0,*) 'wrf_error_fatal: ', trim(msg)
  stop 2
end subroutine wrf_error_fatal

program ref_lw
  use parkind, only : rb => kind_rb, im => kind_im
  use parrrtm, only : nbndlw, ngptlw
  use rrtmg_lw_init, only : rrtmg_lw_ini
  use rrtmg_lw_rad, only : rrtmg_lw
  use mcica_subcol_gen_lw, only : mcica_subcol_lw
  implicit none
  character(len=512) :: ftab, fcol, fout, fmode
  logical :: clouds
  integer(kind=im) :: irng, permuteseed, iflags(3)
  real(kind=rb), allocatable :: cldfrac(:,:), clwp(:,:), ciwp(:,:), cswp(:,:), rel(:,:), rei(:,:), res(:,:)
  real(kind=rb), allocatable :: taucld(:,:,:)
  integer :: ncol, nlay, iclean, nrep, ic, irep
  integer(kind=im) :: icld, inflglw, iceflglw, liqflglw
  real(kind=rb) :: cpdair
  real(kind=rb), allocatable :: play(:,:), plev(:,:), tlay(:,:), tlev(:,:), tsfc(:), emis(:,:)
  real(kind=rb), allocatable :: h2ovmr(:,:), o3vmr(:,:), co2vmr(:,:), ch4vmr(:,:), n2ovmr(:,:), o2vmr(:,:)
  real(kind=rb), allocatable :: cfc11vmr(:,:), cfc12vmr(:,:), cfc22vmr(:,:), ccl4vmr(:,:)
  real(kind=rb), allocatable :: cldfmcl(:,:,:), taucmcl(:,:,:), ciwpmcl(:,:,:), clwpmcl(:,:,:), cswpmcl(:,:,:)
  real(kind=rb), allocatable :: reicmcl(:,:), relqmcl(:,:), resnmcl(:,:), tauaer(:,:,:)
  real(kind=rb), allocatable :: uflx(:,:), dflx(:,:), hr(:,:), uflxc(:,:), dflxc(:,:), hrc(:,:)
  real(kind=rb), allocatable :: uflxcln(:,:), dflxcln(:,:), buf(:)
  integer :: t0, t1, trate
  real(kind=8) :: secs

  if (command_argument_count() < 3) stop 'usage: ref_lw tables columns out'
  call get_command_argument(1, ftab)
  call get_command_argument(2, fcol)
  call get_command_argument(3, fout)
  clouds = .false.
  if (command_argument_count() >= 4) then
     call get_command_argument(4, fmode)
     clouds = trim(fmode) == 'clouds'
  end if
  call read_tables(trim(ftab))
  cpdair = 7.0_rb * 287.0_rb / 2.0_rb
  call rrtmg_lw_ini(cpdair)

  open(21, file=trim(fcol), access='stream', form='unformatted', status='old')
  read(21) ncol, nlay, iclean, nrep
  open(22, file=trim(fout), access='stream', form='unformatted', status='replace')
  allocate(play(1,nlay), plev(1,nlay+1), tlay(1,nlay), tlev(1,nlay+1), tsfc(1), emis(1,nbndlw))
  allocate(h2ovmr(1,nlay), o3vmr(1,nlay), co2vmr(1,nlay), ch4vmr(1,nlay), n2ovmr(1,nlay), o2vmr(1,nlay))
  allocate(cfc11vmr(1,nlay), cfc12vmr(1,nlay), cfc22vmr(1,nlay), ccl4vmr(1,nlay))
  allocate(cldfmcl(ngptlw,1,nlay), taucmcl(ngptlw,1,nlay), ciwpmcl(ngptlw,1,nlay), clwpmcl(ngptlw,1,nlay))
  allocate(cswpmcl(ngptlw,1,nlay), reicmcl(1,nlay), relqmcl(1,nlay), resnmcl(1,nlay), tauaer(1,nlay,nbndlw))
  allocate(uflx(1,nlay+1), dflx(1,nlay+1), hr(1,nlay), uflxc(1,nlay+1), dflxc(1,nlay+1), hrc(1,nlay))
  allocate(uflxcln(1,nlay+1), dflxcln(1,nlay+1), buf(nlay*nbndlw))
  allocate(cldfrac(1,nlay), clwp(1,nlay), ciwp(1,nlay), cswp(1,nlay), rel(1,nlay), rei(1,nlay), res(1,nlay))
  allocate(taucld(nbndlw,1,nlay))
  cldfmcl = 0._rb; taucmcl = 0._rb; ciwpmcl = 0._rb; clwpmcl = 0._rb; cswpmcl = 0._rb
  reicmcl = 10._rb; relqmcl = 10._rb; resnmcl = 10._rb
  inflglw = 2; iceflglw = 3; liqflglw = 1

  call system_clock(t0, trate)
  do ic = 1, ncol
     read(21) play(1,:), plev(1,:), tlay(1,:), tlev(1,:), tsfc(1)
     read(21) h2ovmr(1,:), o3vmr(1,:), co2vmr(1,:), ch4vmr(1,:), n2ovmr(1,:), o2vmr(1,:)
     read(21) cfc11vmr(1,:), cfc12vmr(1,:), cfc22vmr(1,:), ccl4vmr(1,:)
     read(21) emis(1,:)
     read(21) buf; tauaer(1,:,:) = reshape(buf, (/nlay, nbndlw/))
     if (clouds) then
        read(21) cldfrac(1,:), clwp(1,:), ciwp(1,:), cswp(1,:), rel(1,:), rei(1,:), res(1,:)
        read(21) iflags
        inflglw = iflags(1); iceflglw = iflags(2); liqflglw = iflags(3)
        taucld = 0._rb
        icld = 2; irng = 0; permuteseed = 150
        call mcica_subcol_lw(1, 1, nlay, icld, permuteseed, irng, play, &
             cldfrac, ciwp, clwp, cswp, rei, rel, res, taucld, cldfmcl, &
             ciwpmcl, clwpmcl, cswpmcl, reicmcl, relqmcl, resnmcl, taucmcl)
     end if
     do irep = 1, max(1, nrep)
        icld = 2
        call rrtmg_lw(1, nlay, icld, play, plev, tlay, tlev, tsfc, &
             h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr, cfc11vmr, cfc12vmr, cfc22vmr, ccl4vmr, emis, &
             inflglw, iceflglw, liqflglw, cldfmcl, taucmcl, ciwpmcl, clwpmcl, cswpmcl, reicmcl, relqmcl, resnmcl, &
             tauaer, uflx, dflx, hr, uflxc, dflxc, hrc, uflxcln, dflxcln, iclean)
     end do
     write(22) uflx(1,:), dflx(1,:), hr(1,:), uflxc(1,:), dflxc(1,:), hrc(1,:), uflxcln(1,:), dflxcln(1,:)
     if (clouds) write(22) cldfmcl(:,1,:)
  end do
  call system_clock(t1)
  secs = real(t1 - t0, 8) / real(trate, 8)
  write(*,'(a,i8,a,i4,a,f12.6)') 'ref_lw columns=', ncol, ' reps=', max(1,nrep), ' seconds=', secs
  close(21); close(22)

contains

  subroutine read_tables(fn)
    use rrlw_kg01, only : fa01=>fracrefao, fb01=>fracrefbo, ka01=>kao, kb01=>kbo, ma01=>kao_mn2, mb01=>kbo_mn2, &
                          se01=>selfrefo, fo01=>forrefo
    use rrlw_kg02, only : fa02=>fracrefao, fb02=>fracrefbo, ka02=>kao, kb02=>kbo, se02=>selfrefo, fo02=>forrefo
    use rrlw_kg03, only : fa03=>fracrefao, fb03=>fracrefbo, ka03=>kao, kb03=>kbo, ma03=>kao_mn2o, mb03=>kbo_mn2o, &
                          se03=>selfrefo, fo03=>forrefo
    use rrlw_kg04, only : fa04=>fracrefao, fb04=>fracrefbo, ka04=>kao, kb04=>kbo, se04=>selfrefo, fo04=>forrefo
    use rrlw_kg05, only : fa05=>fracrefao, fb05=>fracrefbo, ka05=>kao, kb05=>kbo, ma05=>kao_mo3, cc05=>ccl4o, &
                          se05=>selfrefo, fo05=>forrefo
    use rrlw_kg06, only : fa06=>fracrefao, ka06=>kao, ma06=>kao_mco2, c106=>cfc11adjo, c206=>cfc12o, &
                          se06=>selfrefo, fo06=>forrefo
    use rrlw_kg07, only : fa07=>fracrefao, fb07=>fracrefbo, ka07=>kao, kb07=>kbo, ma07=>kao_mco2, mb07=>kbo_mco2, &
                          se07=>selfrefo, fo07=>forrefo
    use rrlw_kg08, only : fa08=>fracrefao, fb08=>fracrefbo, ka08=>kao, kb08=>kbo, ma08=>kao_mco2, mb08=>kbo_mco2, &
                          na08=>kao_mn2o, nb08=>kbo_mn2o, oa08=>kao_mo3, c208=>cfc12o, c308=>cfc22adjo, &
                          se08=>selfrefo, fo08=>forrefo
    use rrlw_kg09, only : fa09=>fracrefao, fb09=>fracrefbo, ka09=>kao, kb09=>kbo, ma09=>kao_mn2o, mb09=>kbo_mn2o, &
                          se09=>selfrefo, fo09=>forrefo
    use rrlw_kg10, only : fa10=>fracrefao, fb10=>fracrefbo, ka10=>kao, kb10=>kbo, se10=>selfrefo, fo10=>forrefo
    use rrlw_kg11, only : fa11=>fracrefao, fb11=>fracrefbo, ka11=>kao, kb11=>kbo, ma11=>kao_mo2, mb11=>kbo_mo2, &
                          se11=>selfrefo, fo11=>forrefo
    use rrlw_kg12, only : fa12=>fracrefao, ka12=>kao, se12=>selfrefo, fo12=>forrefo
    use rrlw_kg13, only : fa13=>fracrefao, fb13=>fracrefbo, ka13=>kao, ma13=>kao_mco2, na13=>kao_mco, mb13=>kbo_mo3, &
                          se13=>selfrefo, fo13=>forrefo
    use rrlw_kg14, only : fa14=>fracrefao, fb14=>fracrefbo, ka14=>kao, kb14=>kbo, se14=>selfrefo, fo14=>forrefo
    use rrlw_kg15, only : fa15=>fracrefao, ka15=>kao, ma15=>kao_mn2, se15=>selfrefo, fo15=>forrefo
    use rrlw_kg16, only : fa16=>fracrefao, fb16=>fracrefbo, ka16=>kao, kb16=>kbo, se16=>selfrefo, fo16=>forrefo
    character(len=*), intent(in) :: fn
    open(11, file=fn, form='unformatted', status='old')
    read(11) fa01, fb01, ka01, kb01, ma01, mb01, se01, fo01
    read(11) fa02, fb02, ka02, kb02, se02, fo02
    read(11) fa03, fb03, ka03, kb03, ma03, mb03, se03, fo03
    read(11) fa04, fb04, ka04, kb04, se04, fo04
    read(11) fa05, fb05, ka05, kb05, ma05, cc05, se05, fo05
    read(11) fa06, ka06, ma06, c106, c206, se06, fo06
    read(11) fa07, fb07, ka07, kb07, ma07, mb07, se07, fo07
    read(11) fa08, fb08, ka08, kb08, ma08, mb08, na08, nb08, oa08, c208, c308, se08, fo08
    read(11) fa09, fb09, ka09, kb09, ma09, mb09, se09, fo09
    read(11) fa10, fb10, ka10, kb10, se10, fo10
    read(11) fa11, fb11, ka11, kb11, ma11, mb11, se11, fo11
    read(11) fa12, ka12, se12, fo12
    read(11) fa13, fb13, ka13, ma13, na13, mb13, se13, fo13
    read(11) fa14, fb14, ka14, kb14, se14, fo14
    read(11) fa15, ka15, ma15, se15, fo15
    read(11) fa16, fb16, ka16, kb16, se16, fo16
    close(11)
  end subroutine read_tables

end program ref_lw
