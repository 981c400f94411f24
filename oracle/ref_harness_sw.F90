! oracle/ref_harness_sw.F90 -- TEST INFRASTRUCTURE.
!
! Host program for the reference's own shortwave column solver.  It is linked
! against object code compiled by oracle/build_ref.sh directly from
! /root/reference/WRF-Chem_code/v3.9.1/phys/module_ra_rrtmg_{lw,sw}.F (the
! self-contained inner modules only; no stand-in modules).  This file contains
! no reference code: it fills the reference's module tables from a
! RRTMG_SW_DATA-layout file (same READ field order as sw_kgbNN, SW:11379-12366),
! calls rrtmg_sw_ini and rrtmg_sw, and dumps the results.
!
! usage: ref_sw <tables> <columns.bin> <out.bin> [dump_tables.bin|-] [clouds]
!
! With the fifth argument "clouds" every column record is followed by the in-cloud state the WRF wrapper hands to
! the McICA generator (cldfrac, clwp, ciwp, cswp, rel, rei, res per layer and inflgsw/iceflgsw/liqflgsw); the
! harness then calls the reference's mcica_subcol_sw (icld = 2, permuteseed = 1, irng = 0 as at SW:10950-10957)
! before rrtmg_sw, and also dumps the sub-column cloud mask.
!
! The only external the reference objects need from their host is the error
! hook wrf_error_fatal (cloud-optics range checks, SW:2168-2262); it is
! provided below as an abort and is never reached on cloud-free input.

subroutine wrf_error_fatal(msg)
  character(len=*), intent(in) :: msg
  write(0,*) 'wrf_error_fatal: ', trim(msg)
  stop 2
end subroutine wrf_error_fatal

program ref_sw
  use parkind, only : rb => kind_rb, im => kind_im
  use parrrsw, only : nbndsw, ngptsw, naerec
  use rrtmg_sw_init, only : rrtmg_sw_ini
  use rrtmg_sw_rad, only : rrtmg_sw
  use mcica_subcol_gen_sw, only : mcica_subcol_sw
  implicit none
  character(len=512) :: ftab, fcol, fout, fdump, fmode
  logical :: clouds, ecmwf
  integer(kind=im) :: irng, permuteseed, iflags(3)
  real(kind=rb), allocatable :: cldfrac(:,:), clwp(:,:), ciwp(:,:), cswp(:,:), rel(:,:), rei(:,:), res(:,:)
  real(kind=rb), allocatable :: taucld(:,:,:), ssacld(:,:,:), asmcld(:,:,:), fsfcld(:,:,:)
  integer :: ncol, nlay, iclean, nrep, ic, irep, nargs
  integer(kind=im) :: icld, inflgsw, iceflgsw, liqflgsw, dyofyr
  real(kind=rb) :: cpdair, adjes, scon
  real(kind=rb), allocatable :: play(:,:), plev(:,:), tlay(:,:), tlev(:,:), tsfc(:)
  real(kind=rb), allocatable :: h2ovmr(:,:), o3vmr(:,:), co2vmr(:,:), ch4vmr(:,:), n2ovmr(:,:), o2vmr(:,:)
  real(kind=rb), allocatable :: asdir(:), asdif(:), aldir(:), aldif(:), coszen(:)
  real(kind=rb), allocatable :: cldfmcl(:,:,:), taucmcl(:,:,:), ssacmcl(:,:,:), asmcmcl(:,:,:), fsfcmcl(:,:,:)
  real(kind=rb), allocatable :: ciwpmcl(:,:,:), clwpmcl(:,:,:), cswpmcl(:,:,:)
  real(kind=rb), allocatable :: reicmcl(:,:), relqmcl(:,:), resnmcl(:,:)
  real(kind=rb), allocatable :: tauaer(:,:,:), ssaaer(:,:,:), asmaer(:,:,:), ecaer(:,:,:)
  real(kind=rb), allocatable :: swuflx(:,:), swdflx(:,:), swhr(:,:), swuflxc(:,:), swdflxc(:,:), swhrc(:,:)
  real(kind=rb), allocatable :: swuflxcln(:,:), swdflxcln(:,:)
  real(kind=rb), allocatable :: sibvisdir(:,:), sibvisdif(:,:), sibnirdir(:,:), sibnirdif(:,:)
  real(kind=rb), allocatable :: swdkdir(:,:), swdkdif(:,:)
  real(kind=rb), allocatable :: buf(:)
  real(kind=rb) :: scal(7)
  integer :: t0, t1, trate
  real(kind=8) :: secs

  nargs = command_argument_count()
  if (nargs < 3) stop 'usage: ref_sw tables columns out [dump]'
  call get_command_argument(1, ftab)
  call get_command_argument(2, fcol)
  call get_command_argument(3, fout)
  fdump = ''
  if (nargs >= 4) call get_command_argument(4, fdump)
  if (trim(fdump) == '-') fdump = ''
  clouds = .false.; ecmwf = .false.
  if (nargs >= 5) then
     call get_command_argument(5, fmode)
     clouds = trim(fmode) == 'clouds'
     ecmwf = trim(fmode) == 'ecaer'      ! aer_opt = 1: six ECMWF aerosol types, iaer = 6
  end if

  call read_tables(trim(ftab))
  cpdair = 7.0_rb * 287.0_rb / 2.0_rb       ! WRF module_model_constants: cp = 7.*r_d/2.
  call rrtmg_sw_ini(cpdair)
  if (len_trim(fdump) > 0) call dump_tables(trim(fdump))

  open(21, file=trim(fcol), access='stream', form='unformatted', status='old')
  read(21) ncol, nlay, iclean, nrep
  open(22, file=trim(fout), access='stream', form='unformatted', status='replace')

  allocate(play(1,nlay), plev(1,nlay+1), tlay(1,nlay), tlev(1,nlay+1), tsfc(1))
  allocate(h2ovmr(1,nlay), o3vmr(1,nlay), co2vmr(1,nlay), ch4vmr(1,nlay), n2ovmr(1,nlay), o2vmr(1,nlay))
  allocate(asdir(1), asdif(1), aldir(1), aldif(1), coszen(1))
  allocate(cldfmcl(ngptsw,1,nlay), taucmcl(ngptsw,1,nlay), ssacmcl(ngptsw,1,nlay), asmcmcl(ngptsw,1,nlay))
  allocate(fsfcmcl(ngptsw,1,nlay), ciwpmcl(ngptsw,1,nlay), clwpmcl(ngptsw,1,nlay), cswpmcl(ngptsw,1,nlay))
  allocate(reicmcl(1,nlay), relqmcl(1,nlay), resnmcl(1,nlay))
  allocate(tauaer(1,nlay,nbndsw), ssaaer(1,nlay,nbndsw), asmaer(1,nlay,nbndsw), ecaer(1,nlay,naerec))
  allocate(swuflx(1,nlay+1), swdflx(1,nlay+1), swhr(1,nlay), swuflxc(1,nlay+1), swdflxc(1,nlay+1), swhrc(1,nlay))
  allocate(swuflxcln(1,nlay+1), swdflxcln(1,nlay+1))
  allocate(sibvisdir(1,nlay+1), sibvisdif(1,nlay+1), sibnirdir(1,nlay+1), sibnirdif(1,nlay+1))
  allocate(swdkdir(1,nlay+1), swdkdif(1,nlay+1))
  allocate(buf(nlay*nbndsw))
  allocate(cldfrac(1,nlay), clwp(1,nlay), ciwp(1,nlay), cswp(1,nlay), rel(1,nlay), rei(1,nlay), res(1,nlay))
  allocate(taucld(nbndsw,1,nlay), ssacld(nbndsw,1,nlay), asmcld(nbndsw,1,nlay), fsfcld(nbndsw,1,nlay))

  ! cloud-free McICA inputs (what mcica_subcol_sw yields for cldfrac = 0)
  cldfmcl = 0._rb; taucmcl = 0._rb; ssacmcl = 1._rb; asmcmcl = 0._rb; fsfcmcl = 0._rb
  ciwpmcl = 0._rb; clwpmcl = 0._rb; cswpmcl = 0._rb
  reicmcl = 10._rb; relqmcl = 10._rb; resnmcl = 10._rb
  ecaer = 0._rb
  inflgsw = 2; iceflgsw = 3; liqflgsw = 1; dyofyr = 0

  call system_clock(t0, trate)
  do ic = 1, ncol
     read(21) play(1,:), plev(1,:), tlay(1,:), tlev(1,:), tsfc(1)
     read(21) h2ovmr(1,:), o3vmr(1,:), co2vmr(1,:), ch4vmr(1,:), n2ovmr(1,:), o2vmr(1,:)
     read(21) scal
     asdir(1) = scal(1); asdif(1) = scal(2); aldir(1) = scal(3); aldif(1) = scal(4)
     coszen(1) = scal(5); adjes = scal(6); scon = scal(7)
     read(21) buf; tauaer(1,:,:) = reshape(buf, (/nlay, nbndsw/))
     read(21) buf; ssaaer(1,:,:) = reshape(buf, (/nlay, nbndsw/))
     read(21) buf; asmaer(1,:,:) = reshape(buf, (/nlay, nbndsw/))
     if (ecmwf) then
        deallocate(buf); allocate(buf(nlay*naerec))
        read(21) buf; ecaer(1,:,:) = reshape(buf, (/nlay, naerec/))
        deallocate(buf); allocate(buf(nlay*nbndsw))
     end if
     if (clouds) then
        read(21) cldfrac(1,:), clwp(1,:), ciwp(1,:), cswp(1,:), rel(1,:), rei(1,:), res(1,:)
        read(21) iflags
        inflgsw = iflags(1); iceflgsw = iflags(2); liqflgsw = iflags(3)
        taucld = 0._rb; ssacld = 1._rb; asmcld = 0._rb; fsfcld = 0._rb      ! SW:10925-10932
        icld = 2; irng = 0; permuteseed = 1                                  ! SW:10950-10952
        call mcica_subcol_sw(1, 1, nlay, icld, permuteseed, irng, play, &
             cldfrac, ciwp, clwp, cswp, rei, rel, res, taucld, ssacld, asmcld, fsfcld, &
             cldfmcl, ciwpmcl, clwpmcl, cswpmcl, reicmcl, relqmcl, resnmcl, &
             taucmcl, ssacmcl, asmcmcl, fsfcmcl)
     end if
     do irep = 1, max(1, nrep)
        icld = 2
        call rrtmg_sw(1, nlay, icld, play, plev, tlay, tlev, tsfc, &
             h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr, &
             asdir, asdif, aldir, aldif, coszen, adjes, dyofyr, scon, &
             inflgsw, iceflgsw, liqflgsw, cldfmcl, taucmcl, ssacmcl, asmcmcl, fsfcmcl, &
             ciwpmcl, clwpmcl, cswpmcl, reicmcl, relqmcl, resnmcl, &
             tauaer, ssaaer, asmaer, ecaer, &
             swuflx, swdflx, swhr, swuflxc, swdflxc, swhrc, swuflxcln, swdflxcln, merge(1, 0, ecmwf), &
             sibvisdir, sibvisdif, sibnirdir, sibnirdif, swdkdir, swdkdif, iclean)
     end do
     write(22) swuflx(1,:), swdflx(1,:), swhr(1,:), swuflxc(1,:), swdflxc(1,:), swhrc(1,:)
     write(22) swuflxcln(1,:), swdflxcln(1,:)
     write(22) sibvisdir(1,:), sibvisdif(1,:), sibnirdir(1,:), sibnirdif(1,:), swdkdir(1,:), swdkdif(1,:)
     if (clouds) write(22) cldfmcl(:,1,:)
  end do
  call system_clock(t1)
  secs = real(t1 - t0, 8) / real(trate, 8)
  write(*,'(a,i8,a,i4,a,f12.6)') 'ref_sw columns=', ncol, ' reps=', max(1,nrep), ' seconds=', secs
  close(21); close(22)

contains

  subroutine read_tables(fn)
    use rrsw_kg16, only : r16=>rayl, s16=>strrat1, l16=>layreffr, ka16=>kao, kb16=>kbo, se16=>selfrefo, &
                          fo16=>forrefo, sf16=>sfluxrefo
    use rrsw_kg17, only : r17=>rayl, s17=>strrat, l17=>layreffr, ka17=>kao, kb17=>kbo, se17=>selfrefo, &
                          fo17=>forrefo, sf17=>sfluxrefo
    use rrsw_kg18, only : r18=>rayl, s18=>strrat, l18=>layreffr, ka18=>kao, kb18=>kbo, se18=>selfrefo, &
                          fo18=>forrefo, sf18=>sfluxrefo
    use rrsw_kg19, only : r19=>rayl, s19=>strrat, l19=>layreffr, ka19=>kao, kb19=>kbo, se19=>selfrefo, &
                          fo19=>forrefo, sf19=>sfluxrefo
    use rrsw_kg20, only : r20=>rayl, l20=>layreffr, c20=>absch4o, ka20=>kao, kb20=>kbo, se20=>selfrefo, &
                          fo20=>forrefo, sf20=>sfluxrefo
    use rrsw_kg21, only : r21=>rayl, s21=>strrat, l21=>layreffr, ka21=>kao, kb21=>kbo, se21=>selfrefo, &
                          fo21=>forrefo, sf21=>sfluxrefo
    use rrsw_kg22, only : r22=>rayl, s22=>strrat, l22=>layreffr, ka22=>kao, kb22=>kbo, se22=>selfrefo, &
                          fo22=>forrefo, sf22=>sfluxrefo
    use rrsw_kg23, only : r23=>raylo, g23=>givfac, l23=>layreffr, ka23=>kao, se23=>selfrefo, &
                          fo23=>forrefo, sf23=>sfluxrefo
    use rrsw_kg24, only : ra24=>raylao, rb24=>raylbo, s24=>strrat, l24=>layreffr, oa24=>abso3ao, ob24=>abso3bo, &
                          ka24=>kao, kb24=>kbo, se24=>selfrefo, fo24=>forrefo, sf24=>sfluxrefo
    use rrsw_kg25, only : r25=>raylo, l25=>layreffr, oa25=>abso3ao, ob25=>abso3bo, ka25=>kao, sf25=>sfluxrefo
    use rrsw_kg26, only : r26=>raylo, sf26=>sfluxrefo
    use rrsw_kg27, only : r27=>raylo, k27=>scalekur, l27=>layreffr, ka27=>kao, kb27=>kbo, sf27=>sfluxrefo
    use rrsw_kg28, only : r28=>rayl, s28=>strrat, l28=>layreffr, ka28=>kao, kb28=>kbo, sf28=>sfluxrefo
    use rrsw_kg29, only : r29=>rayl, l29=>layreffr, h29=>absh2oo, c29=>absco2o, ka29=>kao, kb29=>kbo, &
                          se29=>selfrefo, fo29=>forrefo, sf29=>sfluxrefo
    character(len=*), intent(in) :: fn
    open(11, file=fn, form='unformatted', status='old')
    read(11) r16, s16, l16, ka16, kb16, se16, fo16, sf16
    read(11) r17, s17, l17, ka17, kb17, se17, fo17, sf17
    read(11) r18, s18, l18, ka18, kb18, se18, fo18, sf18
    read(11) r19, s19, l19, ka19, kb19, se19, fo19, sf19
    read(11) r20, l20, c20, ka20, kb20, se20, fo20, sf20
    read(11) r21, s21, l21, ka21, kb21, se21, fo21, sf21
    read(11) r22, s22, l22, ka22, kb22, se22, fo22, sf22
    read(11) r23, g23, l23, ka23, se23, fo23, sf23
    read(11) ra24, rb24, s24, l24, oa24, ob24, ka24, kb24, se24, fo24, sf24
    read(11) r25, l25, oa25, ob25, ka25, sf25
    read(11) r26, sf26
    read(11) r27, k27, l27, ka27, kb27, sf27
    read(11) r28, s28, l28, ka28, kb28, sf28
    read(11) r29, l29, h29, c29, ka29, kb29, se29, fo29, sf29
    close(11)
  end subroutine read_tables

  ! reduced (112 g-point) tables, in the reference's memory order, for the
  ! parity test of the init path (cmbgbNN)
  subroutine dump_tables(fn)
    use rrsw_tbl, only : exp_tbl
    use rrsw_wvn, only : rwgt
    use rrsw_kg16, only : a16=>absa, b16=>absb, e16=>selfref, f16=>forref, x16=>sfluxref
    use rrsw_kg17, only : a17=>absa, b17=>absb, e17=>selfref, f17=>forref, x17=>sfluxref
    use rrsw_kg18, only : a18=>absa, b18=>absb, e18=>selfref, f18=>forref, x18=>sfluxref
    use rrsw_kg19, only : a19=>absa, b19=>absb, e19=>selfref, f19=>forref, x19=>sfluxref
    use rrsw_kg20, only : a20=>absa, b20=>absb, e20=>selfref, f20=>forref, x20=>sfluxref, m20=>absch4
    use rrsw_kg21, only : a21=>absa, b21=>absb, e21=>selfref, f21=>forref, x21=>sfluxref
    use rrsw_kg22, only : a22=>absa, b22=>absb, e22=>selfref, f22=>forref, x22=>sfluxref
    use rrsw_kg23, only : a23=>absa, e23=>selfref, f23=>forref, x23=>sfluxref, y23=>rayl
    use rrsw_kg24, only : a24=>absa, b24=>absb, e24=>selfref, f24=>forref, x24=>sfluxref, &
                          ya24=>rayla, yb24=>raylb, oa24=>abso3a, ob24=>abso3b
    use rrsw_kg25, only : a25=>absa, x25=>sfluxref, y25=>rayl, oa25=>abso3a, ob25=>abso3b
    use rrsw_kg26, only : x26=>sfluxref, y26=>rayl
    use rrsw_kg27, only : a27=>absa, b27=>absb, x27=>sfluxref, y27=>rayl
    use rrsw_kg28, only : a28=>absa, b28=>absb, x28=>sfluxref
    use rrsw_kg29, only : a29=>absa, b29=>absb, e29=>selfref, f29=>forref, x29=>sfluxref, h29=>absh2o, c29=>absco2
    character(len=*), intent(in) :: fn
    open(12, file=fn, access='stream', form='unformatted', status='replace')
    write(12) exp_tbl, rwgt
    write(12) a16, b16, e16, f16, x16
    write(12) a17, b17, e17, f17, x17
    write(12) a18, b18, e18, f18, x18
    write(12) a19, b19, e19, f19, x19
    write(12) a20, b20, e20, f20, x20, m20
    write(12) a21, b21, e21, f21, x21
    write(12) a22, b22, e22, f22, x22
    write(12) a23, e23, f23, x23, y23
    write(12) a24, b24, e24, f24, x24, ya24, yb24, oa24, ob24
    write(12) a25, x25, y25, oa25, ob25
    write(12) x26, y26
    write(12) a27, b27, x27, y27
    write(12) a28, b28, x28
    write(12) a29, b29, e29, f29, x29, h29, c29
    close(12)
  end subroutine dump_tables

end program ref_sw
