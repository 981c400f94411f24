#!/bin/bash
set -o pipefail
out=gpurun_out/pmcx
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 -L > $out/list.txt 2>&1 || true
i=0
for set in "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT" \
           "SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_IFETCH SQ_IFETCH_LEVEL" \
           "SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_INSTS_SENDMSG" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TA_TCP_STATE_READ_sum" \
           "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $out/p$i -o pmc -- python3 bench.py --mode sw --steps 3 --warmup 1 --no-cpu > $out/p$i.log 2>&1 || echo "pass $i failed: $set"
  echo "pass $i done"
done
