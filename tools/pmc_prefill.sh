#!/bin/bash
# PMC passes over the prefill attention kernels (tools/pmc_prefill.py): one rocprofv3 run per counter set.
TAG=${1:-r03}
shift
IMPLS="$@"
R=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/${TAG}_pmc_prefill.txt
: > $OUT
SETS=${PMC_SETS:-all}
for C in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA" \
         "SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_BUSY_CU_CYCLES SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA" \
         "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
  NAME=$(echo $C | cut -d' ' -f1)
  if [ "$SETS" = "mem" ] && [[ "$NAME" == SQ_* ]]; then continue; fi
  timeout 200 rocprofv3 --pmc $C -d $R/gpurun_out/${TAG}_pmcp_$NAME -- python $R/tools/pmc_prefill.py $IMPLS > $R/gpurun_out/${TAG}_pmcp_$NAME.log 2>&1
  DB=$(find $R/gpurun_out/${TAG}_pmcp_$NAME -name "*results.db" | head -1)
  if [ -n "$DB" ]; then
    timeout 120 python $R/tools/pmc_prefill_summary.py $DB >> $OUT 2>&1
  else
    echo "no db for $C" >> $OUT; tail -5 $R/gpurun_out/${TAG}_pmcp_$NAME.log >> $OUT
  fi
  rm -rf $R/gpurun_out/${TAG}_pmcp_$NAME
done
cat $OUT
