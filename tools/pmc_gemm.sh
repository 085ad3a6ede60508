#!/bin/bash
# PMC passes over the M = 256 projection path of one layer (tools/pmc_gemm.py): one rocprofv3 run per counter set.
TAG=${1:-r03}
R=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum"; do
  NAME=$(echo $C | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $C -d $R/gpurun_out/${TAG}_pmcg_$NAME -- python $R/tools/pmc_gemm.py --manifest $R/gpurun_out/${TAG}_pmc_gemm_manifest.json > $R/gpurun_out/${TAG}_pmcg_$NAME.log 2>&1
  DB=$(find $R/gpurun_out/${TAG}_pmcg_$NAME -name "*results.db" | head -1)
  if [ -n "$DB" ]; then
    timeout 120 python $R/tools/pmc_gemm_summary.py $DB $R/gpurun_out/${TAG}_pmc_gemm_manifest.json --json $R/gpurun_out/${TAG}_pmc_gemm_$NAME.json > $R/gpurun_out/${TAG}_pmc_gemm_$NAME.txt 2>&1
    cat $R/gpurun_out/${TAG}_pmc_gemm_$NAME.txt | cut -c1-200
  else
    echo "no db for $C"; tail -5 $R/gpurun_out/${TAG}_pmcg_$NAME.log
  fi
  find $R/gpurun_out/${TAG}_pmcg_$NAME -name "*.db" -delete
done
