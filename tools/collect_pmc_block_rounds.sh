#!/bin/bash
# PMC counters of the Jacobi block-round kernel (compress! chi 256 -> 128, 24 sites): separate --pmc passes of four counters
# each, kernel-trace only, aggregated per dispatch by tools/_pmc_agg.py -> gpurun_out/r02_pmc_block_rounds.json
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
rm -rf $O/pmcb; mkdir -p $O/pmcb
i=0
for set in "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d $O/pmcb/p$i --output-format csv -- python3 $R/tools/_prof_compress.py 256 > $O/pmcb/p$i.log 2>&1
  f=$(find $O/pmcb/p$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 $R/tools/_pmc_agg.py "$f" "jacobi_block_round_nov<double, 8, 4, 64, false" > $O/pmcb/agg$i.txt
done
cat $O/pmcb/agg*.txt > $O/r02_pmc_block_rounds_raw.txt
cat $O/r02_pmc_block_rounds_raw.txt
rm -rf $O/pmcb
