#!/bin/bash
# hardware counters of the attention kernels on one layer shape: tools/pmc_attn.sh enc|dec|cross
kind=${1:-enc}
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_attn_$kind
rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_INSTS_BRANCH"; do
  i=$((i+1))
  rocprofv3 --output-format csv --pmc $set -d $out/p$i -o p$i -- python $GRAFT_REPO_ROOT/tools/attn_bench.py $kind > $out/log$i.txt 2>&1
done
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $out attn_ > $GRAFT_REPO_ROOT/gpurun_out/pmc_attn_$kind.txt
find $out -name "*.db" -delete; find $out -name "*.csv" -size +2M -delete
