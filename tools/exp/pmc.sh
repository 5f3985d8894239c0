#!/bin/bash
# usage: pmc.sh <outdir> <script.py>   -- separate rocprofv3 --pmc passes (no trace domains mixed in)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=$1; shift
mkdir -p $out
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_WAVES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_VALU" \
           "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set -d $out/pass$i -- python3 "$@" > $out/pass$i.log 2>&1 || echo "pass $i failed"
done
python3 tools/rocpd_summary.py $out > $out/summary.txt 2>&1
grep -E "pairdot|kt_expand|bc_expand|jsd_lut|valu_tile|count_kernel" $out/summary.txt | head -80
