#!/bin/bash
# busy counters of the float32 Eucl kernels (stream and tile variant) at N = 50 000: bash tools/exp/pmc_eucl_f32.sh
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/pmc_eucl_f32; rm -rf $out; mkdir -p $out
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set -d $out/db/a$i -- python3 tools/exp/eucl_f32_time.py 50000 > $out/a$i.log 2>&1 || echo "pass a$i failed"
  [ -f tools/exp/variants/libB.so ] && PO_ALLOW_VARIANT=1 PO_LIB_PATH=tools/exp/variants/libB.so timeout -k 10 300 rocprofv3 --pmc $set -d $out/db/b$i -- python3 tools/exp/eucl_f32_time.py 50000 > $out/b$i.log 2>&1 || echo "pass b$i failed"
done
python3 tools/pmc_busy.py $out/db > $out/summary.txt 2>&1
rm -rf $out/db
cat $out/summary.txt
