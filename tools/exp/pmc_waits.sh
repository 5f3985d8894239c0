#!/bin/bash
# wave-state breakdown of one tile kernel:  pmc_waits.sh <metric> [pattern]   (separate rocprofv3 --pmc passes, no trace domains)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
m=${1:-JSD}; pat=${2:-1111}
out=gpurun_out/pmc_waits_$m; rm -rf $out; mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set -d $out/pass$i -- python3 tools/one_launch.py 50000 $m 2 $pat > $out/pass$i.log 2>&1 || echo "pass $i failed"
done
python3 tools/rocpd_summary.py $out | grep -E "rows_kernel|pairdot_tile|valu_tile|gram_" | grep -v "top_kernels"
rm -rf $out/pass*/
