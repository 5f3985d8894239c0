cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/pmc_count; rm -rf $out; mkdir -p $out
i=0
for set in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set -d $out/pass$i -- python3 tools/stage1_time.py 1111_both > $out/pass$i.log 2>&1 || echo "pass $i failed"
done
python3 tools/rocpd_summary.py $out | grep -E "count_kernel"
rm -rf $out/pass1 $out/pass2
