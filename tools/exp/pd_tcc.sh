#!/bin/bash
# VERDICT r03 item 6, counters first: L2 (TCC) hits / misses / fabric read requests of the pair-dot kernels, one --pmc pass
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/pd_tcc; rm -rf $out; mkdir -p $out
timeout -k 10 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCC_REQ_sum -d $out/p1 -- python3 tools/pmc_workload.py > $out/p1.log 2>&1 || echo "pass failed"
python3 tools/rocpd_summary.py $out/p1 2>&1 | grep -E "pairdot_tile_kernel|jsd_lut_rows|gram_i8_tile_kernel<1" 
rm -rf $out/p1
