#!/bin/bash
# Profiles of one round, on the GPU box:  bash tools/profile_round.sh r02
#   1. rocprofv3 --kernel-trace --stats around the default bench.py run (the summary the bench line must agree with)
#   2. separate --pmc passes (no trace domains mixed in) over tools/pmc_workload.py: busy figures of every tile kernel
#   3. FETCH_SIZE / WRITE_SIZE passes (separate) over tools/pmc_workload.py: HBM-side traffic of every tile kernel -> traffic.json
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tag=${1:-r05}; out=gpurun_out/prof_$tag; mkdir -p $out
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $out/stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/bench.json 2> $out/bench.err || echo "stats pass failed"
python3 tools/rocpd_summary.py $out/stats > $out/${tag}_bench_kernel_stats.txt 2>&1
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $set -d $out/busy/pass$i -- python3 tools/pmc_workload.py > $out/busy_pass$i.log 2>&1 || echo "busy pass $i failed"
done
python3 tools/pmc_busy.py $out/busy $out/pmc_busy.json > $out/${tag}_pmc_busy.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c -d $out/traffic/$c -- python3 tools/pmc_workload.py > $out/traffic_$c.log 2>&1 || echo "traffic pass $c failed"
done
python3 tools/pmc_traffic.py $out/traffic "profiles/${tag}_pmc_traffic.txt (rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes over tools/pmc_workload.py; tools/pmc_traffic.py)" $out/traffic.json > $out/${tag}_pmc_traffic.txt 2>&1
rm -rf $out/stats $out/busy $out/traffic       # the databases are large; the text summaries are what is kept
ls -la $out
