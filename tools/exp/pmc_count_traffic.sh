cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/pmc_count_tr; rm -rf $out; mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c -d $out/$c -- python3 tools/stage1_time.py 1111_both > $out/$c.log 2>&1 || echo "pass $c failed"
done
python3 tools/rocpd_summary.py $out | grep -E "count_kernel"
rm -rf $out
