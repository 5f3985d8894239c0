cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/s1prof; mkdir -p gpurun_out/s1prof
for cfg in ${CFGS:-1111_both 1111_plus 11011011_both 111111_both 1101_both 111111_minus 1111_minus 1101_minus 110100111_both ragged ragged_dirty}; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/s1prof/$cfg -- python3 tools/stage1_time.py $cfg > gpurun_out/s1prof/$cfg.log 2>&1
  echo "== $cfg: $(python3 tools/rocpd_summary.py gpurun_out/s1prof/$cfg | grep -E 'count_kernel')"
  rm -rf gpurun_out/s1prof/$cfg
done
