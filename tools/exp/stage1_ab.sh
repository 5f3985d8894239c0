cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
cp phyloligo_amd/libphyloligo_amd.so /tmp/orig.so
for v in 1_0 1_1 8_0 orig; do
  if [ $v = orig ]; then cp /tmp/orig.so phyloligo_amd/libphyloligo_amd.so; else cp tools/exp/libcnt_$v.so phyloligo_amd/libphyloligo_amd.so; fi
  for cfg in 1111_both ragged; do
    rm -rf gpurun_out/s1prof/x
    timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/s1prof/x -- python3 tools/stage1_time.py $cfg > /dev/null 2>&1
    echo "== passes_lean=$v $cfg: $(python3 tools/rocpd_summary.py gpurun_out/s1prof/x | grep -E 'count_kernel')"
  done
done
rm -rf gpurun_out/s1prof/x
