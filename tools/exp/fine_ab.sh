cd $GRAFT_REPO_ROOT
cp phyloligo_amd/libphyloligo_amd.so /tmp/orig.so
for v in fine orig; do
  if [ $v = orig ]; then cp /tmp/orig.so phyloligo_amd/libphyloligo_amd.so; else cp tools/exp/libfine.so phyloligo_amd/libphyloligo_amd.so; fi
  echo "== $v"; timeout -k 10 200 python tools/one_launch.py 50000 JSD 3 1111 equal notable 2>&1 | grep JSD
  timeout -k 10 200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz_pairwise.py -m gpu -q -k "JSD or jsd or near or dupl or fuzz" 2>&1 | tail -2
done
cp /tmp/orig.so phyloligo_amd/libphyloligo_amd.so
