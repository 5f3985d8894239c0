cd $GRAFT_REPO_ROOT
cp phyloligo_amd/libphyloligo_amd.so /tmp/orig.so
for v in "$@" orig; do
  if [ $v = orig ]; then cp /tmp/orig.so phyloligo_amd/libphyloligo_amd.so; else cp tools/exp/lib$v.so phyloligo_amd/libphyloligo_amd.so; fi
  echo "== $v"; timeout -k 10 300 python ${PD_SCRIPT:-tools/pairdot_bench.py} 2>&1 | grep -E "KT|BC"
done
cp /tmp/orig.so phyloligo_amd/libphyloligo_amd.so
