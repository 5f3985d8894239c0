cd $GRAFT_REPO_ROOT
cp phyloligo_amd/libphyloligo_amd.so /tmp/orig.so
for v in nont orig nont orig; do
  if [ $v = orig ]; then cp /tmp/orig.so phyloligo_amd/libphyloligo_amd.so; else cp tools/exp/lib$v.so phyloligo_amd/libphyloligo_amd.so; fi
  echo "== $v: $(timeout -k 5 200 python tools/exp/nt_ab.py 2>&1 | grep JSD)"
done
cp /tmp/orig.so phyloligo_amd/libphyloligo_amd.so
