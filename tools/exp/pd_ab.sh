cd $GRAFT_REPO_ROOT
cp phyloligo_amd/libphyloligo_amd.so /tmp/orig.so
for v in pdd_2 pdd_4 pdd_8 orig; do
  if [ $v = orig ]; then cp /tmp/orig.so phyloligo_amd/libphyloligo_amd.so; else cp tools/exp/lib$v.so phyloligo_amd/libphyloligo_amd.so; fi
  echo "== $v"; timeout -k 10 200 python tools/exp/kt_only.py 2>&1 | grep KT
done
cp /tmp/orig.so phyloligo_amd/libphyloligo_amd.so
