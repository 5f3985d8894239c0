cd $GRAFT_REPO_ROOT
cp phyloligo_amd/libphyloligo_amd.so /tmp/orig.so
for v in s1 s2 s3; do
  cp tools/exp/libpd_$v.so phyloligo_amd/libphyloligo_amd.so
  echo "== stagger variant $v (TR=128)"; timeout -k 10 200 python tools/exp/kt_only.py 2>&1 | grep KT
done
cp /tmp/orig.so phyloligo_amd/libphyloligo_amd.so
