#!/bin/bash
# ab.sh "<command>" VARIANT...   run <command> once per variant (tools/exp/variants/libVARIANT.so, through PO_LIB_PATH) and
# once with the product library ("orig"), on the GPU box.  Nothing is copied over the installed library.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
CMD=$1; shift
for v in "$@" orig; do
  echo "== $v"
  if [ "$v" = orig ]; then env -u PO_LIB_PATH timeout -k 10 400 bash -c "$CMD" 2>&1 | grep -v "PO_LIB_PATH set"
  else PO_ALLOW_VARIANT=1 PO_LIB_PATH=tools/exp/variants/lib$v.so timeout -k 10 400 bash -c "$CMD" 2>&1 | grep -v "PO_LIB_PATH set"; fi
done
true
