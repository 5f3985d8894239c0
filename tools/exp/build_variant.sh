#!/bin/bash
# build_variant.sh NAME "-DSWITCH=1 ..."  ->  tools/exp/variants/libNAME.so (own object directory; the product library is
# never touched).  Runs in the build container (hipcc cross-compiles); the variants travel to the GPU box with the snapshot.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
NAME=$1; shift
mkdir -p "$ROOT/tools/exp/variants"
make -C "$ROOT/phyloligo_amd/csrc" -j8 OBJDIR="$ROOT/tools/exp/variants/obj_$NAME" OUT="$ROOT/tools/exp/variants/lib$NAME.so" EXTRA="$*" 2>&1 | grep -E "error|warning: v|Error" || true
ls -la "$ROOT/tools/exp/variants/lib$NAME.so"
