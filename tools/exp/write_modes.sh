#!/bin/bash
# tools/ubench/container_write_modes on the GPU box's /tmp (the filesystem bench.py and the tests write containers to)
cd "${GRAFT_REPO_ROOT:-.}/tools/ubench" && g++ -O2 -std=c++17 -o container_write_modes container_write_modes.cpp -lpthread || exit 1
for cfg in "2 4" "2 8" "1 8" "8 2"; do
  for m in pwrite mmap mmap+pop big; do ./container_write_modes /tmp/cw.bin 30000 $cfg $m; done
done
