#!/bin/bash
# After `bash tools/profile_round.sh rNN` on the GPU box (results merged back under gpurun_out/prof_rNN/): copy the summaries
# that are judged into profiles/ (tracked).  traffic.json / pmc_busy.json are what bench.py quotes; both carry the source hash
# of the library they were measured on (tests/test_host_cpu.py compares it with the tree).
tag=${1:-r05}; src=gpurun_out/prof_$tag
cd "$(dirname "$0")/.."
for f in ${tag}_bench_kernel_stats.txt ${tag}_pmc_busy.txt ${tag}_pmc_traffic.txt traffic.json pmc_busy.json; do cp -v $src/$f profiles/$f || exit 1; done
python3 - "$src/bench.json" "profiles/${tag}_bench_jsd_n50000_under_rocprof.json" <<'PY'
import json, sys
line = [ln for ln in open(sys.argv[1]) if ln.startswith("{")][-1]
json.dump(json.loads(line), open(sys.argv[2], "w"), indent=1)
print("wrote", sys.argv[2])
PY
echo "tree source hash: $(python3 tools/source_hash.py)"; grep -o '"src_hash": "[0-9a-f]*"' profiles/traffic.json; grep -o '"_src_hash": "[0-9a-f]*"' profiles/pmc_busy.json
