#!/bin/bash
# BASELINE configs[4] at its TRUE size on one rank (rank 0 of 8 of the 10 M-entity / 100 M-triple / 1 k-relation / dim-512 layer),
# under rocprofv3 --kernel-trace --stats; run on the GPU box from the repo root. Outputs under gpurun_out/prof_scale_full/.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_scale_full
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 tools/bench_scale_shard.py ${1:-10000000} ${2:-100000000} 1000 512 512,200 8 0 > "$OUT/scale_full.json" 2> "$OUT/scale_full.log"
rc=$?
tail -20 "$OUT/scale_full.log"
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*kernel_stats.csv" | head; cat "$OUT/scale_full.json" | tail -1 | cut -c1-600
exit $rc
