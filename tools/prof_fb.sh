#!/bin/bash
# rocprofv3 kernel stats of the FB15k-237-shape step (not product code); outputs gpurun_out/prof_fb/
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_fb
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-eval --no-fb --no-scale --shape fb15k237 --zipf 1.1 > "$OUT/bench.json" 2> "$OUT/trace.log"
find "$OUT" -name "*kernel_trace.csv" -delete
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_fb/trace/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print('%-100s calls %6s avg %9.1f us  %5s%%' % (r['Name'][:100], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
PY
