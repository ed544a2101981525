#!/bin/bash
# Round-2 profiles of bench.py (run on the GPU box from the repo root): rocprofv3 --kernel-trace --stats of the bench
# command (2 layers and 1 layer, so the two fused launches can be told apart), the evaluation kernels, and the PMC
# passes (FETCH_SIZE / WRITE_SIZE in separate runs, --kernel-trace only, as MI355X_MICROARCH.md prescribes).
# Results land under gpurun_out/prof_r02/; tools/profile_r02_summary.py turns them into profiles/r02_*.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_r02
rm -rf "$OUT"; mkdir -p "$OUT"
B="--steps 100 --warmup 10 --no-cpu-baseline --no-eval --no-fb"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_l2" -o t -- python3 bench.py $B > "$OUT/bench_l2.json" 2> "$OUT/trace_l2.log"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_l1" -o t -- python3 bench.py $B --layers 1 > "$OUT/bench_l1.json" 2> "$OUT/trace_l1.log"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_fb" -o t -- python3 bench.py $B --shape fb15k237 --zipf 1.1 > "$OUT/bench_fb.json" 2> "$OUT/trace_fb.log"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_eval" -o t -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fb > "$OUT/bench_eval.json" 2> "$OUT/trace_eval.log"
P="--steps 10 --warmup 2 --no-cpu-baseline --no-eval --no-fb"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o p -- python3 bench.py $P > /dev/null 2> "$OUT/pmc_fetch.log"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o p -- python3 bench.py $P > /dev/null 2> "$OUT/pmc_write.log"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch_fb" -o p -- python3 bench.py $P --shape fb15k237 --zipf 1.1 > /dev/null 2> "$OUT/pmc_fetch_fb.log"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write_fb" -o p -- python3 bench.py $P --shape fb15k237 --zipf 1.1 > /dev/null 2> "$OUT/pmc_write_fb.log"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_cal_f" -o p -- python3 tools/pmc_calibrate.py > /dev/null 2> "$OUT/pmc_cal_f.log"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_cal_w" -o p -- python3 tools/pmc_calibrate.py > /dev/null 2> "$OUT/pmc_cal_w.log"
python3 tools/profile_r02_summary.py "$OUT" || { echo "summary failed"; tail -5 "$OUT"/*.log; }
# the raw per-dispatch traces are large (gpurun brings back at most 64 MiB): keep the summaries only
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*counter_collection.csv" -size +2M -delete
du -sh "$OUT"; ls "$OUT"
