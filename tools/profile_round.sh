#!/bin/bash
# Profiles of bench.py for round $1 (default r04; run on the GPU box from the repo root): rocprofv3 --kernel-trace --stats of the bench
# command (2 layers and 1 layer, so the two fused launches can be told apart), the evaluation kernels, and the PMC
# passes (FETCH_SIZE / WRITE_SIZE in separate runs, --kernel-trace only, as MI355X_MICROARCH.md prescribes).
# Plus the configs[4] slice (one rank's 1/8 of a 2M-entity / 20M-triple / dim-512 layer: layer_fused3_kernel) under the
# same tracer. Results land under gpurun_out/prof_<round>/; tools/profile_round_summary.py turns them into profiles/<round>_*.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R=${1:-r04}
OUT=$PWD/gpurun_out/prof_$R
rm -rf "$OUT"; mkdir -p "$OUT"
B="--steps 100 --warmup 10 --no-cpu-baseline --no-eval --no-fb --no-scale"
# PMC passes first: profiles/<round>_traffic.json then carries the fingerprint of these sources and the traced bench lines quote it
P="--steps 10 --warmup 2 --no-cpu-baseline --no-eval --no-fb --no-scale"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o p -- python3 bench.py $P > /dev/null 2> "$OUT/pmc_fetch.log"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o p -- python3 bench.py $P > /dev/null 2> "$OUT/pmc_write.log"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch_fb" -o p -- python3 bench.py $P --shape fb15k237 --zipf 1.1 > /dev/null 2> "$OUT/pmc_fetch_fb.log"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write_fb" -o p -- python3 bench.py $P --shape fb15k237 --zipf 1.1 > /dev/null 2> "$OUT/pmc_write_fb.log"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_cal_f" -o p -- python3 tools/pmc_calibrate.py > /dev/null 2> "$OUT/pmc_cal_f.log"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_cal_w" -o p -- python3 tools/pmc_calibrate.py > /dev/null 2> "$OUT/pmc_cal_w.log"
python3 tools/profile_round_summary.py "$OUT" "$R" --traffic-only > /dev/null || echo "traffic summary failed"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_l2" -o t -- python3 bench.py $B > "$OUT/bench_l2.json" 2> "$OUT/trace_l2.log"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_l1" -o t -- python3 bench.py $B --layers 1 > "$OUT/bench_l1.json" 2> "$OUT/trace_l1.log"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_fb" -o t -- python3 bench.py $B --shape fb15k237 --zipf 1.1 > "$OUT/bench_fb.json" 2> "$OUT/trace_fb.log"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_eval" -o t -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fb --no-scale > "$OUT/bench_eval.json" 2> "$OUT/trace_eval.log"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_scale" -o t -- python3 tools/bench_scale_shard.py 2000000 20000000 1000 512 512 8 0 > "$OUT/scale_512.json" 2> "$OUT/trace_scale.log"
python3 tools/bench_scale_shard.py 2000000 20000000 1000 512 200 8 0 > "$OUT/scale_200.json" 2> "$OUT/scale_200.log"
python3 tools/profile_round_summary.py "$OUT" "$R" || { echo "summary failed"; tail -5 "$OUT"/*.log; }
# the raw per-dispatch traces are large (gpurun brings back at most 64 MiB): keep the summaries only
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*counter_collection.csv" -size +2M -delete
du -sh "$OUT"; ls "$OUT"
