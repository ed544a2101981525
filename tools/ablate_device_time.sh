#!/bin/bash
# Device-side duration of the lockstep fused launch under each ablation mask (diagnostics build): rocprofv3 kernel trace,
# one process per (shape, mask) so that the per-kernel average is that mask's. Eager timing loops are host-bound below
# ~25 us per launch (Python + hipFuncSetAttribute), which is why the skeleton must be read from the device side.
#   bash tools/ablate_device_time.sh "0 3 7 4 1 2" "100x200 200x200"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp MGCN_LIB=$PWD/kgc-gcn_amd/csrc/libmgcn_hip_diag.so
OUT=$PWD/gpurun_out/ablate_dev
rm -rf "$OUT"; mkdir -p "$OUT"
for dims in ${2:-100x200 200x200}; do
  for m in ${1:-0 3 7}; do
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${dims}_$m" -o t -- python3 tools/fused_ablate.py --masks $m --dims $dims > /dev/null 2> "$OUT/${dims}_$m.log"
    f=$(find "$OUT/${dims}_$m" -name "*kernel_stats.csv" | head -1)
    echo "$dims mask $m: $(grep layer_fused2_kernel "$f" | head -1 | awk -F, '{print $2" calls, avg ns "$4}')"
  done
done
