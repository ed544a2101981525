#!/bin/sh
# A/B of library builds on the bench workload (not product code): sh tools/ab_libs.sh libA.so libB.so [rounds] [shape] [zipf]
# e.g. sh tools/ab_libs.sh libmgcn_hip_old.so libmgcn_hip.so 3 fb15k237 1.1
cd "$(dirname "$0")/.."
A=$1; B=$2; N=${3:-3}; SHAPE=${4:-wn18rr}; ZIPF=${5:-0}
for i in $(seq $N); do for L in $A $B; do echo -n "$L "; MGCN_LIB=$PWD/kgc-gcn_amd/csrc/$L timeout -k 10 120 python tools/ab_fused2.py $SHAPE $ZIPF 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print([(round(l['us'],1), round(l['two_launch_us'],1), '%.1e' % l['max_abs_vs_two_launch'], l['rel_bit_equal']) for l in d['layers']])"; done; done
