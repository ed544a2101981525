#!/bin/bash
# A/B of alternative builds of the library on the bench step (not product code): tools/ab_libs.sh lib1.so lib2.so ...
for lib in "$@"; do
  for rep in 1 2; do
    echo "$lib rep $rep: $(MGCN_LIB=$PWD/kgc-gcn_amd/csrc/$lib python bench.py --steps 100 --warmup 10 --no-eval --no-cpu-baseline --no-scale --no-fb 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['ms_per_step'],4), {k:round(v['us'],1) for k,v in j['kernels'].items() if 'fused' in k})")"
  done
done
