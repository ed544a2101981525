#!/bin/bash
# Geometry sweep of the elastic kernel on the configs[4] slice (2M / 20M / dim 512; not product code)
for t in 0 0x13 0x23 0x33 0x2013 0x2023 0x123 0x133; do
  echo "tune=$t $(MGCN_FUSED_TUNE=$t python tools/bench_scale_shard.py 2000000 20000000 1000 512 512,200 8 0 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print([(r['O'], round(r['layer_ms'],3)) for r in j['runs']])" 2>&1 | tail -1)"
done
