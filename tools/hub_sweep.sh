for thr in 64 48 32 24 16; do for chk in 64 32; do
  echo "thr=$thr chk=$chk $(MGCN_HUB_THRESHOLD=$thr MGCN_HUB_CHUNK=$chk python bench.py --shape fb15k237 --zipf 1.1 --steps 50 --warmup 10 --no-eval --no-cpu-baseline --no-scale --no-fb 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['ms_per_step'],4), {k:round(v['us'],1) for k,v in j['kernels'].items() if 'fused' in k or 'hub' in k})")"
done; done
