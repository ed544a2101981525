import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('value %.4g edges/s  step %.1f us' % (r["value"], r["ms_per_step"]*1e3))
for k,v in r["kernels"].items(): print('  %-14s %7.1f us  %s' % (k, v["us"], {a:round(b,3) for a,b in v.items() if "frac" in a}))
if 'eval' in r: print({k: round(v,6) for k,v in r['eval'].items() if not isinstance(v,str)})
