cd "$(dirname "$0")/.."
for i in 1 2; do for L in libmgcn_hip_old.so libmgcn_hip.so; do echo -n "$L "; MGCN_LIB=$PWD/kgc-gcn_amd/csrc/$L timeout -k 10 120 python tools/ab_fused2.py fb15k237 1.1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print([(round(l['us'],1), round(l['two_launch_us'],1), '%.1e' % l['max_abs_vs_two_launch']) for l in d['layers']])"; done; done
