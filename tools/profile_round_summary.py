"""Summarise gpurun_out/prof_<round> (tools/profile_round.sh) into the tracked files under profiles/:
<round>_kernel_stats_{wn18rr_2layer,wn18rr_1layer,fb15k237,eval,scale_shard}.csv (rocprofv3 --stats kernel summaries),
<round>_scale_shard.json (the configs[4] slice, 512 -> 512 and 512 -> 200) and <round>_traffic.json
(PMC FETCH_SIZE / WRITE_SIZE per fused launch, corrected by the calibration pass, with the kernel-source fingerprint bench.py
checks before quoting it)."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def one(pattern):
    hits = sorted(glob.glob(pattern, recursive=True))
    if not hits:
        raise SystemExit('missing ' + pattern)
    return hits[0]


def counter_per_dispatch(d, kernel_substr):
    rows = list(csv.DictReader(open(one(os.path.join(d, '**', '*counter_collection.csv')))))
    vals = [(int(r['Dispatch_Id']), float(r['Counter_Value'])) for r in rows if kernel_substr in r['Kernel_Name']]
    vals.sort()
    return [v for _, v in vals]


def main():
    out = sys.argv[1]
    rnd = sys.argv[2] if len(sys.argv) > 2 else 'r04'
    prof = os.path.join(ROOT, 'profiles')
    traffic_only = '--traffic-only' in sys.argv     # (the PMC passes run first, so that the traced bench lines can quote the file)
    for tag, name in () if traffic_only else (('trace_l2', 'wn18rr_2layer'), ('trace_l1', 'wn18rr_1layer'), ('trace_fb', 'fb15k237'), ('trace_eval', 'eval'),
                      ('trace_scale', 'scale_shard')):
        shutil.copy(one(os.path.join(out, tag, '**', '*kernel_stats.csv')), os.path.join(prof, '%s_kernel_stats_%s.csv' % (rnd, name)))
    for tag, name in () if traffic_only else (('bench_l2', 'wn18rr_2layer'), ('bench_l1', 'wn18rr_1layer'), ('bench_fb', 'fb15k237'), ('bench_eval', 'eval')):
        line = open(os.path.join(out, tag + '.json')).read().strip().splitlines()[-1]
        json.dump(json.loads(line), open(os.path.join(prof, '%s_bench_under_rocprof_%s.json' % (rnd, name)), 'w'), indent=1)
    if not traffic_only:
        shard = {}
        for tag in ('scale_512', 'scale_200'):
            shard[tag] = json.loads(open(os.path.join(out, tag + '.json')).read().strip().splitlines()[-1])
        shard['note'] = 'tools/bench_scale_shard.py: rank 0 of 8 of a 2M-entity / 20M-triple / dim-512 graph, the rank holding only its ' \
                        '10.2 GB shard of the 81.9 GB table; scale_512 ran under rocprofv3 --kernel-trace (%s_kernel_stats_scale_shard.csv)' % rnd
        json.dump(shard, open(os.path.join(prof, '%s_scale_shard.json' % rnd), 'w'), indent=1)
    # calibration: tools/pmc_calibrate.py's 512 MiB streaming copy = the kernel with the largest WRITE_SIZE (exactly
    # 524 288 KiB); FETCH_SIZE of the same kernel gives the read scale (gfx950 counts half of a wide streaming read)
    def by_name(d):
        rows = list(csv.DictReader(open(one(os.path.join(d, '**', '*counter_collection.csv')))))
        acc = {}
        for r in rows:
            acc.setdefault(r['Kernel_Name'], []).append(float(r['Counter_Value']))
        return acc
    cw, cf = by_name(os.path.join(out, 'pmc_cal_w')), by_name(os.path.join(out, 'pmc_cal_f'))
    copy_name = max((k for k in cw if max(cw[k]) >= 500000.0), key=lambda k: max(cf.get(k, [0.0])))   # writes AND reads 512 MiB
    cal_w, cal_f = cw[copy_name], cf.get(copy_name, [])
    write_scale = 512.0 * 1024 / max(cal_w)
    fetch_scale = 512.0 * 1024 / max(cal_f) if cal_f else 2.0
    assert 0.9 < write_scale < 1.1 and 1.8 < fetch_scale < 2.2, (copy_name, max(cal_w), cal_f[:3])
    res = {'source': 'tools/profile_round.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) '
                     'on `python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-eval --no-fb --no-scale [--shape fb15k237 --zipf 1.1]`',
           'calibration': {'copy_kernel': copy_name[:60], 'copy_512MiB_FETCH_SIZE_KiB': max(cal_f) if cal_f else None, 'copy_512MiB_WRITE_SIZE_KiB': max(cal_w) if cal_w else None,
                           'fetch_scale': fetch_scale, 'write_scale': write_scale,
                           'note': 'counters are KiB; FETCH_SIZE counts half of a wide streaming read on gfx950 (MI355X_MICROARCH.md, HBM)'},
           'note': 'fabric-side bytes per launch (Infinity-Cache hits are counted, not excluded)',
           'source_fingerprint': bench.source_fingerprint()}
    for shape, suffix in (('wn18rr', ''), ('fb15k237', '_fb')):
        f = counter_per_dispatch(os.path.join(out, 'pmc_fetch' + suffix), 'layer_fused')
        w = counter_per_dispatch(os.path.join(out, 'pmc_write' + suffix), 'layer_fused')
        n = min(len(f), len(w)) // 2 * 2
        f, w = f[-n:], w[-n:]                                      # launches alternate layer 1, layer 2
        res[shape] = {}
        for li in (0, 1):
            rb = 1024.0 * fetch_scale * sum(f[li::2]) / len(f[li::2])
            wb = 1024.0 * write_scale * sum(w[li::2]) / len(w[li::2])
            res[shape]['layer_fused_l%d' % (li + 1)] = {'read_bytes': rb, 'write_bytes': wb, 'traffic_bytes': rb + wb, 'launches': len(f[li::2])}
    json.dump(res, open(os.path.join(prof, '%s_traffic.json' % rnd), 'w'), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
