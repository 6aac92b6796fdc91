#!/usr/bin/env python
"""Turns the counter_collection CSVs of two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of
`bench.py --steps 1 --warmup 1 --no-cpu-baseline` into profiles/<tag>_pmc_hbm.md and
profiles/pmc_traffic.json (read by bench.py for roofline.traffic).

usage: python profiles/summarize_pmc.py gpurun_out/prof3 r01 [--batch 4096 --config cfg2]

Corrections (MI355X_MICROARCH.md, HBM section): counter unit is KiB; on gfx950 FETCH_SIZE counts a
128-B request as 64 B for 16 B/lane streaming reads (all streams here are 16 B/lane) -> doubled;
WRITE_SIZE is exact for 16 B/lane stores -> as is."""
import collections
import csv
import json
import os
import statistics
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def main():
    d, tag = sys.argv[1], sys.argv[2]
    batch = int(sys.argv[sys.argv.index('--batch') + 1]) if '--batch' in sys.argv else 4096
    config = sys.argv[sys.argv.index('--config') + 1] if '--config' in sys.argv else 'cfg2'
    data = {}
    for nm in ('fetch', 'write'):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(os.path.join(d, f'{nm}_counter_collection.csv'))):
            if 'pmg_' in r['Kernel_Name']:
                agg[r['Kernel_Name']].append(float(r['Counter_Value']) * 1024 / 1e9)
        data[nm] = agg
    med = statistics.median
    rows = []
    # the forward kernel is instantiated per mode: pmg_fwd_kernel<UPW, TPP, MODE, MULTI>, MODE 0 = decode,
    # 1 = teacher-forced (older builds: one instantiation, told apart by the bytes written)
    fwd = [k for k in data['write'] if 'pmg_fwd' in k]
    if len(fwd) > 1:
        for k in fwd:
            mode = int(k.split('<')[1].split('>')[0].split(',')[-2])     # <..., MODE, MULTI>: second to last
            name = {0: 'decode', 1: 'teacher_fwd'}.get(mode)
            if name:
                rows.append((name, f'pmg_fwd_kernel ({name})', len(data['fetch'][k]), med(data['fetch'][k]),
                             med(data['write'][k])))
    else:
        fw = fwd[0]
        wv, fv = data['write'][fw], data['fetch'][fw]
        thr = (min(wv) + max(wv)) / 2
        for name, idx in (('decode', [i for i, v in enumerate(wv) if v < thr]),
                          ('teacher_fwd', [i for i, v in enumerate(wv) if v >= thr])):
            rows.append((name, 'pmg_fwd_kernel (' + name + ')', len(idx), med([fv[i] for i in idx]),
                         med([wv[i] for i in idx])))
    wg_f = wg_w = 0.0
    for k in data['fetch']:
        if 'pmg_fwd' in k or 'pack' in k or 'aux' in k:
            continue
        f, w = med(data['fetch'][k]), med(data['write'].get(k, [0.0]))
        short = k.replace('(anonymous namespace)::', '').replace('void ', '')[:64]
        if 'pmg_bwd' in k:
            rows.append(('bptt', short, len(data['fetch'][k]), f, w))
        else:
            wg_f += f
            wg_w += w
            rows.append((None, short, len(data['fetch'][k]), f, w))
    rows.append(('wgrad', 'wgrad total (3 GEMM variants + reduce + finalize)', 0, wg_f, wg_w))
    lines = [f'# {tag} HBM traffic per launch (rocprofv3 --pmc, separate passes), {config} B={batch}', '',
             'Command per pass: `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` `--output-format csv -- python3 '
             'bench.py --steps 1 --warmup 1 --no-cpu-baseline`', '',
             'FETCH_SIZE doubled (gfx950 correction for 16 B/lane reads), WRITE_SIZE as is; GB = 1e9 B; medians '
             'over the launches of the run.', '',
             '| kernel | launches | FETCH_SIZE raw GB | read GB (x2) | written GB | total GB |', '|---|---|---|---|---|---|']
    from bench import kernel_source_hash
    # the hash ties the figures to the kernel sources they were measured on: bench.py reports
    # roofline.traffic only while the sources are unchanged
    js = {'config': config, 'batch': batch, 'tag': tag, 'kernel_sources_sha256': kernel_source_hash(), 'kernels': {}}
    for key, short, n, f, w in rows:
        lines.append(f'| {short} | {n or ""} | {f:.2f} | {2 * f:.2f} | {w:.2f} | {2 * f + w:.2f} |')
        if key:
            js['kernels'][key] = {'read_GB': round(2 * f, 3), 'write_GB': round(w, 3), 'total_GB': round(2 * f + w, 3)}
    open(os.path.join(HERE, f'{tag}_pmc_hbm.md'), 'w').write('\n'.join(lines) + '\n')
    out_json = 'pmc_traffic.json' if config == 'cfg2' else f'pmc_traffic_{config}.json'
    json.dump(js, open(os.path.join(HERE, out_json), 'w'), indent=1)
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
