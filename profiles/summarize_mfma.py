#!/usr/bin/env python
"""MFMA utilisation per kernel from the `mfma` pass of profiles/profile_all.sh
(rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE).

usage: python profiles/summarize_mfma.py gpurun_out/prof6 r01 [--label 'cfg4 B=4096']"""
import collections
import csv
import os
import statistics
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    d, tag = sys.argv[1], sys.argv[2]
    label = sys.argv[sys.argv.index('--label') + 1] if '--label' in sys.argv else 'cfg2 B=4096'
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(os.path.join(d, 'mfma_counter_collection.csv'))):
        k = r['Kernel_Name']
        if 'pmg_fwd' in k or 'pmg_bwd' in k or 'pmg_wgrad_kernel' in k:
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    med = statistics.median
    lines = [f'# {tag} MFMA counters per launch (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES '
             f'SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE), {label}', '',
             'MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8 XCDs) * 1024 SIMDs): GRBM_GUI_ACTIVE comes back '
             'summed over the 8 XCDs, the busy counter summed over all SIMDs.',
             'MFMA flop = SQ_INSTS_VALU_MFMA_MOPS_F32 * 512.  pmg_fwd_kernel<WV, UPW, TPP, MODE, MULTI>: MODE 0 = decode, '
             '1 = teacher-forced.', '',
             '| kernel | launches | MFMA busy cycles (sum over SIMDs) | GUI active cycles (sum over XCDs) | MfmaUtil | '
             'MFMA TFLOP per launch |', '|---|---|---|---|---|---|']
    for k, c in agg.items():
        busy, gui, mops = med(c['SQ_VALU_MFMA_BUSY_CYCLES']), med(c['GRBM_GUI_ACTIVE']), med(c['SQ_INSTS_VALU_MFMA_MOPS_F32'])
        short = k.replace('(anonymous namespace)::', '').replace('void ', '')
        lines.append(f'| {short} | {len(c["GRBM_GUI_ACTIVE"])} | {busy:.3e} | {gui:.3e} | {busy / (gui / 8 * 1024):.3f} | '
                     f'{mops * 512 / 1e12:.3f} |')
    open(os.path.join(HERE, f'{tag}_pmc_mfma.md'), 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
