#!/usr/bin/env python3
"""Condenses rocprofv3 output directories into the small CSVs kept under profiles/.

    python tools/summarize_rocprof.py stats  <dir with *_kernel_stats.csv>        > profiles/rNN_kernel_stats.csv
    python tools/summarize_rocprof.py pmc    <FETCH_SIZE dir> <WRITE_SIZE dir>    > profiles/rNN_pmc_traffic.csv
    python tools/summarize_rocprof.py sq     <dir of one --pmc SQ_... pass>       > profiles/rNN_sq_counters.csv

PMC traffic follows MI355X_MICROARCH.md (HBM / rocprofv3): FETCH_SIZE and WRITE_SIZE are collected in separate passes and
are reported in KiB; on gfx950 FETCH_SIZE counts wide coalesced reads at half their bytes, so read bytes = 2 x FETCH_SIZE;
WRITE_SIZE is exact for 16-byte-per-lane stores."""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r'\(.*', '', name)
    m = re.match(r'_Z\d+([a-z_0-9]+kernel)', name)
    return (m.group(1) if m else name.replace('void ', ''))[:48] + ('<' + name.split('I', 1)[1][:60] if name.startswith('_Z') and 'I' in name else '')


def stats(d):
    f = max(glob.glob(d + '/**/*_kernel_stats.csv', recursive=True), key=os.path.getmtime)     # newest run in the directory
    w = csv.writer(sys.stdout)
    w.writerow(['kernel', 'calls', 'total_ns', 'avg_ns', 'percent', 'min_ns', 'max_ns'])
    for r in csv.DictReader(open(f)):
        w.writerow([r['Name'][:160], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'], r['MinNs'], r['MaxNs']])


def pmc(df, dw):
    def load(d):
        f = max(glob.glob(d + '/**/*_counter_collection.csv', recursive=True), key=os.path.getmtime)
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[r['Kernel_Name']].append(float(r['Counter_Value']))
        return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}
    fe, n = load(df)
    wr, _ = load(dw)
    w = csv.writer(sys.stdout)
    w.writerow(['kernel', 'dispatches', 'FETCH_SIZE_KiB_avg', 'WRITE_SIZE_KiB_avg', 'hbm_bytes_per_launch(2*fetch+write)'])
    for k in sorted(fe, key=lambda k: -fe[k] * n[k]):
        w.writerow([k[:160], n[k], round(fe[k], 1), round(wr.get(k, 0.0), 1), int((2 * fe[k] + wr.get(k, 0.0)) * 1024)])


def sq(d):
    """Per kernel: the average of every counter of the pass over its dispatches; last column = matrix-pipe busy cycles / (active cycles
    x 1024 SIMDs) when both counters are present (GRBM_GUI_ACTIVE is summed over the 8 XCDs)."""
    f = max(glob.glob(d + '/**/*_counter_collection.csv', recursive=True), key=os.path.getmtime)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    names = sorted({c for k in agg for c in agg[k]})
    w = csv.writer(sys.stdout)
    w.writerow(['kernel', 'dispatches'] + names + ['mfma_busy_frac_of_chip(MFMA_BUSY/(GUI_ACTIVE/8*1024))'])
    def tot(k):
        v = agg[k].get('SQ_WAVE_CYCLES') or [0.0]
        return -sum(v)
    for k in sorted(agg, key=tot):
        n = max(len(v) for v in agg[k].values())
        avg = {c: (sum(v) / len(v) if v else 0.0) for c, v in agg[k].items()}
        frac = ''
        if avg.get('GRBM_GUI_ACTIVE') and 'SQ_VALU_MFMA_BUSY_CYCLES' in avg:
            frac = round(avg['SQ_VALU_MFMA_BUSY_CYCLES'] / (avg['GRBM_GUI_ACTIVE'] / 8 * 1024), 4)
        w.writerow([k[:160], n] + [round(avg.get(c, 0.0), 1) for c in names] + [frac])


if __name__ == '__main__':
    if sys.argv[1] == 'stats':
        stats(sys.argv[2])
    elif sys.argv[1] == 'sq':
        sq(sys.argv[2])
    else:
        pmc(sys.argv[2], sys.argv[3])
