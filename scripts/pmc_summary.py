#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSVs: per kernel (short name), mean counter value
per dispatch and mean duration from the kernel trace of the same pass."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1]
pat = re.compile(r'k_rowpass|k_colpass|k_update_pack_H|k_sum_partials|k_loss_from_parts|k_tile_V|k_wrule_slabs|k_w8_from_wb|k_post|k_slab_sum')


def short(name):
    m = re.search(r'k_[a-zA-Z_]+\d?', name)
    base = m.group(0) if m else name[:40]
    if 'k_rowpass' in name:
        mm = re.search(r'k_rowpass4?<([\d, ]+)>', name)
        if mm:
            a = [x.strip() for x in mm.group(1).split(',')]
            base += '<KT=%s,odd=%s,mode=%s%s%s>' % (a[0], a[1], a[2], ',fp8 tiles' if len(a) > 6 and a[6] in ('1', '2') else '',
                                                     ',column-split' if len(a) > 5 and a[5] == '1' else '')
        else:
            mm = re.search(r'k_rowpass4?ILi(\d+)ELi(\d+)ELi(\d+)', name)
            if mm:
                base += '<KT=%s,odd=%s,mode=%s>' % mm.groups()
    if 'k_colpass_q2' in name:
        mm = re.search(r'k_colpass_q2<([\d, ]+)>', name)
        if mm:
            a = [x.strip() for x in mm.group(1).split(',')]
            if len(a) > 3 and a[3] == '1':
                base += '<fp8 tiles>'
    return base


vals = defaultdict(lambda: defaultdict(list))
durs = defaultdict(list)
for f in sorted(glob.glob(os.path.join(root, 'p*', '**', '*counter_collection.csv'), recursive=True)):
    for row in csv.DictReader(open(f)):
        name = row.get('Kernel_Name', '')
        if not pat.search(name):
            continue
        vals[short(name)][row['Counter_Name']].append(float(row['Counter_Value']))
for f in sorted(glob.glob(os.path.join(root, 'p*', '**', '*kernel_trace.csv'), recursive=True)):
    for row in csv.DictReader(open(f)):
        name = row.get('Kernel_Name', '')
        if not pat.search(name):
            continue
        durs[short(name)].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e6)

for k in sorted(vals):
    d = durs.get(k, [])
    print('== %s   dispatches/pass~%d   mean duration under PMC %.3f ms' % (
        k, len(d) // max(1, len(glob.glob(os.path.join(root, 'p*.log')))), sum(d) / max(1, len(d))))
    for cname in sorted(vals[k]):
        v = vals[k][cname]
        print('   %-32s %18.1f  (n=%d)' % (cname, sum(v) / len(v), len(v)))

# HBM traffic of the dominant kernel per launch, corrected as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of a wide
# coalesced stream (128-B requests tallied at 64 B) -> doubled; WRITE_SIZE is exact.
import json
traffic = {}
for kname in vals:
    if 'k_rowpass' in kname and 'mode=0' in kname or kname.startswith('k_colpass') or kname.startswith('k_wrule_slabs') or kname.startswith('k_w8') or kname.startswith('k_post'):
        fs = vals[kname].get('FETCH_SIZE'); ws = vals[kname].get('WRITE_SIZE')
        if fs and ws:
            traffic[kname] = {'fetch_bytes_corrected': 2 * 1024 * sum(fs) / len(fs),
                              'write_bytes': 1024 * sum(ws) / len(ws),
                              'hbm_bytes_per_launch': 2 * 1024 * sum(fs) / len(fs) + 1024 * sum(ws) / len(ws)}
            # the shader clock the kernel held: GRBM_GUI_ACTIVE counts busy cycles summed over the 8 XCDs (an in-run cycle
            # count, unlike sysfs clocks sampled from outside), divided by the kernel's mean duration under the PMC passes
            ga, d = vals[kname].get('GRBM_GUI_ACTIVE'), durs.get(kname, [])
            if ga and d:
                traffic[kname]['sclk_mhz_from_grbm_gui_active'] = (sum(ga) / len(ga)) / 8.0 / (sum(d) / len(d) * 1e-3) / 1e6
                traffic[kname]['mean_duration_ms_under_pmc'] = sum(d) / len(d)
json.dump(traffic, open(os.path.join(root, 'traffic.json'), 'w'), indent=1)
print('traffic', json.dumps(traffic))
