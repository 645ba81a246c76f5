"""Condense the rocprofv3 --pmc passes of scripts/r02_profiles.sh: per kernel, mean of each counter per dispatch, the
HBM traffic per launch (gfx950 correction: FETCH_SIZE x 2 for wide streaming reads incl. LDS-DMA, MI355X_MICROARCH.md
section HBM; WRITE_SIZE exact; both in KB) and the matrix-pipe duty derived from the SQ pass.
usage: pmc_summary.py <dir with pmc_fetch/ pmc_write/ pmc_sq/ ...>   -> text on stdout + <dir>/pmc_summary.json"""
import collections
import csv
import glob
import json
import os
import sys


def short(name):
    name = name.replace('void rgp::', '').replace('rgp::', '').replace('unsigned short', 'bf16')
    return name.split('(')[0][:110]


def main():
    root = sys.argv[1]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, 'pmc_*', '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
            if r['Counter_Name'] in ('FETCH_SIZE', 'SQ_WAVE_CYCLES'):
                dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6)
    out = {}
    for k, cs in sorted(agg.items(), key=lambda kv: -sum(dur.get(kv[0], [0]))):
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        e = {'dispatches': max(len(v) for v in cs.values()), 'ms_mean_under_pmc': round(sum(dur[k]) / max(len(dur[k]), 1), 4)}
        e.update({c: m[c] for c in sorted(m)})
        if 'FETCH_SIZE' in m and 'WRITE_SIZE' in m:
            e['hbm_bytes_per_launch'] = (2.0 * m['FETCH_SIZE'] + m['WRITE_SIZE']) * 1024.0
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in m and m.get('GRBM_GUI_ACTIVE', 0) > 0:
            # GRBM_GUI_ACTIVE sums the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES sums busy cycles over all SIMDs (256 CUs x 4)
            e['mfma_pipe_duty'] = m['SQ_VALU_MFMA_BUSY_CYCLES'] / (m['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0)
        out[k] = e
    # which build these counters belong to (bench.py reports them as stale once a kernel's sources have changed)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from recurrent_gaze_prediction_amd import _lib
    out['_meta'] = {'kernel_source_hashes': _lib.kernel_source_hashes()}
    json.dump(out, open(os.path.join(root, 'pmc_summary.json'), 'w'), indent=1)
    for k, e in [kv for kv in out.items() if kv[0] != '_meta'][:24]:
        print(k)
        for c, v in e.items():
            print('    %-30s %s' % (c, ('%.6g' % v) if isinstance(v, float) else v))


if __name__ == '__main__':
    main()
