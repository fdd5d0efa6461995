#!/usr/bin/env python3
"""SQ counters of the batch launches (largest grid per kernel) from a rocprofv3 `--pmc SQ_*`
counter-collection CSV (tools/profile_bench.sh pass 4) -> JSON.
    python tools/collect_sq.py SQ.csv OUT.json
SQ_WAVE_CYCLES = SQ_ACTIVE_INST_ANY + SQ_WAIT_ANY + SQ_WAIT_INST_ANY (quad-cycles,
MI355X_MICROARCH.md): issuing / parked in s_waitcnt or a barrier / issue-stalled."""
import collections
import csv
import json
import re
import sys


def short(name):
    name = re.sub(r'^void\s+', '', name)
    return re.sub(r'^spkd::', '', name).split('(')[0].split('<')[0]


def main():
    disp = collections.defaultdict(lambda: collections.defaultdict(float))
    grid = {}
    with open(sys.argv[1], newline='') as f:
        for row in csv.DictReader(f):
            k = (short(row['Kernel_Name']), row['Dispatch_Id'])
            disp[k][row['Counter_Name']] += float(row['Counter_Value'])
            grid[k] = int(row['Grid_Size'])
    out = {}
    for name in sorted(set(k[0] for k in disp)):
        gmax = max(g for k, g in grid.items() if k[0] == name)
        big = [disp[k] for k in disp if k[0] == name and grid[k] == gmax]
        d = {c: sum(b[c] for b in big) / len(big) for c in big[0]}
        wc = d.get('SQ_WAVE_CYCLES', 0.0)
        rec = {'launches': len(big)}
        rec.update({c: int(v) for c, v in d.items()})
        if wc:
            rec['share_issuing'] = round(d['SQ_ACTIVE_INST_ANY'] / wc, 3)
            rec['share_waiting'] = round(d['SQ_WAIT_ANY'] / wc, 3)
            rec['share_issue_stalled'] = round(d['SQ_WAIT_INST_ANY'] / wc, 3)
        out[name] = rec
    with open(sys.argv[2], 'w') as f:
        json.dump({'note': 'per batch launch; cycle counters in quad-cycles summed over waves',
                   'kernels': out}, f, indent=1, sort_keys=True)
    for k in ('k_gw', 'k_matrix', 'k_ahc'):
        if k in out:
            print(k, out[k])


if __name__ == '__main__':
    main()
