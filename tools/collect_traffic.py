#!/usr/bin/env python3
"""Turns two rocprofv3 counter-collection CSVs (one `--pmc FETCH_SIZE` pass, one
`--pmc WRITE_SIZE` pass of the same bench.py command) into the per-kernel HBM traffic
file bench.py reads its `roofline.traffic` from.

    python tools/collect_traffic.py FETCH.csv WRITE.csv OUT.json

FETCH_SIZE / WRITE_SIZE are in kilobytes (MI355X_MICROARCH.md); on gfx950 FETCH_SIZE
counts half of what a streaming read moves (calibrated on k_chunk_stats, which reads
every frame exactly once), hence traffic = 2 x FETCH + WRITE."""
import csv
import json
import re
import sys


def per_kernel(path, counter):
    tot, launches = {}, {}
    with open(path, newline='') as f:
        for row in csv.DictReader(f):
            if row.get('Counter_Name') != counter:
                continue
            name = re.sub(r'^void\s+', '', row['Kernel_Name'])
            name = re.sub(r'^spkd::', '', name).split('(')[0].split('<')[0]
            tot[name] = tot.get(name, 0.0) + float(row['Counter_Value'])
            launches.setdefault(name, set()).add(row['Dispatch_Id'])
    return {k: (v, len(launches[k])) for k, v in tot.items()}


def main():
    fetch = per_kernel(sys.argv[1], 'FETCH_SIZE')
    write = per_kernel(sys.argv[2], 'WRITE_SIZE')
    out = {'unit': 'bytes per launch', 'formula': '2 * FETCH_SIZE + WRITE_SIZE (both reported in KB)',
           'kernels': {}}
    for k in sorted(set(fetch) | set(write)):
        fv, fl = fetch.get(k, (0.0, 1))
        wv, wl = write.get(k, (0.0, 1))
        fb, wb = 1024.0 * fv / max(fl, 1), 1024.0 * wv / max(wl, 1)
        out['kernels'][k] = {'launches_fetch_pass': fl, 'launches_write_pass': wl,
                             'fetch_size_bytes_per_launch': int(fb), 'write_size_bytes_per_launch': int(wb),
                             'hbm_bytes_per_launch_fetch_doubled': int(2 * fb + wb)}
    with open(sys.argv[3], 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    for k, v in out['kernels'].items():
        print('%-20s %8.2f GB' % (k, v['hbm_bytes_per_launch_fetch_doubled'] / 1e9))


if __name__ == '__main__':
    main()
