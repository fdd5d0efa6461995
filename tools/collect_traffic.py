#!/usr/bin/env python3
"""Turns two rocprofv3 counter-collection CSVs (one `--pmc FETCH_SIZE` pass, one
`--pmc WRITE_SIZE` pass of the same bench.py command, tools/profile_bench.sh) into the
per-kernel HBM traffic file bench.py reads its `roofline.traffic` from.

    python tools/collect_traffic.py FETCH.csv WRITE.csv OUT.json [KERNEL_TRACE.csv]

Per kernel only the BATCH launches count -- the dispatches with the kernel's largest grid
(bench.py also runs one file through the file-based scripts to verify the batch result;
those small launches would spoil an average).  FETCH_SIZE / WRITE_SIZE are in kilobytes
(MI355X_MICROARCH.md); on gfx950 FETCH_SIZE counts half of what a streaming read moves
(calibrated in round 1 on k_chunk_stats, which reads every frame exactly once), hence
traffic = 2 x FETCH + WRITE.
With the `--kernel-trace` CSV of the same build as fourth argument the batch launches'
average duration is stored next to the bytes: bench.py only quotes a traffic figure while
the kernel it measures runs within 5 % of the duration recorded here."""
import csv
import json
import re
import sys


def short(name):
    name = re.sub(r'^void\s+', '', name)
    return re.sub(r'^spkd::', '', name).split('(')[0].split('<')[0]


def per_kernel(path, counter):
    """{kernel: (mean counter value over the batch launches, their number)}"""
    disp = {}
    with open(path, newline='') as f:
        for row in csv.DictReader(f):
            if row.get('Counter_Name') != counter:
                continue
            k = (short(row['Kernel_Name']), row['Dispatch_Id'])
            v, g = disp.get(k, (0.0, 0))
            disp[k] = (v + float(row['Counter_Value']), int(row['Grid_Size']))
    out = {}
    for name in set(k[0] for k in disp):
        rows = [v for (n, _), v in disp.items() if n == name]
        gmax = max(g for _, g in rows)
        big = [v for v, g in rows if g == gmax]
        out[name] = (sum(big) / len(big), len(big))
    return out


def kernel_ms(path):
    """{kernel: mean duration in ms of its batch launches} from a kernel-trace CSV"""
    rows = {}
    with open(path, newline='') as f:
        for row in csv.DictReader(f):
            grid = int(row['Grid_Size_X']) * int(row['Grid_Size_Y']) * int(row['Grid_Size_Z'])
            rows.setdefault(short(row['Kernel_Name']), []).append(
                (grid, (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e6))
    out = {}
    for name, r in rows.items():
        gmax = max(g for g, _ in r)
        big = [ms for g, ms in r if g == gmax]
        out[name] = (sum(big) / len(big), len(big))
    return out


def main():
    fetch = per_kernel(sys.argv[1], 'FETCH_SIZE')
    write = per_kernel(sys.argv[2], 'WRITE_SIZE')
    ms = kernel_ms(sys.argv[4]) if len(sys.argv) > 4 else {}
    out = {'unit': 'bytes per batch launch (the dispatches with the largest grid)',
           'formula': '2 * FETCH_SIZE + WRITE_SIZE (both reported in KB)', 'kernels': {}}
    for k in sorted(set(fetch) | set(write)):
        fv, fl = fetch.get(k, (0.0, 0))
        wv, wl = write.get(k, (0.0, 0))
        fb, wb = 1024.0 * fv, 1024.0 * wv
        out['kernels'][k] = {'launches_fetch_pass': fl, 'launches_write_pass': wl,
                             'fetch_size_bytes_per_launch': int(fb), 'write_size_bytes_per_launch': int(wb),
                             'hbm_bytes_per_launch_fetch_doubled': int(2 * fb + wb),
                             'kernel_ms': round(ms[k][0], 4) if k in ms else None,
                             'launches_timed': ms[k][1] if k in ms else 0}
    with open(sys.argv[3], 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    for k, v in out['kernels'].items():
        print('%-20s %8.2f GB  %s ms' % (k, v['hbm_bytes_per_launch_fetch_doubled'] / 1e9, v['kernel_ms']))


if __name__ == '__main__':
    main()
