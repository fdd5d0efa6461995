#!/usr/bin/env python3
"""Where a kernel's scratch (spill) accesses sit relative to its loops.

    make -C speaker-diarization_amd/csrc spkd_hip.gfx950.s
    python tools/scratch_report.py speaker-diarization_amd/csrc/spkd_hip.gfx950.s k_gw

Prints, for the named kernel: lines, DPP instructions, scratch loads / stores, and for
every backward branch (a loop) the number of scratch instructions inside it.  The loop
that contains the elimination code (the DPP instructions) is the item loop."""
import bisect
import re
import sys


def main():
    path, name = sys.argv[1], sys.argv[2]
    s = open(path).read().splitlines()
    starts = [i for i, l in enumerate(s) if re.match(r'^_Z\w*%s\w*:' % re.escape(name), l) or l.startswith(name + ':')]
    if not starts:
        sys.exit('kernel %s not found' % name)
    start = starts[0]
    end = next(i for i in range(start, len(s)) if 's_endpgm' in s[i])
    body = [l.split(';')[0] for l in s[start:end]]
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
    dpp = [i for i, l in enumerate(body) if 'row_newbcast' in l]
    sload = [i for i, l in enumerate(body) if 'scratch_load' in l]
    sstore = [i for i, l in enumerate(body) if 'scratch_store' in l]
    print('%s: %d lines, %d DPP, %d scratch loads, %d scratch stores' % (name, len(body), len(dpp), len(sload), len(sstore)))
    loops = []
    for i, l in enumerate(body):
        m = re.search(r's_c?branch\w* (\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    for lo, hi in sorted(set(loops)):
        nd = sum(1 for k in dpp if lo <= k <= hi)
        nl = sum(1 for k in sload if lo <= k <= hi)
        ns = sum(1 for k in sstore if lo <= k <= hi)
        if hi - lo > 200 or nl or ns:
            print('  loop lines %6d..%6d (%5d instr): %5d DPP, %4d scratch loads, %4d scratch stores' % (lo, hi, hi - lo, nd, nl, ns))


if __name__ == '__main__':
    main()
