#!/bin/bash
# Memory-side counters of the batch launches (run on the MI355X box from the repository root):
#   tools/profile_memside.sh OUTDIR
# two --pmc passes (four TCC counters each): L2 hits / misses / requests, fabric read and write
# requests by size, cycles the L2's read interface was out of DRAM credits.
set -o pipefail
R=$(pwd)
OUT=$R/${1:-gpurun_out/memside}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-extras"
keep() {
    for f in $(find $1 -name "*counter_collection.csv"); do
        (head -1 $f; grep -E 'spkd::|k_gather_records' $f) > $f.small; mv $f.small $f
    done
}
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum --output-format csv -d $OUT/a -- python3 $R/bench.py $ARGS > $OUT/a.json 2> $OUT/a.err || exit 1
keep $OUT/a
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $OUT/b -- python3 $R/bench.py $ARGS > $OUT/b.json 2> $OUT/b.err || exit 1
keep $OUT/b
du -sh $OUT
