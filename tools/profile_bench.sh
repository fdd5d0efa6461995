#!/bin/bash
# The rocprofv3 passes the committed profiles/ come from (run on the MI355X box from the
# repository root):  tools/profile_bench.sh OUTDIR
#   1. --kernel-trace --stats            -> per-kernel average durations
#   2. --pmc FETCH_SIZE, 3. --pmc WRITE_SIZE (separate passes)  -> fabric bytes per launch
#   4. --pmc SQ_* (one pass)             -> wave cycles: active / waiting / issue-stalled, VALU instructions
# The batch is generated on the device by torch (tens of thousands of small dispatches), so
# the per-dispatch CSVs are filtered down to this library's kernels before they are kept.
set -o pipefail
R=$(pwd)
OUT=$R/${1:-gpurun_out/profile}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-extras"
keep() {  # keep the header and the rows of spkd kernels
    for f in $(find $1 -name "*counter_collection.csv" -o -name "*kernel_trace.csv"); do
        (head -1 $f; grep -E 'spkd::|k_gather_records' $f) > $f.small; mv $f.small $f
    done
}
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $ARGS > $OUT/stats.json 2> $OUT/stats.err || exit 1
keep $OUT/stats
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $OUT/fetch.json 2> $OUT/fetch.err || exit 1
keep $OUT/fetch
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $OUT/write.json 2> $OUT/write.err || exit 1
keep $OUT/write
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/sq -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $OUT/sq.json 2> $OUT/sq.err || exit 1
keep $OUT/sq
du -sh $OUT
