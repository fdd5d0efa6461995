#!/bin/bash
# Same-box A/B of library builds:  tools/ab_bench.sh OUTDIR NAME [NAME ...]
# runs bench.py (no CPU baseline, no extras) once per build, alternating twice, with
# SPKD_HIP_LIBRARY pointing at speaker-diarization_amd/csrc/libspkd_hip_NAME.so
# ("main" = libspkd_hip.so), and prints throughput and per-kernel milliseconds.
R=$(pwd)
OUT=$R/$1; shift
mkdir -p $OUT
for rep in 1 2; do
  for n in "$@"; do
    lib=$R/speaker-diarization_amd/csrc/libspkd_hip_$n.so
    [ "$n" = main ] && lib=$R/speaker-diarization_amd/csrc/libspkd_hip.so
    SPKD_HIP_LIBRARY=$lib timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --no-extras > $OUT/$n.$rep.json 2> $OUT/$n.$rep.err || exit 1
    python3 -c "
import json,sys; d=json.load(open('$OUT/$n.$rep.json')); print('%-10s rep $rep  %.1f h/s  %.1f ms/step ' % ('$n', d['value'], d['ms_per_step']), {k: round(v['ms_per_launch'],2) for k,v in d['kernels'].items()})"
  done
done
