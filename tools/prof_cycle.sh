#!/bin/bash
# One AMR V-cycle by kernel: rocprofv3 kernel trace of tools/bench_amr.py, condensed by tools/big_kernels.py.
#   tools/prof_cycle.sh <config> <window_ms> <out.txt> [min_us]
# The program goes directly after "--" (no env/bash hop under the profiler).
set -e
cfg=$1; win=$2; out=$3; minus=${4:-100}
export TMPDIR=/tmp
d=gpurun_out/prof_$cfg
rm -rf $d
rocprofv3 --kernel-trace --stats -d $d -o $cfg --output-format csv -- python3 tools/bench_amr.py --config $cfg --steps 3 --warmup 1 > $d.log 2>&1
tail -1 $d.log
python3 tools/big_kernels.py $d $out $win $minus
python3 tools/prof_summary.py stats $d ${out%.txt}_stats.md || true
rm -rf $d
