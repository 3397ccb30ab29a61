#!/bin/bash
# The measurement pass behind profiles/r03_final_*: bench.py (the driver's command), its rocprofv3 kernel summary, one AMR V-cycle of
# C3 / C4 / C5 by kernel, the 19-point kernels at 512^3, and the HBM traffic of the 19-point kernels from the PMC counters.
#   tools/final_measure.sh <tag>        (writes gpurun_out/<tag>_*)
set -e
tag=${1:-r03_final}
export TMPDIR=/tmp
o=gpurun_out
mkdir -p $o
python3 bench.py > $o/${tag}_bench.json 2> $o/${tag}_bench.err
rm -rf $o/prof_bench
rocprofv3 --kernel-trace --stats -d $o/prof_bench -o bench --output-format csv -- python3 bench.py --no-c4 --no-c5 --no-cpu-baseline --no-cartesian > $o/${tag}_bench_prof.log 2>&1
python3 tools/prof_summary.py stats $o/prof_bench $o/${tag}_c2_kernel_stats.md > /dev/null
rm -rf $o/prof_bench
for c in c3 c4 c5; do
  python3 tools/bench_amr.py --config $c --steps 5 > $o/${tag}_$c.json 2>/dev/null
done
tools/prof_cycle.sh c3 11 $o/${tag}_c3_cycle.txt 50 > /dev/null
tools/prof_cycle.sh c4 110 $o/${tag}_c4_cycle.txt 100 > /dev/null
tools/prof_cycle.sh c5 51 $o/${tag}_c5_cycle.txt 100 > /dev/null
python3 tools/bench_full19.py --n 512 --metric bathy > $o/${tag}_full19_bathy_512.json 2>/dev/null
python3 tools/bench_full19.py --n 384 --metric sheared > $o/${tag}_full19_sheared_384.json 2>/dev/null
# PMC: separate passes, counters only (no trace domains)
for cnt in FETCH_SIZE WRITE_SIZE; do
  rm -rf $o/pmc_$cnt
  rocprofv3 --pmc $cnt -d $o/pmc_$cnt -o pmc --output-format csv -- python3 tools/bench_full19.py --n 384 --metric bathy --reps 3 > $o/${tag}_pmc_$cnt.log 2>&1
done
python3 tools/prof_summary.py pmc $o/pmc_FETCH_SIZE $o/pmc_WRITE_SIZE $o/${tag}_full19_pmc_traffic_raw.json $((384*384*384)) "k_full_march<0,k_full_march<2,k_ghost_ops" > /dev/null
rm -rf $o/pmc_FETCH_SIZE $o/pmc_WRITE_SIZE
# the same two passes over the bench command: HBM bytes per launch of the headline kernels (profiles/traffic.json is written from this)
for cnt in FETCH_SIZE WRITE_SIZE; do
  rm -rf $o/pmcb_$cnt
  rocprofv3 --pmc $cnt -d $o/pmcb_$cnt -o pmc --output-format csv -- python3 bench.py --no-c4 --no-c5 --no-cpu-baseline --no-cartesian --steps 3 --warmup 1 > $o/${tag}_pmcb_$cnt.log 2>&1
done
python3 tools/prof_summary.py pmc $o/pmcb_FETCH_SIZE $o/pmcb_WRITE_SIZE $o/${tag}_c2_pmc_traffic_raw.json $((512*512*512)) > /dev/null
rm -rf $o/pmcb_FETCH_SIZE $o/pmcb_WRITE_SIZE
ls $o | grep $tag
