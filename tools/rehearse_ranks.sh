#!/bin/bash
# bench.py's N > 1 path with 2 and 4 ranks on ONE GPU over the shared-memory transport (SOMAR_BENCH_COMM=shm; RCCL refuses
# several ranks on one device), next to the one-rank run of the same reduced problem: plans, packing, agglomerated tail and
# box-ordered sums with more than two ranks.  Not a performance run.   usage: tools/rehearse_ranks.sh [outdir]
OUT=${1:-gpurun_out}
A="--steps 3 --warmup 1 --size 256 --no-cpu-baseline --c4-scale 2 --c4-steps 2"
timeout -k 10 300 python bench.py --gpus 1 $A > $OUT/n1.json 2> $OUT/n1.err
for N in 2 4; do
    SOMAR_BENCH_COMM=shm timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N \
        --master-addr 127.0.0.1 --master-port 2954$N bench.py --gpus $N $A > $OUT/n$N.json 2> $OUT/n$N.err
    echo "rc$N=$?"
done
python - <<PY
import json
for n in (1, 2, 4):
    try:
        L = [l for l in open("$OUT/n%d.json" % n) if l.startswith("{")]
        d = json.loads(L[-1])
        c5 = d.get("c5_amr", {})
        print(n, "C2", round(d["value"], 2), d["vcycle_contraction"], "twin", d.get("vcycle_contraction_single_process_same_layout"), "depth", d["config"]["mg_depth"], "| C4 ms",
              round(d["c4_amr"]["ms_per_amr_vcycle"], 2), d["c4_amr"]["amr_vcycle_contraction"], "| C5",
              c5.get("amr_vcycle_contraction", c5))
    except Exception as e:
        print(n, "ERR", repr(e))
PY
for f in $OUT/n4.err $OUT/n2.err; do tail -n 4 $f | cut -c1-300; done
