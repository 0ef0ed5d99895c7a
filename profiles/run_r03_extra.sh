#!/bin/bash
# The r03 bench lines run_r03_all.sh does not make (GPU box, repo root): a C4 shard, the second form forced (A/B), and the soak of the
# default pass against the bit-sliced scan.   usage: bash profiles/run_r03_extra.sh [tag]
TAG=${1:-r03}
python bench.py --config C4_shard --steps 300 --warmup 20 --no-cpu-baseline --no-secondary > gpurun_out/${TAG}_bench_C4_shard.json 2>/dev/null || exit 1
echo "C4 shard done"
PCRAMP_SEED3=0 python bench.py --no-cpu-baseline --no-secondary > gpurun_out/${TAG}_bench_n1_second_form.json 2>/dev/null || exit 1
echo "second form done"
python profiles/dbg/soak_lean.py 40000 > gpurun_out/${TAG}_soak_third_form.txt 2>&1 || exit 1
cat gpurun_out/${TAG}_soak_third_form.txt
