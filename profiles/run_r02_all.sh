#!/bin/bash
# Everything the r02 numbers in DESIGN.md come from (GPU box, repo root): profiles of the default bench command, the bench lines,
# the 0.81 setting, the k_seed2 ablation.  Results under gpurun_out/; copy the summaries into profiles/.
set -e
bash profiles/run_r02_profiles.sh
cp gpurun_out/r02_hbm_traffic.json gpurun_out/r02_valu_pmc.json profiles/        # bench.py reads the static PMC figures from profiles/
python bench.py > gpurun_out/r02_bench_n1.json 2> gpurun_out/r02_bench_n1.err
python bench.py --steps 2000 --warmup 30 --target-threshold 0.9 --no-cpu-baseline --no-secondary > gpurun_out/r02_bench_thr081.json 2> gpurun_out/r02_bench_thr081.err
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_thr081 -- python3 bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-secondary --target-threshold 0.9 > gpurun_out/r02_thr081.log 2>&1
python3 profiles/summarize.py stats gpurun_out/r02_thr081 > gpurun_out/r02_kernel_stats_thr081.md
bash profiles/dbg/s2dbg.sh "0 128 40 32 8 1 2" > gpurun_out/r02_seed2_ablation.txt 2>&1
