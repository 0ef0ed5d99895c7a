#!/bin/bash
# Everything the r02 numbers in DESIGN.md come from (GPU box, repo root): profiles of the default bench command and of the 0.81
# setting, the bench lines, the k_seed2 ablation, the other shapes.  Results under gpurun_out/; the summaries are copied into
# profiles/ by hand afterwards (the two PMC json files bench.py reads are copied here, before the bench runs).
set -e
bash profiles/run_r02_profiles.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
T="bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-secondary --target-threshold 0.9"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_thr081 -- python3 $T > gpurun_out/r02_thr081.log 2>&1
python3 profiles/summarize.py stats gpurun_out/r02_thr081 > gpurun_out/r02_kernel_stats_thr081.md
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM -d gpurun_out/r02_sq_thr081 -- python3 $T > gpurun_out/r02_sq_thr081.log 2>&1
python3 profiles/summarize.py sq gpurun_out/r02_valu_pmc_thr081.json gpurun_out/r02_sq_thr081 > gpurun_out/r02_valu_pmc_thr081.md
cp gpurun_out/r02_hbm_traffic.json gpurun_out/r02_valu_pmc.json gpurun_out/r02_valu_pmc_thr081.json profiles/   # bench.py reads the static PMC figures from profiles/
python bench.py > gpurun_out/r02_bench_n1.json 2> gpurun_out/r02_bench_n1.err
python bench.py --steps 2000 --warmup 30 --target-threshold 0.9 --no-cpu-baseline --no-secondary > gpurun_out/r02_bench_thr081.json 2> gpurun_out/r02_bench_thr081.err
bash profiles/dbg/s2dbg.sh "0 128 40 32 8 1 2" > gpurun_out/r02_seed2_ablation.txt 2>&1
# other shapes, same code
for c in C3 C5_shard; do python bench.py --config $c --steps 2000 --warmup 30 --no-cpu-baseline --no-secondary > gpurun_out/r02_bench_$c.json 2> gpurun_out/r02_bench_$c.err; done
python bench.py --optimize-shifts --steps 1000 --warmup 30 --no-cpu-baseline --no-secondary > gpurun_out/r02_bench_shifts.json 2> gpurun_out/r02_bench_shifts.err
