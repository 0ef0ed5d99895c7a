#!/bin/bash
# Everything the r03 numbers in DESIGN.md come from (GPU box, repo root): profiles of the default bench command (trace + four
# counter passes), of the secondary kernels and of pcr_optimize_batch, then -- with the fresh counter summaries copied into profiles/,
# where bench.py reads its static figures -- the bench lines: default, driver-like (--steps 20), the 0.81 setting, the other shapes.
# Results under gpurun_out/; the summaries are copied into profiles/ by hand afterwards.   usage: bash profiles/run_r03_all.sh [tag]
TAG=${1:-r03}
bash profiles/run_r03_profiles.sh $TAG || exit 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp gpurun_out/${TAG}_hbm_traffic.json gpurun_out/${TAG}_valu_pmc.json gpurun_out/${TAG}_wave_cycles_pmc.json gpurun_out/${TAG}_secondary_pmc.json profiles/
echo "profiles done"
python bench.py > gpurun_out/${TAG}_bench_n1.json 2> gpurun_out/${TAG}_bench_n1.err || exit 1
echo "bench n1 done"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/${TAG}_bench_steps20.json 2>/dev/null
python bench.py --steps 1500 --warmup 60 --target-threshold 0.9 --no-cpu-baseline --no-secondary > gpurun_out/${TAG}_bench_thr081.json 2>/dev/null
python bench.py --config C3 --steps 3000 --warmup 60 --no-cpu-baseline --no-secondary > gpurun_out/${TAG}_bench_C3.json 2>/dev/null
python bench.py --config C5_shard --steps 2000 --warmup 60 --no-cpu-baseline --no-secondary > gpurun_out/${TAG}_bench_C5_shard.json 2>/dev/null
echo "shapes done"
python bench.py --config C5 --steps 300 --warmup 60 --no-cpu-baseline --no-secondary > gpurun_out/${TAG}_bench_C5_strong_n1.json 2>/dev/null
python bench.py --optimize-shifts --steps 1000 --warmup 60 --no-cpu-baseline --no-secondary > gpurun_out/${TAG}_bench_shifts.json 2>/dev/null
bash profiles/dbg/s2dbg.sh "0 128 40 32 8 1 2" > gpurun_out/${TAG}_seed2_ablation_final.txt 2>&1
echo "all done"
