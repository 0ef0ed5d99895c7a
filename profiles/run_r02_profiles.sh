#!/bin/bash
# Profiles of the default bench command (run on the GPU box from the repo root; results land under gpurun_out/).
# rocprofv3 gets the interpreter itself after `--`; PMC passes run separately from the trace pass and from each other
# (FETCH_SIZE and WRITE_SIZE do not fit one pass; MI355X_MICROARCH.md, rocprofv3 PMC slots).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_stats -- python3 $B > gpurun_out/r02_stats.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d gpurun_out/r02_fetch -- python3 $B > gpurun_out/r02_fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d gpurun_out/r02_write -- python3 $B > gpurun_out/r02_write.log 2>&1
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM -d gpurun_out/r02_sq -- python3 $B > gpurun_out/r02_sq.log 2>&1
python3 profiles/summarize.py stats gpurun_out/r02_stats > gpurun_out/r02_kernel_stats.md
python3 profiles/summarize.py pmc gpurun_out/r02_hbm_traffic.json gpurun_out/r02_fetch gpurun_out/r02_write > gpurun_out/r02_hbm_traffic.md
python3 profiles/summarize.py sq gpurun_out/r02_valu_pmc.json gpurun_out/r02_sq > gpurun_out/r02_valu_pmc.md
