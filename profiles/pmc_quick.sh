#!/bin/bash
# quick counter pass of the default bench (GPU box, repo root): bash profiles/pmc_quick.sh TAG [extra bench args]
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-secondary $@"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats -- python3 $B > gpurun_out/${TAG}_stats.log 2>&1
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM -d gpurun_out/${TAG}_sq -- python3 $B > gpurun_out/${TAG}_sq.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT -d gpurun_out/${TAG}_sq2 -- python3 $B > gpurun_out/${TAG}_sq2.log 2>&1 || true
python3 profiles/summarize.py stats gpurun_out/${TAG}_stats > gpurun_out/${TAG}_stats.md; head -6 gpurun_out/${TAG}_stats.md
python3 profiles/summarize.py sq gpurun_out/${TAG}_sq.json gpurun_out/${TAG}_sq | grep -i "k_seed\|kernel |"
python3 profiles/summarize.py sq gpurun_out/${TAG}_sq2.json gpurun_out/${TAG}_sq2 | grep -i "k_seed\|kernel |" || true
