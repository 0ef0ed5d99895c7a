#!/bin/bash
# Profiles of the default bench command (run on the GPU box from the repo root; results land under gpurun_out/).
# rocprofv3 gets the interpreter itself after `--`; PMC passes run separately from the trace pass and from each other
# (FETCH_SIZE and WRITE_SIZE do not fit one pass; MI355X_MICROARCH.md, rocprofv3 PMC slots).  Usage:
#   bash profiles/run_r03_profiles.sh [tag] [passes]     passes: any of "stats fetch write sq sqb sec secsq opt optsq" (default: all)
TAG=${1:-r03}
PASSES=${2:-"stats fetch write sq sqb sec secsq opt optsq"}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-secondary"
for p in $PASSES; do
case $p in
stats) rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats -- python3 $B > gpurun_out/${TAG}_stats.log 2>&1
       python3 profiles/summarize.py stats gpurun_out/${TAG}_stats > gpurun_out/${TAG}_kernel_stats.md ;;
fetch) rocprofv3 --output-format csv --pmc FETCH_SIZE -d gpurun_out/${TAG}_fetch -- python3 $B > gpurun_out/${TAG}_fetch.log 2>&1 ;;
write) rocprofv3 --output-format csv --pmc WRITE_SIZE -d gpurun_out/${TAG}_write -- python3 $B > gpurun_out/${TAG}_write.log 2>&1
       python3 profiles/summarize.py pmc gpurun_out/${TAG}_hbm_traffic.json gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write > gpurun_out/${TAG}_hbm_traffic.md ;;
sq)    rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM -d gpurun_out/${TAG}_sq -- python3 $B > gpurun_out/${TAG}_sq.log 2>&1
       python3 profiles/summarize.py sq gpurun_out/${TAG}_valu_pmc.json gpurun_out/${TAG}_sq > gpurun_out/${TAG}_valu_pmc.md ;;
sqb)   # where the wave cycles go: issuing (ACTIVE_INST_*), stalled at issue (WAIT_INST_*), parked on s_waitcnt / barriers (WAIT_ANY)
       rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM -d gpurun_out/${TAG}_sqb -- python3 $B > gpurun_out/${TAG}_sqb.log 2>&1
       python3 profiles/summarize.py sq gpurun_out/${TAG}_wave_cycles_pmc.json gpurun_out/${TAG}_sqb > gpurun_out/${TAG}_wave_cycles_pmc.md ;;
sec)   # the secondary kernels (SW, thermodynamics, C3's background path, pcr_optimize_batch)
       rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_sec -- python3 profiles/dbg/secondary_prof.py > gpurun_out/${TAG}_sec.log 2>&1
       python3 profiles/summarize.py stats gpurun_out/${TAG}_sec > gpurun_out/${TAG}_kernel_stats_secondary.md ;;
secsq) rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAVES SQ_WAIT_INST_ANY -d gpurun_out/${TAG}_secsq -- python3 profiles/dbg/secondary_prof.py sw thermo c3bg > gpurun_out/${TAG}_secsq.log 2>&1
       python3 profiles/summarize.py sq gpurun_out/${TAG}_secondary_pmc.json gpurun_out/${TAG}_secsq > gpurun_out/${TAG}_secondary_pmc.md ;;
opt)   # pcr_optimize_batch alone: 256 sampler assays, C2 targets + 2 000 backgrounds, two calls (the first sizes the buffers)
       rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_opt -- python3 profiles/dbg/opt_prof.py batch 256 > gpurun_out/${TAG}_opt.log 2>&1
       python3 profiles/summarize.py stats gpurun_out/${TAG}_opt > gpurun_out/${TAG}_optimize_batch_stats.md ;;
optsq) rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAVES -d gpurun_out/${TAG}_optsq -- python3 profiles/dbg/opt_prof.py batch 256 > gpurun_out/${TAG}_optsq.log 2>&1
       python3 profiles/summarize.py sq gpurun_out/${TAG}_optimize_batch_pmc.json gpurun_out/${TAG}_optsq > gpurun_out/${TAG}_optimize_batch_pmc.md ;;
esac
done
exit 0
