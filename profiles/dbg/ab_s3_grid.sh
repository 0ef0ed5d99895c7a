# A/B of k_seed3's workgroup size and workgroups per CU (GPU box): value, ms per step, kernel ms by events
for cfg in "1024 2" "512 4" "1024 2"; do set -- $cfg; PCRAMP_S3_WG=$1 PCRAMP_S3_PER_CU=$2 python bench.py --no-cpu-baseline --no-secondary > gpurun_out/ab.json 2>gpurun_out/ab.err; python -c "
import json; d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]); print('$cfg', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])" >> gpurun_out/ab_s3_grid.txt; done; cat gpurun_out/ab_s3_grid.txt
