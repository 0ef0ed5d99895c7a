# A/B of k_post's waves per workgroup (GPU box): value, ms per step
for pw in 8 16 4 8 16; do PCRAMP_POST_WAVES=$pw python bench.py --no-cpu-baseline --no-secondary > gpurun_out/ab.json 2>gpurun_out/ab.err; python -c "
import json; d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]); print('post_waves $pw', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])" >> gpurun_out/ab_post_waves.txt; done; cat gpurun_out/ab_post_waves.txt
