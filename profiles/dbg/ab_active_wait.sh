# A/B of the runtime's active-wait window (ROC_ACTIVE_WAIT_TIMEOUT, microseconds of spinning before a host wait falls back to the interrupt)
for w in 0 100 0 100; do
ROC_ACTIVE_WAIT_TIMEOUT=$w python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/aw.json 2>gpurun_out/aw.err
python -c "
import json; d=json.loads(open('gpurun_out/aw.json').read().strip().splitlines()[-1]); s=d['secondary']
print('wait $w: steps20', d['value'], d['ms_per_step'], '| optimize_batch', s['optimize_batch']['ms_total'], '| design', s['design_loop']['ms_per_iteration'], '| c5 batch', s['optimize_batch_c5_shard']['ms_total'], '| thermo abi', s['thermodynamics']['abi_ms_per_call'], '| c3 bg', s['c3_background']['select_words_ms'], s['c3_background']['find_background_match_ms'], '| sw abi', s['smith_waterman']['abi_ms_per_call'], '| optimize1', s['optimize']['ms_per_assay'])"
done
