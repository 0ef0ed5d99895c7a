for cfg in "16 torch" "32 torch" "16 abi" "32 abi" "64 abi"; do set -- $cfg
PCRAMP_BENCH_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --gather-every $1 --exchange $2 --no-cpu-baseline --no-secondary > gpurun_out/fd.json 2>gpurun_out/fd.err || { tail -3 gpurun_out/fd.err; continue; }
python -c "
import json; d=json.loads(open('gpurun_out/fd.json').read().strip().splitlines()[-1]); print('steps20 $cfg', d['value'], d['ms_per_step'])"
PCRAMP_BENCH_FORCE_DIST=1 python bench.py --gather-every $1 --exchange $2 --no-cpu-baseline --no-secondary > gpurun_out/fd.json 2>gpurun_out/fd.err || { tail -3 gpurun_out/fd.err; continue; }
python -c "
import json; d=json.loads(open('gpurun_out/fd.json').read().strip().splitlines()[-1]); print('steps8000 $cfg', d['value'], d['ms_per_step'])"
done
