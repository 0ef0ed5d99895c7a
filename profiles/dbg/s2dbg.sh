# ablation of k_seed2 at the default bench (GPU box, repo root): bash profiles/dbg/s2dbg.sh "0 1 9" 
for m in ${1:-0}; do PCRAMP_S2DBG=$m python bench.py --steps 600 --warmup 30 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('dbg',$m, 'step ms', round(d['ms_per_step'],4), 'kernel ms', round(d['roofline']['kernel_ms'],4))"; done
