for m in 0 9; do PCRAMP_S2DBG=$m python bench.py --steps 400 --warmup 30 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('dbg',$m, 'step ms', round(d['ms_per_step'],4), 'kernel ms', round(d['roofline']['kernel_ms'],4))"; done
