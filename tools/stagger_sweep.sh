for s in 0 40 80 120 160 240 400; do AGDIFF_STAGGER=$s python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traj 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stagger',$s,'conv_ms',round(d['roofline']['avg_launch_ms'],4), 'ms/step', round(d['ms_per_step'],3))"; done
