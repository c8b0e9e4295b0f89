for rep in 1 2; do for l in "$@"; do AGDIFF_LIB=$PWD/$l python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-traj 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$l','conv_ms',round(d['roofline']['avg_launch_ms'],4),'step',round(d['ms_per_step'],3))"; done; done
