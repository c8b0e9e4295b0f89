# A/B two AGDIFF_ABLATE settings on the same box, interleaved: bash tools/ab_conv.sh <a> <b>
for rep in 1 2 3; do for a in $1 $2; do AGDIFF_ABLATE=$a python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-traj 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ablate',$a,'conv_ms',round(d['roofline']['avg_launch_ms'],4),'step',round(d['ms_per_step'],3))"; done; done
