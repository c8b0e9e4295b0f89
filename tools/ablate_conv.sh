# needs the diagnostic build: make -C agdiff_amd/csrc clean all EXTRA=-DAG_CONV_ABLATE
for a in 0 1 2 4 8 16 31 30 29; do AGDIFF_ABLATE=$a python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traj 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ablate',$a,'conv_ms',round(d['roofline']['avg_launch_ms'],4))"; done
