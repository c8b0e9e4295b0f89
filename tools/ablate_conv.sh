#!/bin/bash
# Timing experiments on k_cfconv_fused (the filter-MLP CFConv: what runs when the polynomials are refused or off): phases of the
# kernel taken out one by one.  Needs the diagnostic build, the only one that reads AGDIFF_ABLATE from the environment:
#   make -C agdiff_amd/csrc clean all EXTRA=-DAG_CONV_ABLATE;  bash tools/ablate_conv.sh   (run with --radius-poly off to reach the kernel)
for a in 0 1 2 4 8 16 31 30 29; do AGDIFF_ABLATE=$a python bench.py --workload drugs --radius-poly off --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-traj 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ablate',$a,'conv_ms',round(d['roofline']['avg_launch_ms'],4))"; done
