#!/bin/bash
# Regenerate the measurement records of this round (run on the GPU box via gpurun; results land in
# gpurun_out/profiles_new/ and are copied into profiles/ by hand afterwards).   bash tools/refresh_profiles.sh r02
set -x
R=${1:-r02}
out=$GRAFT_REPO_ROOT/gpurun_out/profiles_new
mkdir -p $out
cd $GRAFT_REPO_ROOT
python bench.py --breakdown $out/${R}_bf16x3_breakdown.json 2>$out/bench.err | tail -1 > $out/${R}_bf16x3_bench.json
python bench.py --radius-poly off --steps 300 --no-cpu-baseline 2>/dev/null | tail -1 > $out/${R}_bench_poly_off.json
python bench.py --precision f32 --steps 300 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 > $out/${R}_bench_f32.json
python bench.py --workload drugs200 --no-cpu-baseline 2>/dev/null | tail -1 > $out/${R}_bench_drugs200.json
python bench.py --workload drugs200 --schedule default --no-cpu-baseline 2>/dev/null | tail -1 > $out/${R}_bench_drugs200_default_sched.json
python bench.py --workload qm9 --mols 40 --copies 64 --steps 1000 --no-cpu-baseline 2>/dev/null | tail -1 > $out/${R}_bench_qm9.json
python bench.py --workload large --mols 2 --copies 128 --steps 300 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 > $out/${R}_bench_large.json
python bench.py --workload alanine --mols 1 --copies 250 --schedule default --job-steps 100 --steps 100 --warmup 0 --no-cpu-baseline 2>/dev/null | tail -1 > $out/${R}_bench_alanine.json
python bench.py --mols 1 --copies 100 --steps 2000 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 > $out/${R}_bench_small_batch.json
python bench.py --mols 1 --copies 25 --steps 2000 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 > $out/${R}_bench_small_batch_25.json
python bench.py --force-dist --steps 300 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 > $out/${R}_force_dist.json
python bench.py --force-dist --scaling strong --steps 300 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 > $out/${R}_force_dist_strong.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstats_refresh -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extra > $out/kstats_bench.json 2>/dev/null
cp /tmp/kstats_refresh/*/*kernel_stats.csv $out/${R}_bf16x3_bench_kernel_stats.csv
cd $GRAFT_REPO_ROOT && bash tools/pmc_traffic.sh gpurun_out/pmc_traffic > $out/${R}_bf16x3_pmc_traffic.txt 2>&1
cp gpurun_out/pmc_traffic/traffic.json $out/${R}_bf16x3_pmc_traffic_raw.json
bash tools/pmc_sq.sh > $out/${R}_bf16x3_pmc_sq.txt 2>&1
tail -n 12 $out/${R}_bf16x3_pmc_traffic.txt; tail -n 30 $out/${R}_bf16x3_pmc_sq.txt
