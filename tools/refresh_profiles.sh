#!/bin/bash
# Regenerate the committed measurement records under profiles/ (run on the GPU box via gpurun; results are written
# to gpurun_out/profiles_new/ and copied into profiles/ by hand afterwards).
set -x
out=$GRAFT_REPO_ROOT/gpurun_out/profiles_new
mkdir -p $out
cd $GRAFT_REPO_ROOT
python bench.py --breakdown $out/r01_bf16x3_breakdown.json 2>$out/bench.err | tail -1 > $out/r01_bf16x3_bench.json
python bench.py --precision f32 --steps 300 --no-cpu-baseline 2>/dev/null | tail -1 > $out/r01_bench_f32.json
python bench.py --schedule default --steps 1000 --no-cpu-baseline 2>/dev/null | tail -1 > $out/r01_bench_default_sched.json
python bench.py --schedule default --steps 1000 --no-cpu-baseline --no-skip 2>/dev/null | tail -1 > $out/r01_bench_default_sched_noskip.json
python bench.py --workload qm9 --steps 1000 --no-cpu-baseline 2>/dev/null | tail -1 > $out/r01_bench_qm9.json
python bench.py --workload large --mols 2 --copies 128 --steps 300 --no-cpu-baseline 2>/dev/null | tail -1 > $out/r01_bench_large.json
python bench.py --workload alanine --mols 1 --copies 250 --schedule default --job-steps 100 --steps 100 --warmup 0 --no-cpu-baseline 2>/dev/null | tail -1 > $out/r01_bench_alanine.json
python bench.py --mols 1 --copies 100 --steps 500 --no-cpu-baseline 2>/dev/null | tail -1 > $out/r01_bench_small_batch.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kstats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $out/kstats_bench.json 2>/dev/null
cp $out/kstats/*/*kernel_stats.csv $out/r01_bf16x3_bench_kernel_stats.csv
cd $GRAFT_REPO_ROOT && bash tools/pmc_traffic.sh gpurun_out/profiles_new/pmc > $out/pmc_traffic.txt 2>&1
cat $out/pmc_traffic.txt
