#!/bin/bash
# Regenerate the measurement records of a round (run on the GPU box via gpurun; results land in gpurun_out/profiles_new/ and
# are copied into profiles/ by hand afterwards).  Two parts, each within one gpurun call:
#   bash tools/refresh_profiles.sh r04 bench [precision]   the bench lines (default command first; `default` = only it, `rest` = the others)
#   bash tools/refresh_profiles.sh r04 prof  [precision]   rocprofv3 kernel stats of the default command, counter passes, step timeline
R=${1:-r04}; part=${2:-bench}; PREC=${3:-f16x3}
out=$GRAFT_REPO_ROOT/gpurun_out/profiles_new
mkdir -p $out
cd $GRAFT_REPO_ROOT
say() { echo "$(date +%T) $*"; }
if [ $part = default ]; then
  say "default bench (drugs200)"; t0=$SECONDS; python bench.py --gpus 1 --steps 20 --warmup 5 2>$out/${R}_bench_default.err | tail -1 > $out/${R}_${PREC}_bench.json; say "default bench took $((SECONDS - t0)) s wall"
elif [ $part = bench ] || [ $part = rest ]; then
  if [ $part = bench ]; then
  say "default bench (drugs200)"; t0=$SECONDS; python bench.py --gpus 1 --steps 20 --warmup 5 2>$out/${R}_bench_default.err | tail -1 > $out/${R}_${PREC}_bench.json; say "default bench took $((SECONDS - t0)) s wall"
  fi
  say "drugs 8x128 + breakdown"; python bench.py --workload drugs --breakdown $out/${R}_${PREC}_breakdown.json --no-cpu-baseline 2>/dev/null | tail -1 > $out/${R}_bench_drugs_8x128.json
  say "8x128 unfused front"; python bench.py --workload drugs --front unfused --steps 500 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 > $out/${R}_bench_drugs_8x128_unfused_front.json
  say "8x128 poly off"; python bench.py --workload drugs --radius-poly off --steps 300 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 > $out/${R}_bench_poly_off.json
  say "8x128 f32"; python bench.py --workload drugs --precision f32 --steps 300 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 > $out/${R}_bench_f32.json
  say "8x128 bf16x3"; python bench.py --workload drugs --precision bf16x3 --steps 500 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 > $out/${R}_bench_drugs_8x128_bf16x3.json
  say "default command in bf16x3"; python bench.py --precision bf16x3 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 > $out/${R}_bench_default_bf16x3.json
  say "sharpness sweep"; python tools/sharpness_sweep.py --out $out/${R}_sharpness_sweep.json > /dev/null 2>&1
  say "per-kernel stand-alone times (both branches on one stream), 36 x 128 batch"; bash tools/kstats.sh . serial --workload drugs --mols 36 --copies 128 --serial > $out/${R}_kernel_times_serial_36x128.txt 2>&1
  say "default schedule"; python bench.py --schedule default --no-cpu-baseline --no-extra 2>/dev/null | tail -1 > $out/${R}_bench_drugs200_default_sched.json
  say "qm9"; python bench.py --workload qm9 --mols 40 --copies 64 --steps 1000 --no-cpu-baseline 2>/dev/null | tail -1 > $out/${R}_bench_qm9.json
  say "large"; python bench.py --workload large --mols 2 --copies 128 --steps 300 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 > $out/${R}_bench_large.json
  say "alanine"; python bench.py --workload alanine --mols 1 --copies 250 --schedule default --job-steps 100 --steps 100 --warmup 0 --no-cpu-baseline 2>/dev/null | tail -1 > $out/${R}_bench_alanine.json
  say "small batches"; python bench.py --workload drugs --mols 1 --copies 100 --steps 2000 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 > $out/${R}_bench_small_batch.json
  say "force-dist"; python bench.py --workload drugs --force-dist --steps 300 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 > $out/${R}_force_dist.json
  python bench.py --workload drugs --force-dist --scaling strong --steps 300 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 > $out/${R}_force_dist_strong.json
  say "done"; ls -la $out
else
  say "kernel stats of the default command"
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kstats_refresh && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstats_refresh -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra > $out/${R}_kstats_bench.json 2>/dev/null)
  cp /tmp/kstats_refresh/*/*kernel_stats.csv $out/${R}_${PREC}_bench_kernel_stats.csv
  say "counter passes"; bash tools/pmc_bench.sh $R $PREC > $out/pmc_bench.log 2>&1
  say "step timeline (8 x 128 Drugs batch)"
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt && rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python3 $GRAFT_REPO_ROOT/bench.py --workload drugs --steps 30 --warmup 5 --no-cpu-baseline --no-extra --no-traj > /dev/null 2>&1)
  python3 tools/step_timeline.py /tmp/kt > $out/${R}_step_timeline.txt
  say "done"; head -12 $out/${R}_${PREC}_bench_kernel_stats.csv
fi
