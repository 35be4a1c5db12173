#!/bin/bash
# Everything profiles/ holds for a round, in one go (GPU box, from the repo root):
#   scripts/collect_round_profiles.sh r02     -> gpurun_out/profiles_r02/*  (copy what is to be judged into profiles/)
set -e
tag=${1:-rXX}
out=gpurun_out/profiles_$tag
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o b -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-e2e \
   > $out/${tag}_bench_under_rocprof.json 2> $out/bench_prof.err
cp $out/prof/*/b_kernel_stats.csv $out/${tag}_kernel_stats.csv 2>/dev/null || cp $out/prof/b_kernel_stats.csv $out/${tag}_kernel_stats.csv
echo "stats done"
python3 bench.py > $out/${tag}_bench.json 2> $out/bench.err
echo "bench done"
bash scripts/collect_traffic.sh > $out/traffic.log 2>&1 || true
cp profiles/traffic_latest.json $out/traffic_latest.json
echo "traffic done"
# the level-1 restriction: the default (DMA) form and the register-staged form it replaced
scripts/pmc_kernel.sh ${tag}_rs "restrict_stream_k<double, 64, 8, 5, 64, 4, false, true, true>" -- scripts/time_transfer.py 512 > /dev/null
cp gpurun_out/pmc_${tag}_rs/summary.txt $out/${tag}_restrict_counters.txt
NDSM_RS_VARIANT=8 scripts/pmc_kernel.sh ${tag}_rs8 "restrict_stream_k<double, 64, 8, 5, 64, 4, false, true, false>" -- scripts/time_transfer.py 512 > /dev/null
cp gpurun_out/pmc_${tag}_rs8/summary.txt $out/${tag}_restrict_staged_form_counters.txt
# the dominant launch (two-sweep Laplace pass of level 1) and the correction launch: instruction mix, waits, LDS
scripts/pmc_kernel.sh ${tag}_s2 "rbgs3_fused_k<double, 2, 136, 30, 1024, 4, true, 0, true, false>" -- scripts/run_sweeps.py 512 7 zero > /dev/null
cp gpurun_out/pmc_${tag}_s2/summary.txt $out/${tag}_smoother_s2_counters.txt
scripts/pmc_kernel.sh ${tag}_prol "rbgs3_fused_k<double, 1, 132, 31, 1024, 4, true, 3, false, false>" -- scripts/cycle_trace.py 512 > /dev/null
cp gpurun_out/pmc_${tag}_prol/summary.txt $out/${tag}_smoother_correction_counters.txt
echo "counters done"
python3 scripts/time_pipeline.py 512 > $out/${tag}_pipeline_phases.txt 2>&1 || true
python3 scripts/time_tail.py 512 > $out/${tag}_levels.txt 2>&1 || true
ls -la $out
