#!/bin/bash
# Collect the per-round profile evidence on the GPU box (run through gpurun from the repo root):
#   BX_COMMIT=$(git rev-parse --short HEAD) ... bash tools/collect_profiles.sh r03k [bf16|f32]
# writes gpurun_out/<tag>_*: bench JSON (plain and under rocprof), rocprofv3 kernel stats of the bench command, the per-step
# kernel breakdown + launch timeline (hipGraph-replayed steps only), the PMC HBM-traffic summary (FETCH_SIZE / WRITE_SIZE in
# separate passes, provenance header) and SQ counters.  With f32: the step-level evidence of the fp32-storage path only (suffix _f32).
tag=${1:-rXX}
dt=${2:-bf16}
R=$GRAFT_REPO_ROOT
export BX_PROFILE_DTYPE=$dt
sfx=""; [ "$dt" = "f32" ] && sfx="_f32"
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
if [ "$dt" = "bf16" ]; then
  python3 $R/bench.py > $R/gpurun_out/${tag}_bench_default.json 2> $R/gpurun_out/${tag}_bench_default.err
  echo "bench done rc=$?" >> $R/gpurun_out/${tag}_log.txt
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 $R/bench.py --steps 50 --cpu-steps 2 > $R/gpurun_out/${tag}_bench_under_rocprof.json 2> /tmp/prof_bench.err
  f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $R/gpurun_out/${tag}_kernel_stats.csv
  echo "stats done: $f" >> $R/gpurun_out/${tag}_log.txt
fi
rocprofv3 --kernel-trace -d /tmp/prof_steps$sfx -- python3 $R/tools/step_profile.py --steps 50 --dtype $dt > /tmp/prof_steps.out 2>&1
python3 $R/tools/step_profile.py --summarize /tmp/prof_steps$sfx --steps 50 > $R/gpurun_out/${tag}_step_breakdown$sfx.txt 2>> $R/gpurun_out/${tag}_log.txt
python3 $R/tools/step_profile.py --timeline /tmp/prof_steps$sfx --steps 50 > $R/gpurun_out/${tag}_step_timeline$sfx.txt 2>> $R/gpurun_out/${tag}_log.txt
echo "breakdown done ($dt)" >> $R/gpurun_out/${tag}_log.txt
BX_GRAPH_LOOPS=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_fetch$sfx -- python3 $R/tools/step_profile.py --steps 4 --dtype $dt > /tmp/prof_fetch.out 2>&1
BX_GRAPH_LOOPS=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof_write$sfx -- python3 $R/tools/step_profile.py --steps 4 --dtype $dt > /tmp/prof_write.out 2>&1
python3 $R/tools/step_profile.py --pmc /tmp/prof_fetch$sfx /tmp/prof_write$sfx --steps 4 > $R/gpurun_out/${tag}_pmc_hbm_traffic$sfx.txt 2>> $R/gpurun_out/${tag}_log.txt
echo "pmc done ($dt)" >> $R/gpurun_out/${tag}_log.txt
BX_GRAPH_LOOPS=0 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES -d /tmp/prof_sq$sfx -- python3 $R/tools/step_profile.py --steps 4 --dtype $dt > /tmp/prof_sq.out 2>&1
python3 $R/tools/step_profile.py --sq /tmp/prof_sq$sfx --steps 4 > $R/gpurun_out/${tag}_sq_counters$sfx.txt 2>> $R/gpurun_out/${tag}_log.txt
echo "sq done ($dt)" >> $R/gpurun_out/${tag}_log.txt
cat $R/gpurun_out/${tag}_log.txt
head -12 $R/gpurun_out/${tag}_step_breakdown$sfx.txt
