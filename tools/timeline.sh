#!/bin/bash
# per-launch timeline of the replayed training step only (quick: one rocprofv3 pass):  bash tools/timeline.sh TAG
tag=${1:-tl}
shift
args="$@"      # extra step_profile.py arguments, e.g. --dtype f32 or --ddp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/prof_tl -- python3 $R/tools/step_profile.py --steps 50 $args > $R/gpurun_out/${tag}_run.txt 2>&1
python3 $R/tools/step_profile.py --timeline /tmp/prof_tl --steps 50 > $R/gpurun_out/${tag}_step_timeline.txt
python3 $R/tools/step_profile.py --summarize /tmp/prof_tl --steps 50 > $R/gpurun_out/${tag}_step_breakdown.txt
tail -2 $R/gpurun_out/${tag}_run.txt; head -3 $R/gpurun_out/${tag}_step_breakdown.txt
