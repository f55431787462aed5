#!/bin/bash
# per-launch timeline of one replayed Grad-CAM sweep batch (configs[3]):  bash tools/timeline_gradcam.sh TAG
tag=${1:-tl}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/prof_gc -- python3 $R/tools/step_profile.py --gradcam --steps 50 > $R/gpurun_out/${tag}_gradcam_run.txt 2>&1
python3 $R/tools/step_profile.py --timeline /tmp/prof_gc --steps 50 --delim k_gradcam_head > $R/gpurun_out/${tag}_gradcam_timeline.txt
grep batches $R/gpurun_out/${tag}_gradcam_run.txt; head -3 $R/gpurun_out/${tag}_gradcam_timeline.txt
