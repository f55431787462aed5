#!/bin/bash
# A/B of the in-launch finalize policy and the pooled conv3 epilogue (run through gpurun from the repo root):
#   bash tools/ab_tree.sh <tag>     -> gpurun_out/<tag>_ab.txt, gpurun_out/<tag>_bd_<rows>.txt
tag=${1:-ab}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for fuse in 1 0; do for rows in 0 128 1000000; do
  echo "== BX_FUSE_POOL=$fuse BX_TREE_MAX_ROWS=$rows" >> $R/gpurun_out/${tag}_ab.txt
  BX_FUSE_POOL=$fuse BX_TREE_MAX_ROWS=$rows python3 $R/bench.py --no-extras --steps 100 --warmup 20 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" >> $R/gpurun_out/${tag}_ab.txt
done; done
for rows in 0 128; do
  export BX_TREE_MAX_ROWS=$rows
  rocprofv3 --kernel-trace -d /tmp/prof_steps_$rows -- python3 $R/tools/step_profile.py --steps 50 > /tmp/prof_steps.out 2>&1
  python3 $R/tools/step_profile.py --summarize /tmp/prof_steps_$rows --steps 50 > $R/gpurun_out/${tag}_bd_$rows.txt 2>&1
done
cat $R/gpurun_out/${tag}_ab.txt
