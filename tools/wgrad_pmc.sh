#!/bin/bash
# cache-side counters of the weight-gradient kernels (one conv_bench pass per counter group):  bash tools/wgrad_pmc.sh TAG
tag=${1:-wg}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS" "TA_BUSY_avr TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $grp -d /tmp/prof_wg$i -- python3 $R/tools/conv_bench.py --kinds wgrad --iters 3 > /tmp/prof_wg$i.out 2>&1
  echo "group $i ($grp): rc=$?" >> $R/gpurun_out/${tag}_wgrad_pmc.txt
  f=$(find /tmp/prof_wg$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" >> $R/gpurun_out/${tag}_wgrad_pmc.txt <<'PY'
import csv, sys, collections
f = sys.argv[1]
if not f:
    print("  (no counter file)"); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "wgrad" not in k: continue
    k = k[:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in acc:
    print("  " + k + "  " + "  ".join(f"{c}={v / max(1, n[(k, c)]):.4g}/launch" for c, v in acc[k].items()))
PY
done
tail -40 $R/gpurun_out/${tag}_wgrad_pmc.txt
