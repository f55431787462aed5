#!/usr/bin/env python3
"""ONE diagnostic run of the state in which dist.destroy_process_group() aborted once in round 2 (gpurun_out/r02p_gputests.txt:
'Fatal Python error: Aborted' at distributed_c10d.py:2264 = backend.shutdown(), no C++ message captured): a 1-rank RCCL group
with, ALIVE at the bare destroy call, a DataParallel wrapper, a FlatAdamW, a graph-pieces GraphedTrainStep (two hipGraphs sharing a
pool, captured while the group was alive) and the two c10d Work handles of the last step's asynchronous all-reduces -- i.e. what
the reference's calling pattern (a bare cleanup() = destroy_process_group, XAI_Multimodality.py:70-71) leaves behind, WITHOUT
brainxai.cleanup()'s releases.  The child process runs with c10d / RCCL / HIP logging on, so that an abort -- if it recurs -- comes
with its message.  Run once; not a loop.   python tools/pg_teardown_probe.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "--child" not in sys.argv:
    env = dict(os.environ, TORCH_CPP_LOG_LEVEL="INFO", TORCH_SHOW_CPP_STACKTRACES="1", NCCL_DEBUG="WARN", TORCH_NCCL_DEBUG_INFO_TEMP_FILE="/tmp/nccl_trace_",
               HSA_ENABLE_IPC_MODE_LEGACY="0", BX_DDP_GRAPH="pieces", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    tail = "\n".join(p.stdout.splitlines()[-60:])
    print(tail)
    print(f"[probe] child exit code {p.returncode} ({'ABORTED' if p.returncode < 0 else 'clean teardown'})")
    sys.exit(0)

sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import brainxai  # noqa: E402
from brainxai import train as T  # noqa: E402

dev = torch.device("cuda:0")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
torch.manual_seed(3)
net = brainxai.build_multimodal(19, 2000, 4, dropout=0.5, compute_dtype=torch.bfloat16).to(dev).train()
ddp = brainxai.DataParallel(net)
opt = brainxai.FlatAdamW(net.parameters(), lr=1e-3)
step = brainxai.GraphedTrainStep(net, opt, brainxai.KLDivLoss(), ddp=ddp, strict=True)
g = torch.Generator().manual_seed(1)
eeg, spec = torch.randn(4, 1, 19, 2000, generator=g).to(dev), torch.rand(4, 4, 32, 64, generator=g).to(dev)
lab = torch.softmax(torch.randn(4, 6, generator=g), 1).to(dev)
for _ in range(5):
    step([eeg, spec], lab)
# keep raw Work handles alive the way round 2's _Reduction did (no wait(), no drop)
works = [dist.all_reduce(opt.flat_g.narrow(0, 0, 1024), op=dist.ReduceOp.AVG, async_op=True) for _ in range(2)]
print(f"[probe] alive at teardown: {len(step._graphs)} graphed step(s) of kind {[e[0] for e in step._graphs.values()]}, wrapper, optimizer, {len(works)} un-waited Work handles", flush=True)
torch.cuda.synchronize()
dist.destroy_process_group()                       # the bare call, as in round 2's test and the reference
print("[probe] destroy_process_group() returned", flush=True)
del works, step, ddp, opt, net
