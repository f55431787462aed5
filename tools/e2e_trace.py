#!/usr/bin/env python3
"""Timeline (kernels + memory copies, with their queues) of a few overlapped end-to-end iterations, from a rocprofv3 database:
    rocprofv3 --kernel-trace --memory-copy-trace -d OUT -- python3 tools/e2e_probe.py --trace
    python3 tools/e2e_trace.py OUT"""
import glob
import os
import re
import sqlite3
import subprocess
import sys

db = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True))[-1]
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tabs if "kernel_dispatch" in t][0]
mt = [t for t in tabs if "memory_copy" in t]
print("# tables:", kt, mt)
kcols = [r[1] for r in con.execute(f"pragma table_info({kt})")]
print("# kernel columns:", kcols)
q = "queue_id" if "queue_id" in kcols else "0"
st = "stream_id" if "stream_id" in kcols else "0"
sym = [t for t in tabs if "info_kernel_symbol" in t][0]
rows = [(s, e, "K", f"q{qq}/s{ss}", n) for n, s, e, qq, ss in con.execute(
    f"select s.kernel_name, d.start, d.end, d.{q}, d.{st} from {kt} d join {sym} s on d.kernel_id = s.id")]
if mt:
    mcols = [r[1] for r in con.execute(f"pragma table_info({mt[0]})")]
    print("# copy columns:", mcols)
    name = "name" if "name" in mcols else mcols[1]
    size = "size" if "size" in mcols else "0"
    rows += [(s, e, "C", "", f"{n} {sz / 1e6:.1f} MB") for n, s, e, sz in con.execute(f"select {name}, start, end, {size} from {mt[0]}")]
rows.sort()
names = sorted({r[4] for r in rows if r[2] == "K"})
dem = subprocess.run(["c++filt"], input="\n".join(n.replace(".kd", "") for n in names), capture_output=True, text=True).stdout.split("\n")
short = {n: re.sub(r"\(.*", "", d).replace("void ", "")[:48] for n, d in zip(names, dem)}
# the last 4 AdamW launches delimit three iterations
ends = [i for i, r in enumerate(rows) if r[2] == "K" and short[r[4]].startswith("k_adamw")]
lo, hi = ends[-4] + 1, ends[-1] + 1
t0 = rows[lo][0]
big = [r for r in rows[lo:hi] if (r[1] - r[0]) > 30000 or r[2] == "C" or "stack" in short.get(r[4], "") or "specreg" in short.get(r[4], "") or "adamw" in short.get(r[4], "") or "pack_mfma" in short.get(r[4], "")]
for s, e, kind, where, n in big:
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  {kind} {where:10s} {short.get(n, n)}")
