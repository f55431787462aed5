"""Build libbrainxai.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbrainxai.so")
SOURCES = ["core.hip", "conv3x3.hip", "conv3x3_mfma.hip", "conv3x3_split.hip", "tail.hip", "heads.hip", "eeg.hip", "eeg_generic.hip", "eeg_mfma.hip", "eeg_collapse.hip", "eeg_deep.hip", "attrib.hip", "montage.hip", "specprep.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=fast", "-Wall", "-Wno-unused-function"]
# per-file additions (later flags win): the montage IIR must round every product and sum separately, as scipy's C loop does
EXTRA_FLAGS = {"montage.hip": ["-ffp-contract=off"], "specprep.hip": ["-ffp-contract=off"]}


def _digest():
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)) + ["../../include/brainxai.h"]:
        p = os.path.join(CSRC, name)
        if os.path.isfile(p) and (p.endswith(".hip") or p.endswith(".h")):
            h.update(name.encode())
            h.update(open(p, "rb").read())
    h.update((" ".join(FLAGS) + repr(sorted(EXTRA_FLAGS.items()))).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    stamp = LIB + ".stamp"
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for src in SOURCES:
        obj = os.path.join(HERE, "build", src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [HIPCC, *FLAGS, *EXTRA_FLAGS.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"--- hipcc failed on {src} ---\n{out}\n")
        elif verbose and out.strip():
            sys.stderr.write(f"--- hipcc warnings for {src} ---\n{out}\n")
    if failed:
        raise RuntimeError("libbrainxai build failed")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB])
    open(stamp, "w").write(dig)
    if verbose:
        print(f"built {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
