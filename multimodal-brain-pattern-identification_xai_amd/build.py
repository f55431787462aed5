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
# No packed-fp32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32), round 3.  Found with the EEG branch running beside the
# spectrogram branch (ops.OVERLAP_EEG): k_eeg_sep, whose inner loop hipcc had vectorised into v_pk_fma_f32, returned wrong LOW halves of
# a few result pairs whenever kernels of the other branch shared its CUs (128 of 200 graph replays of the forward differed from the
# serial run, 0 of 200 with scalar v_fmac_f32; tools/overlap_fwd_check.py) -- alone on the chip the same code never failed.  Every
# parity test stands (hipcc contracts a few multiply-adds differently without the packed forms, so results are not bit-identical to
# the packed build's, only equally accurate), and the step time did not move.
# (The host pass of hipcc does not know the feature and says so once per file; build() drops that line.)
if os.environ.get("BX_PACKED_FP32", "0") != "1":
    FLAGS += ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
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
        else:
            out = "\n".join(l for l in out.splitlines() if "is not a recognized feature for this target" not in l)
            if verbose and out.strip():
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
