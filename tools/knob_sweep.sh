# per-shape sweeps of the conv-family tuning knobs (tools/conv_bench.py per setting); run on the GPU box:  bash tools/knob_sweep.sh
cd $GRAFT_REPO_ROOT
for w in 256 384 512 768 1024; do echo "== BX_WGRAD_WANT=$w"; BX_WGRAD_WANT=$w python tools/conv_bench.py --kinds wgrad 2>/dev/null | grep -E "wgrad|totals"; done
for c in 0 1 2; do echo "== BX_CONV_C=$c"; BX_CONV_C=$c python tools/conv_bench.py --kinds fwd,dgrad 2>/dev/null | grep -E "fwd|dgrad|totals" | grep -v "128x256\|64x128"; done
echo "== BX_CONV_NC4_MIN=100000 (NC=2 everywhere)"; BX_CONV_C=0 BX_CONV_NC4_MIN=100000 python tools/conv_bench.py --kinds fwd,dgrad 2>/dev/null | grep -E "fwd|dgrad|totals" | grep -v "128x256\|64x128"
echo "== BX_CONV_NC4_MIN=1 (NC=4 everywhere)"; BX_CONV_C=0 BX_CONV_NC4_MIN=1 python tools/conv_bench.py --kinds fwd,dgrad 2>/dev/null | grep -E "fwd|dgrad|totals" | grep -v "128x256\|64x128"
