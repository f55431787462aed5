"""ctypes binding of libbrainxai.so (the C ABI declared in include/brainxai.h).

The product path has NO fallback: if the shared object is missing or a launcher returns an
error, a RuntimeError is raised.  The library is built in-tree by ``build.py`` (hipcc, gfx950).
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libbrainxai.so")

BX_F32, BX_BF16 = 0, 1
BX_POOL_MAX, BX_POOL_AVG = 0, 1
BX_ALGO_AUTO, BX_ALGO_DIRECT, BX_ALGO_MFMA = 0, 1, 2
BX_EPI_RELU = 1
BX_EPI_MASK_BITS = 2
BX_TAIL_SYNC_WORDS = 8192

vp, i32, i64, u32, f32, sz = C.c_void_p, C.c_int, C.c_int64, C.c_uint32, C.c_float, C.c_size_t


class TailDesc(C.Structure):
    _fields_ = [("B", i32), ("H", i32), ("W", i32), ("Cin_p", i32), ("C", i32), ("pool", i32), ("training", i32),
                ("eps", f32), ("momentum", f32), ("dropout_p", f32), ("salt", u32), ("dtype", i32), ("sync", vp), ("route", vp)]


class EegDesc(C.Structure):
    _fields_ = [("B", i32), ("Chans", i32), ("T", i32), ("F1", i32), ("D", i32), ("F2", i32), ("K1", i32), ("K2", i32),
                ("P1", i32), ("P2", i32), ("training", i32), ("eps", f32), ("momentum", f32), ("dropout_p", f32),
                ("salt", u32), ("dtype", i32), ("collapse", i32), ("dropout_p2", f32), ("x_slot", vp)]


class EegParams(C.Structure):
    _fields_ = [(n, vp) for n in ("conv1_w", "bn1_w", "bn1_b", "bn1_rm", "bn1_rv", "bn1_nbt", "dw_w", "bn2_w", "bn2_b",
                                  "bn2_rm", "bn2_rv", "bn2_nbt", "sep_w", "bn3_w", "bn3_b", "bn3_rm", "bn3_rv", "bn3_nbt")]


class EegDeepDesc(C.Structure):
    _fields_ = [("B", i32), ("T2", i32), ("F2", i32), ("F3", i32), ("K3", i32), ("P3", i32), ("Hd", i32), ("N", i32), ("training", i32),
                ("eps", f32), ("momentum", f32), ("dropout_p", f32), ("salt", u32)]


class EegDeepParams(C.Structure):
    _fields_ = [(n, vp) for n in ("conv2_w", "bn4_w", "bn4_b", "bn4_rm", "bn4_rv", "bn4_nbt", "wq", "bq", "wk", "bk", "wv", "bv",
                                  "w1", "b1", "w2", "b2")]


class EegDeepGrads(C.Structure):
    _fields_ = [(n, vp) for n in ("conv2_w", "bn4_w", "bn4_b", "wq", "bq", "wk", "bk", "wv", "bv", "w1", "b1", "w2", "b2")]


class WgradPending(C.Structure):
    _fields_ = [("partial", vp), ("dw", vp), ("db", vp), ("nsplit", i32), ("Cin", i32), ("Co", i32), ("ma", i32), ("nb", i32),
                ("ztiles", i32), ("nfrag4", i32), ("valid", i32)]


class PackJob(C.Structure):
    _fields_ = [("w_oihw", vp), ("packed_mfma", vp), ("Cout", i32), ("Cin", i32), ("I_p", i32), ("O_p", i32),
                ("transpose_flip", i32), ("block_begin", i32)]


class EegGrads(C.Structure):
    _fields_ = [(n, vp) for n in ("conv1_w", "bn1_w", "bn1_b", "dw_w", "bn2_w", "bn2_b", "sep_w", "bn3_w", "bn3_b")]


P = C.POINTER
# name -> (restype, argtypes); must list every symbol include/brainxai.h declares (tests check this)
SIGNATURES = {
    "bx_version": (i32, []),
    "bx_last_error_string": (C.c_char_p, []),
    "bx_nchw_to_nhwc": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "bx_nhwc_to_nchw": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "bx_conv3x3_pack": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "bx_conv3x3_packed_mfma_bytes": (sz, [i32, i32]),
    "bx_conv3x3_packed_split_bytes": (sz, [i32, i32]),
    "bx_conv3x3_pack_split": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "bx_conv3x3_pack_many": (i32, [vp, i32, i32, vp]),
    "bx_conv3x3_pack_many_layout": (i32, [vp, i32, i32, vp, vp, i32, i32, i32, i32, i32, vp]),
    "bx_conv3x3_pack_layout_ex": (i32, [vp, i32, i32, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "bx_conv3x3_pair_supported": (i32, [i32, i32, i32, i32]),
    "bx_conv3x3_pair": (i32, [vp] * 9 + [i32] * 7 + [vp]),
    "bx_conv3x3_pack_many_step": (i32, [vp, i32, i32, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp]),
    "bx_scale_dev": (i32, [vp, vp, vp, sz, vp]),
    "bx_abs": (i32, [vp, vp, sz, vp]),
    "bx_conv3x3": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "bx_conv3x3_carry": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, P(WgradPending), vp]),
    "bx_conv3x3_wgrad_workspace": (sz, [i32, i32, i32, i32, i32, i32, i32]),
    "bx_conv3x3_wgrad": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, sz, vp]),
    "bx_conv3x3_wgrad_chained": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, sz, P(WgradPending), vp]),
    "bx_conv3x3_wgrad_finish": (i32, [P(WgradPending), vp]),
    "bx_block_tail_workspace": (sz, [P(TailDesc)]),
    "bx_block_tail_fwd": (i32, [P(TailDesc), vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    "bx_set_tree_max_rows": (i32, [i32]),
    "bx_set_tail_fold": (i32, [i32]),
    "bx_block_tail_route_bytes": (sz, [vp]),
    "bx_block_conv3_tail_fwd": (i32, [P(TailDesc), vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    "bx_block_tail_bwd": (i32, [P(TailDesc), vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    "bx_relu": (i32, [vp, vp, sz, i32, vp]),
    "bx_gap_fc_lsm_fwd": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "bx_gap_fc_lsm_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "bx_linear_lsm_fwd": (i32, [vp, vp, vp, vp, i32, i32, i32, vp]),
    "bx_linear_lsm_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "bx_fusion_head_fwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "bx_fusion_head_bwd": (i32, [vp] * 13 + [i32, i32, i32, vp]),
    "bx_mm_head_workspace": (sz, [i32, i32, i32]),
    "bx_mm_head_fwd": (i32, [vp] * 15 + [i32] * 7 + [vp]),
    "bx_mm_head_bwd": (i32, [vp] * 22 + [sz] + [i32] * 7 + [vp]),
    "bx_kldiv_fwd_bwd": (i32, [vp, vp, vp, vp, i32, i32, i32, f32, vp]),
    "bx_eeg_saved_bytes": (sz, [P(EegDesc)]),
    "bx_eeg_workspace": (sz, [P(EegDesc)]),
    "bx_eeg_features_fwd": (i32, [P(EegDesc), P(EegParams), vp, vp, vp, vp, vp, sz, vp]),
    "bx_eeg_features_bwd": (i32, [P(EegDesc), P(EegParams), vp, vp, vp, vp, P(EegGrads), vp, vp, sz, vp]),
    "bx_eeg_deep_saved_bytes": (sz, [P(EegDeepDesc)]),
    "bx_eeg_deep_workspace": (sz, [P(EegDeepDesc)]),
    "bx_eeg_deep_fwd": (i32, [P(EegDeepDesc), P(EegDeepParams), vp, vp, vp, vp, vp, vp, sz, vp]),
    "bx_eeg_deep_bwd": (i32, [P(EegDeepDesc), P(EegDeepParams), vp, vp, vp, vp, vp, P(EegDeepGrads), vp, vp, sz, vp]),
    "bx_attention_workspace": (sz, [i32]),
    "bx_attention_fwd": (i32, [vp] * 10 + [i32, i32, i32, vp]),
    "bx_attention_bwd": (i32, [vp] * 16 + [sz, i32, i32, i32, vp]),
    "bx_gradcam_reduce": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "bx_gradcam_head": (i32, [vp] * 12 + [i32] * 8 + [vp]),
    "bx_gradcam_head_sweep": (i32, [vp] * 4 + [i32] + [vp] * 8 + [i32] * 11 + [vp]),
    "bx_resize_bilinear": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "bx_saliency_reduce": (i32, [vp, vp, i32, i32, i32, i32, f32, i32, vp]),
    "bx_axpby": (i32, [vp, vp, sz, f32, f32, vp]),
    "bx_mul": (i32, [vp, vp, vp, sz, vp]),
    "bx_ig_interpolate": (i32, [vp, vp, vp, vp, sz, i32, vp]),
    "bx_ig_accumulate": (i32, [vp, vp, vp, sz, i32, vp]),
    "bx_eeg_stack_iir": (i32, [vp, vp, vp, i32, i32, i32, i32, P(C.c_double), P(C.c_double), i32, i32, f32, f32, vp]),
    "bx_eeg_montage_workspace": (sz, [i32, i32, i32, i32]),
    "bx_eeg_montage_stack": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, P(C.c_double), P(C.c_double), i32, P(C.c_double), P(C.c_double),
                                   i32, f32, vp, vp, sz, vp]),
    "bx_spec_preprocess_workspace": (sz, [i32, i32, i32]),
    "bx_spec_preprocess": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, P(C.c_double), P(C.c_double), P(C.c_double), P(C.c_double),
                                 f32, vp, vp, sz, vp]),
    "bx_spec_regions_workspace": (sz, [i32]),
    "bx_spec_regions": (i32, [vp, vp, vp] + [i32] * 7 + [P(C.c_double), i32, P(C.c_double), i32, f32, vp, sz, vp]),
    "bx_adamw_step": (i32, [vp, vp, vp, vp, sz, f32, f32, f32, f32, f32, f32, vp, vp]),
    "bx_adamw_partials": (sz, [sz]),
    "bx_adamw_step_words": (sz, [sz]),
    "bx_adamw_step_dev": (i32, [vp, vp, vp, vp, sz, vp, vp, vp, vp, vp]),
    "bx_store_f32x8": (i32, [vp, vp, vp]),
    "bx_store_u64x2": (i32, [vp, C.c_uint64, C.c_uint64, vp]),
    "bx_profile_next_conv3": (i32, [vp, vp]),
    "bx_sumsq": (i32, [vp, sz, vp, vp]),
    "bx_u8_to_nhwc": (i32, [vp, vp, i32, i32, i32, i32, i32, f32, i32, vp]),
    "bx_softmax_rows": (i32, [vp, vp, i32, i32, vp]),
    "bx_class_seed": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "bx_seed_next": (i32, [vp, vp, vp]),
    "bx_seed_next2": (i32, [vp, vp, vp, vp, vp]),
}

_lib = None


def load():
    """Load (once) and return the ctypes handle; raise loudly when the extension is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"brainxai: HIP extension not built ({LIB_PATH} missing). Run `python __graft_entry__.py` "
            "or `python multimodal-brain-pattern-identification_xai_amd/build.py`; there is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype, fn.argtypes = res, args
    if lib.bx_version() != 100:
        raise RuntimeError(f"brainxai: library version {lib.bx_version()} != 100")
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().bx_last_error_string()
        raise RuntimeError(f"brainxai {what} failed (code {rc}): {msg.decode() if msg else ''}")
