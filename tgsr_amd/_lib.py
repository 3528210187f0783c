"""ctypes binding of libtgsr_hip.so (the C ABI declared in include/tgsr_hip.h).

The product has no CPU or eager-PyTorch fallback: if the HIP library is missing, loading raises, and every op
refuses non-HIP tensors.  `build()` compiles the library in-tree (tgsr_amd/lib/) with hipcc for gfx950.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# TGSR_LIB_PATH: another build of the same library (the stamp / experiment builds of tools/); never a different implementation
LIB_PATH = os.environ.get("TGSR_LIB_PATH") or os.path.join(_HERE, "lib", "libtgsr_hip.so")
ABI_VERSION = 2

OK, EINVAL, EUNSUPPORTED, ELAUNCH = 0, -1, -2, -3
EPI_AFFINE, EPI_AFFINE_GLU = 0, 1
ACT_NONE, ACT_TANH_AXPY = 0, 1
DT_BF16, DT_F16 = 1, 2

_vp, _i, _i64, _f, _d = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double

# name -> (restype, argtypes); must list every function of include/tgsr_hip.h (tests/test_abi.py checks it)
SIGNATURES = {
    "tgsr_abi_version": (_i, []),
    "tgsr_last_error": (ctypes.c_char_p, []),
    "tgsr_packed_weight_elems": (_i64, [_i, _i, _i]),
    "tgsr_pack_conv_weight": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "tgsr_pack_conv_weight_dgrad": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "tgsr_bn_fold": (_i, [_vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _vp]),
    "tgsr_conv3x3_fwd": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _i64, _vp, _i64, _i, _i, _vp]),
    "tgsr_packed_upconv_weight_elems": (_i64, [_i, _i]),
    "tgsr_pack_upconv_weight": (_i, [_vp, _vp, _i, _i, _vp]),
    "tgsr_upconv3x3_glu_fwd": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _i64, _vp]),
    "tgsr_packed_upwino_weight_elems": (_i64, [_i, _i]),
    "tgsr_pack_upwino_weight": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "tgsr_upwino_glu_fwd": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _i64, _vp]),
    "tgsr_upwino_fwd": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _i64, _vp]),
    "tgsr_packed_upwino4_weight_elems": (_i64, [_i, _i]),
    "tgsr_pack_upwino4_weight": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "tgsr_upwino4_fwd": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _i64, _i, _vp]),
    "tgsr_packed_wino_weight_elems": (_i64, [_i, _i]),
    "tgsr_pack_wino_weight": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "tgsr_pack_wino_weight_dgrad": (_i, [_vp, _vp, _i, _i, _vp]),
    "tgsr_wino_conv3x3_fwd": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _i64, _vp, _i64, _i, _vp]),
    "tgsr_packed_wino4_weight_elems": (_i64, [_i, _i]),
    "tgsr_pack_wino4_weight": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "tgsr_pack_wino4_weight_dgrad": (_i, [_vp, _vp, _i, _i, _vp]),
    "tgsr_pack_wino4_wide_weight": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "tgsr_pack_wino4_wide_weight_dgrad": (_i, [_vp, _vp, _i, _i, _vp]),
    "tgsr_wino4_wide_stats_nslots": (_i, [_i, _i, _i, _i]),
    "tgsr_wino4_wide_conv3x3_stats_fwd": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _i, _vp, _i64, _vp, _vp]),
    "tgsr_wino4_wide_conv3x3_fwd": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _i64, _vp, _i64, _i, _vp]),
    "tgsr_wino4_conv3x3_fwd": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _i64, _vp, _i64, _i, _vp]),
    "tgsr_conv_to3_fwd": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _i, _i, _vp, _f, _vp, _vp]),
    "tgsr_word_attention_fwd": (_i, [_vp, _i64, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _i64, _vp, _vp]),
    "tgsr_word_project_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "tgsr_bilstm_fwd": (_i, [_vp, _i, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "tgsr_lstm_gate_table": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "tgsr_bilstm_table_fwd": (_i, [_vp, _i, _vp, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp]),
    "tgsr_bilstm_train_fwd": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "tgsr_bilstm_bwd": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tgsr_damsm_words_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp]),
    "tgsr_damsm_words_bwd_ws_elems": (_i64, [_i, _i, _i]),
    "tgsr_damsm_words_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp]),
    "tgsr_func_attention_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp]),
    "tgsr_conv1x1_fwd": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp]),
    "tgsr_linear_fwd": (_i, [_vp, _i, _i, _vp, _vp, _i, _vp, _vp]),
    "tgsr_rowdot_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "tgsr_ca_net_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "tgsr_rowdot_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "tgsr_bn_train_nsplit": (_i, [_i, _i, _i]),
    "tgsr_bn_train_fwd": (_i, [_vp, _i, _i, _i, _vp, _vp, _f, _f, _vp, _vp, _i, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp,
                               _i64, _vp, _vp]),
    "tgsr_bn_train_bwd": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tgsr_dconv_set_split": (_i, [_i]),
    "tgsr_bn_set_fuse_small": (_i, [_i]),
    "tgsr_conv_to3_set_pipe": (_i, [_i]),
    "tgsr_conv4x4s2_split_form": (_i, [_i, _i, _i, _i, _i, _i]),
    "tgsr_conv3x3_gemm_split_form": (_i, [_i, _i, _i, _i, _i, _i]),
    "tgsr_conv4x4s2_ws_elems": (_i64, [_i, _i, _i, _i, _i, _i]),
    "tgsr_conv4x4s2_fwd": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp]),
    "tgsr_conv4x4s2_dgrad": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "tgsr_conv4x4s2_wgrad": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "tgsr_conv3x3_gemm_ws_elems": (_i64, [_i, _i, _i, _i, _i, _i]),
    "tgsr_conv3x3_gemm_fwd": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "tgsr_conv3x3_gemm_dgrad": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "tgsr_conv3x3_gemm_wgrad": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "tgsr_leaky_relu": (_i, [_vp, _vp, _vp, _i64, _vp]),
    "tgsr_glu": (_i, [_vp, _vp, _vp, _i64, _i64, _vp]),
    "tgsr_wino_stats_nslots": (_i, [_i, _i, _i, _i]),
    "tgsr_wino_conv3x3_stats_fwd": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _i, _vp, _i64, _vp, _vp]),
    "tgsr_wino4_stats_nslots": (_i, [_i, _i, _i, _i]),
    "tgsr_wino4_conv3x3_stats_fwd": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _i, _vp, _i64, _vp, _vp]),
    "tgsr_bn_train_fwd_from_stats": (_i, [_vp, _i, _i, _i, _vp, _vp, _f, _f, _vp, _vp, _i, _vp, _i64, _vp, _i, _vp, _vp, _vp, _vp,
                                          _vp, _i64, _vp, _vp]),
    "tgsr_text_tail_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "tgsr_gru_gate_table": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "tgsr_bigru_table_fwd": (_i, [_vp, _i, _vp, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "tgsr_bigru_train_fwd": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "tgsr_bigru_bwd": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tgsr_lp_att_pack_bytes": (_i64, [_i, _i]),
    "tgsr_text_tail_lp_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp]),
    "tgsr_lp_stem_att_fwd": (_i, [_i, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "tgsr_lp_upconv_glu_att_fwd": (_i, [_i, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _vp, _i, _i,
                                        _i, _i, _i, _i, _vp, _vp]),
    "tgsr_multi_copy": (_i, [_i, _vp, _vp, _vp, _vp]),
    "tgsr_axpy_images": (_i, [_i, _vp, _vp, _vp, _vp, _f, _vp]),
    "tgsr_gconv_set_form": (_i, [_i]),
    "tgsr_gconv_nsplit": (_i, [_i, _i, _i]),
    "tgsr_gconv_ws_elems": (_i64, [_i, _i, _i, _i, _i]),
    "tgsr_gconv": (_i, [_i, _vp, _vp, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _i64, _vp, _vp]),
    "tgsr_gconv_pack": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "tgsr_maxpool3s2_fwd": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _i64, _vp]),
    "tgsr_maxpool3s2_bwd": (_i, [_vp, _i64, _vp, _i64, _i, _i, _i, _i, _vp, _i64, _i, _vp, _vp]),
    "tgsr_avgpool3": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _i64, _i, _vp, _vp]),
    "tgsr_sum_stack": (_i, [_vp, _i, _i64, _vp, _vp]),
    "tgsr_interleave2x2": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp, _vp]),
    "tgsr_plane_mean": (_i, [_vp, _vp, _i64, _i, _vp]),
    "tgsr_plane_mean_bwd": (_i, [_vp, _vp, _i64, _i, _vp]),
    "tgsr_relu_mask": (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _i, _i64, _vp]),
    "tgsr_bilinear_fwd": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _vp]),
    "tgsr_bilinear_bwd": (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _vp]),
    "tgsr_adam_flat": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _d, _d, _d, _d, _d, _i, _vp]),
    "tgsr_weighted_bce_fwd": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    "tgsr_weighted_bce_bwd": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "tgsr_axpy_map_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "tgsr_axpy_map_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "tgsr_affine_act_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "tgsr_affine_act_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "tgsr_comm_available": (_i, []),
    "tgsr_comm_unique_id": (_i, [_vp]),
    "tgsr_comm_init": (_i, [_vp, _vp, _i, _i]),
    "tgsr_allreduce_flat": (_i, [_vp, _vp, _i64, _f, _vp]),
    "tgsr_comm_count": (_i, [_vp, _vp, _vp]),
    "tgsr_comm_destroy": (_i, [_vp]),
    "tgsr_sumpool2x2": (_i, [_vp, _i64, _i, _i, _vp, _vp]),
    "tgsr_resize_bilinear_u8": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "tgsr_gaussian_blur_u8": (_i, [_vp, _i, _i, _i, _i, ctypes.c_uint32, ctypes.c_uint32, _i, _vp, _vp, _vp]),
    "tgsr_u8_normalize": (_i, [_vp, _vp, _i64, _vp]),
    "tgsr_to_uint8": (_i, [_vp, _vp, _i64, _vp]),
    "tgsr_conv3x3_wgrad_ws_elems": (_i64, [_i, _i, _i, _i, _i, _i]),
    "tgsr_conv3x3_wgrad": (_i, [_vp, _vp, _i64, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "tgsr_upwino_wgrad_ws_elems": (_i64, [_i, _i, _i, _i, _i]),
    "tgsr_upwino_wgrad": (_i, [_vp, _vp, _i64, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "tgsr_wino_wgrad_ws_elems": (_i64, [_i, _i, _i, _i, _i]),
    "tgsr_wino_wgrad": (_i, [_vp, _vp, _i64, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "tgsr_word_attention_bwd_chunks": (_i, [_i]),
    "tgsr_word_attention_bwd": (_i, [_vp, _i64, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "tgsr_conv_to3_bwd_ws_elems": (_i64, [_i, _i, _i, _i, _i]),
    "tgsr_conv_to3_bwd": (_i, [_vp, _vp, _vp, _f, _vp, _i64, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    # reduced-precision inference path (lp images: zero-bordered channels-last bf16 / f16)
    "tgsr_lp_from_nchw": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "tgsr_lp_to_nchw": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "tgsr_lp_convert": (_i, [_i, _vp, _i, _vp, _i64, _vp]),
    "tgsr_lp_packed_conv3x3_elems": (_i64, [_i, _i]),
    "tgsr_lp_pack_conv3x3_weight": (_i, [_i, _vp, _vp, _i, _i, _vp]),
    "tgsr_lp_conv3x3_fwd": (_i, [_i, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _vp]),
    "tgsr_lp_resblocks_flag_elems": (_i64, [_i, _i, _i]),
    "tgsr_lp_resblocks_fwd": (_i, [_i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _i, _vp, _vp]),
    "tgsr_lp_packed_upconv_elems": (_i64, [_i, _i]),
    "tgsr_lp_pack_upconv_weight": (_i, [_i, _vp, _vp, _i, _i, _vp]),
    "tgsr_lp_upconv_glu_fwd": (_i, [_i, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _vp]),
    "tgsr_lp_head_partial_elems": (_i64, [_i, _i, _i, _i]),
    "tgsr_lp_upconv_glu_head_fwd": (_i, [_i, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _vp]),
    "tgsr_lp_head_combine": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _vp]),
    "tgsr_lp_stem_fwd": (_i, [_i, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _vp]),
    "tgsr_lp_pack_to3_weight": (_i, [_i, _vp, _vp, _i, _i, _vp]),
    "tgsr_lp_conv_to3_fwd": (_i, [_i, _vp, _i, _i, _i, _i, _i, _vp, _i, _i, _vp, _f, _vp, _vp]),
    "tgsr_lp_word_attention_fwd": (_i, [_i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp]),
}

_lib = None


class TgsrError(RuntimeError):
    pass


def build(verbose=False):
    """Compile every HIP source under tgsr_amd/csrc for gfx950 into tgsr_amd/lib/libtgsr_hip.so."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout)
    if r.returncode != 0:
        raise TgsrError("building libtgsr_hip.so failed (see output above)")
    return LIB_PATH


def lib():
    """The loaded library; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TgsrError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` or "
                            "`make -C tgsr_amd/csrc`; tgsr_amd has no CPU fallback" % LIB_PATH)
        # torch bundles its own libamdhip64 (same SONAME as /opt/rocm's): it must be in the process first so the
        # dynamic linker binds our library to THAT runtime - two HIP runtimes in one process do not share a device.
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError = a header function the library does not export
            fn.restype = res
            fn.argtypes = args
        v = L.tgsr_abi_version()
        if v != ABI_VERSION:
            raise TgsrError("libtgsr_hip.so ABI %d != binding ABI %d: rebuild" % (v, ABI_VERSION))
        _lib = L
    return _lib


def check(rc, what):
    if rc == OK:
        return
    names = {EINVAL: "TGSR_EINVAL", EUNSUPPORTED: "TGSR_EUNSUPPORTED", ELAUNCH: "TGSR_ELAUNCH"}
    msg = "%s failed: %s" % (what, names.get(rc, rc))
    if rc == ELAUNCH:
        msg += " (%s)" % lib().tgsr_last_error().decode()
    raise TgsrError(msg)
