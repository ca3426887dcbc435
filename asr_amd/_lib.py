"""ctypes binding of libasr_amd.so (the C ABI declared in include/asr_amd.h).

The HIP library is the product path: if it is missing or fails to load this
module raises -- there is no CPU or eager-PyTorch fallback anywhere in
``asr_amd``.  Build it with ``make`` (or ``__graft_entry__.build()``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libasr_amd.so")

c_f = C.c_void_p      # device pointers travel as integers (tensor.data_ptr())
c_i64 = C.c_int64


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", c_f), ("B", c_f), ("C", c_f),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("a_mode", C.c_int), ("b_mode", C.c_int),
        ("a_s0", c_i64), ("a_s1", c_i64), ("a_d", C.c_int),
        ("b_s0", c_i64), ("b_s1", c_i64), ("b_d", C.c_int),
        ("c_s0", c_i64), ("c_s1", c_i64), ("c_d", C.c_int),
        ("bias1", c_f), ("bias2", c_f),
        ("alpha", C.c_float),
        ("accumulate", C.c_int),
        ("act", C.c_int),
        ("splitk", C.c_int),
        ("nbatch", C.c_int),
        ("batch_a", c_i64), ("batch_b", c_i64), ("batch_c", c_i64),
        ("m_idx", c_f), ("k_idx", c_f),
        ("bias_d", C.c_int), ("bias1_s1", c_i64), ("bias2_s1", c_i64),
        ("c_keep", c_f), ("c_keep_scale", C.c_float),
        ("batch_d", C.c_int), ("batch_a1", c_i64), ("batch_b1", c_i64), ("batch_c1", c_i64),
        ("pad_ok", C.c_int),
        ("a_span", c_i64), ("b_span", c_i64),
        ("a_packed", C.c_int), ("b_packed", C.c_int),
    ]


TENSOR_LIST_MAX = 48


COLSUM_LIST_MAX = 8


class ColsumList(C.Structure):
    _fields_ = [("count", C.c_int), ("src", c_f * COLSUM_LIST_MAX), ("out", c_f * COLSUM_LIST_MAX),
                ("n", C.c_int * COLSUM_LIST_MAX), ("off", c_i64 * COLSUM_LIST_MAX)]


class TensorList(C.Structure):
    _fields_ = [("count", C.c_int), ("p", c_f * TENSOR_LIST_MAX), ("n", c_i64 * TENSOR_LIST_MAX)]


class AttDecFwdArgs(C.Structure):
    _fields_ = [
        ("B", C.c_int), ("T", C.c_int), ("E", C.c_int), ("Hd", C.c_int), ("L", C.c_int), ("V", C.c_int),
        ("hbatch", c_f), ("PH", c_f), ("lengths", c_f), ("targets", c_f),
        ("W_se", c_f), ("w_ee", c_f), ("W_fe", c_f), ("conv_w", c_f),
        ("Emb", c_f), ("W_ss", c_f), ("W_gs", c_f), ("b_gs", c_f),
        ("S", c_f), ("C", c_f), ("alpha", c_f), ("sproj", c_f), ("g", c_f), ("gates", c_f), ("Fc", c_f),
        ("ws", c_f),
    ]


class AttDecBwdArgs(C.Structure):
    _fields_ = [
        ("f", AttDecFwdArgs),
        ("W_seT", c_f), ("W_ssT", c_f), ("W_syT", c_f), ("W_gsT", c_f), ("W_gyT", c_f),
        ("dyp", c_f),
        ("drec", c_f), ("dsp", c_f), ("dg", c_f), ("de", c_f), ("dF", c_f),
        ("dPH", c_f), ("dw_ee", c_f), ("dW_fe", c_f), ("dconv_w", c_f),
        ("ws", c_f),
        ("defer_tail", C.c_int),
    ]


# name -> (restype, argtypes); every symbol include/asr_amd.h declares
SYMBOLS = {
    "asr_version": (C.c_int, []),
    "asr_last_error": (C.c_char_p, []),
    "asr_debug_set": (None, [C.c_int]),
    "asr_ls_ce_fwd_bwd": (C.c_int, [c_f, c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int,
                                    c_f, c_f, c_f, c_f]),
    "asr_sgemm": (C.c_int, [C.POINTER(GemmDesc), c_f]),
    "asr_gemm_precision": (C.c_int, [C.c_int]),
    "asr_colsum": (C.c_int, [c_f, C.c_int, C.c_int, c_i64, c_i64, C.c_int, c_f, C.c_int, c_f, c_f]),
    "asr_colsum_ws_floats": (c_i64, [C.c_int, C.c_int]),
    "asr_colsum_multi_ws_floats": (c_i64, [C.POINTER(ColsumList), C.c_int]),
    "asr_colsum_multi": (C.c_int, [C.POINTER(ColsumList), C.c_int, c_f, c_f]),
    "asr_transpose": (C.c_int, [c_f, c_f, C.c_int, C.c_int, C.c_int, c_f]),
    "asr_scatter_rows": (C.c_int, [c_f, c_f, C.c_int, C.c_int, c_i64, c_f, C.c_int, c_f]),
    "asr_mul_dtanh": (C.c_int, [c_f, c_f, c_f, c_i64, c_f]),
    "asr_scale": (C.c_int, [c_f, c_i64, C.c_float, c_f]),
    "asr_argmax_rows": (C.c_int, [c_f, c_i64, C.c_int, c_i64, c_f, c_f]),
    "asr_sqnorm_ws_floats": (c_i64, []),
    "asr_featurize_ws_floats": (c_i64, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "asr_featurize": (C.c_int, [c_f, c_f, c_f, c_f, c_f, C.c_int, C.c_int, c_f, C.c_int, C.c_int,
                                C.c_int, C.c_int, C.c_int, c_f, c_f, c_f]),
    "asr_ctc_ws_floats": (c_i64, [C.c_int, C.c_int, C.c_int]),
    "asr_ctc_loss_fwd_bwd": (C.c_int, [c_f, c_f, c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       c_f, c_f, c_f, c_f, c_f]),
    "asr_grad_sqnorm": (C.c_int, [C.POINTER(TensorList), c_f, C.c_int, c_f, c_f]),
    "asr_clip_adam_step": (C.c_int, [C.POINTER(TensorList)] * 4 + [c_f] + [C.c_double] * 6 + [c_i64, c_f]),
    "asr_pack_hilo": (C.c_int, [c_f, c_f, c_i64, c_f, c_f]),
    "asr_dropout_fwd_packed": (C.c_int, [c_f, c_f, c_f, c_i64, C.c_float, C.c_uint64, c_f]),
    "asr_blstm_ws_floats": (c_i64, [C.c_int, C.c_int]),
    "asr_blstm_precision": (C.c_int, [C.c_int]),
    "asr_blstm_xcd_groups": (C.c_int, [C.c_int]),
    "asr_blstm_prefetch": (C.c_int, [C.c_int]),
    "asr_blstm_bwd_needs_transposed": (C.c_int, [C.c_int, C.c_int, C.c_int, c_i64, c_i64, C.c_int]),
    "asr_blstm_fwd": (C.c_int, [c_f, c_f, c_f, c_f, c_i64, c_i64, c_f, C.c_int, C.c_int, C.c_int, c_f, c_f]),
    "asr_blstm_fwd_noalloc_max_floats": (c_i64, []),
    "asr_blstm_fwd_noalloc_err_word": (C.c_void_p, []),
    "asr_blstm_fwd_noalloc": (C.c_int, [c_f, c_f, c_f, c_f, c_i64, c_i64, c_f, C.c_int, C.c_int, C.c_int, c_f]),
    "asr_poison_if_flag": (C.c_int, [c_f, c_f, c_f]),
    "asr_debug_occupy": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, c_f]),
    "asr_stream_create": (C.c_int, [C.POINTER(C.c_void_p), c_f, C.c_int]),
    "asr_stream_destroy": (C.c_int, [c_f]),
    "asr_blstm_bwd": (C.c_int, [c_f, c_f, c_f, c_f, c_f, c_i64, c_i64, c_f, c_f, C.c_int, C.c_int, C.c_int,
                                c_f, c_f]),
    "asr_blstm_fwd_range": (C.c_int, [c_f, c_f, c_f, c_f, c_i64, c_i64, c_f, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, c_f, c_f]),
    "asr_blstm_bwd_range": (C.c_int, [c_f, c_f, c_f, c_f, c_f, c_i64, c_i64, c_f, c_f, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, c_f, c_f]),
    "asr_dropout_fwd_block": (C.c_int, [c_f, c_f, c_f, c_i64, c_i64, c_i64, c_i64, C.c_float, C.c_uint64, c_f]),
    "asr_dropout_fwd": (C.c_int, [c_f, c_f, c_f, c_i64, C.c_float, C.c_uint64, c_f]),
    "asr_dropout_bwd": (C.c_int, [c_f, c_f, c_f, c_i64, C.c_float, c_f]),
    "asr_attdec_ws_floats": (c_i64, [C.c_int] * 5),
    "asr_attdec_plan": (C.c_int, [C.c_int] * 5),
    "asr_attdec_fwd": (C.c_int, [C.POINTER(AttDecFwdArgs), c_f]),
    "asr_attdec_persist": (C.c_int, [C.c_int]),
    "asr_attdec_debug_stamps": (None, [c_f]),
    "asr_attdec_pace": (C.c_int, [C.c_int, C.c_int]),
    "asr_attdec_err_offset": (c_i64, [C.c_int] * 5),
    "asr_attdec_greedy": (C.c_int, [C.POINTER(AttDecFwdArgs), c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
    "asr_attdec_step_att": (C.c_int, [C.POINTER(AttDecFwdArgs), c_f, c_f]),
    "asr_attdec_step_cell": (C.c_int, [C.POINTER(AttDecFwdArgs), c_f]),
    "asr_attdec_bwd_ws_floats": (c_i64, [C.c_int] * 5),
    "asr_attdec_bwd_err_offset": (c_i64, [C.c_int] * 5),
    "asr_attdec_bwd": (C.c_int, [C.POINTER(AttDecBwdArgs), c_f]),
    "asr_attdec_bwd_tail": (C.c_int, [C.POINTER(AttDecBwdArgs), c_f]),
    # Conformer encoder (csrc/conformer.hip)
    "asr_layernorm_fwd": (C.c_int, [c_f] * 6 + [c_i64, C.c_int, C.c_float, c_f]),
    "asr_layernorm_bwd_blocks": (c_i64, [c_i64]),
    "asr_layernorm_bwd": (C.c_int, [c_f] * 7 + [c_i64, C.c_int, c_f, c_f]),
    "asr_swish_fwd": (C.c_int, [c_f, c_f, c_f, c_i64, C.c_float, C.c_uint64, c_f]),
    "asr_swish_bwd": (C.c_int, [c_f, c_f, c_f, c_f, c_i64, C.c_float, c_f]),
    "asr_residual_dropout": (C.c_int, [c_f, c_f, c_f, c_f, c_i64, C.c_float, C.c_float, C.c_uint64, C.c_int, c_f]),
    "asr_scale_mask": (C.c_int, [c_f, c_f, c_f, c_i64, C.c_float, c_f]),
    "asr_glu_fwd": (C.c_int, [c_f, c_f, c_i64, C.c_int, c_f]),
    "asr_glu_bwd": (C.c_int, [c_f, c_f, c_f, c_i64, C.c_int, c_f]),
    "asr_relu_bwd": (C.c_int, [c_f, c_f, c_f, c_i64, c_f]),
    "asr_axpby": (C.c_int, [c_f, c_f, c_f, c_i64, C.c_float, C.c_float, c_f]),
    "asr_add_two_bias": (C.c_int, [c_f] * 5 + [c_i64, C.c_int, c_f]),
    "asr_dwconv": (C.c_int, [c_f] * 4 + [C.c_int] * 5 + [c_f]),
    "asr_dwconv_bwd_w_chunks": (c_i64, [C.c_int, C.c_int]),
    "asr_dwconv_bwd_w": (C.c_int, [c_f] * 3 + [C.c_int] * 4 + [c_f]),
    "asr_bn_blocks": (c_i64, [c_i64]),
    "asr_bn_var_part": (C.c_int, [c_f, c_f, C.c_float, c_f, c_i64, C.c_int, c_f]),
    "asr_bn_swish_fwd": (C.c_int, [c_f, c_f, c_f, C.c_float, C.c_float, c_f, c_f, c_f, c_i64, C.c_int, C.c_float, c_f]),
    "asr_bn_swish_bwd_part": (C.c_int, [c_f, c_f, c_f, c_f, C.c_float, C.c_float, c_f, c_f, c_f, c_i64, C.c_int,
                                        C.c_float, c_f]),
    "asr_bn_swish_bwd": (C.c_int, [c_f, c_f, c_f, c_f, C.c_float, C.c_float, c_f, c_f, c_f, c_f, c_i64, C.c_int,
                                   C.c_float, c_f]),
    "asr_relattn_softmax_fwd": (C.c_int, [c_f, c_f, c_f, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, c_f, c_f,
                                          C.c_float, C.c_uint64, c_f]),
    "asr_relattn_softmax_bwd": (C.c_int, [c_f, c_f, c_f, c_f, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, c_f,
                                          C.c_float, c_f]),
    "asr_im2col_3x3s2": (C.c_int, [c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_f]),
    "asr_col2im_3x3s2": (C.c_int, [c_f, c_f, c_f, C.c_int, C.c_int, C.c_int, C.c_int, c_f]),
    "asr_col2im_dw_blocks": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "asr_col2im_3x3s2_dw": (C.c_int, [c_f, c_f, c_f, C.c_int, c_f, C.c_int, C.c_int, C.c_int, C.c_int, c_f]),
}

_lib = None


def load():
    """Load the library once and bind every symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"asr_amd: {LIB_PATH} not found -- the HIP extension is required (run `make` in the repo root); "
            "there is no fallback path")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)      # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise RuntimeError("asr_amd: " + load().asr_last_error().decode("utf-8", "replace") + f" (code {rc})")
