"""ctypes binding of libser_hip.so (the C ABI declared in include/ser_hip.h).

The product path has no CPU or eager-PyTorch fallback: if the shared library is
missing this module raises at import, and every wrapper raises `SerHipError` on a
non-zero return code.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libser_hip.so")

ACT_NONE, ACT_GELU, ACT_RELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3, 4
PREC_BF16, PREC_BF16X3 = 0, 1
MAX_CONV = 8


class SerHipError(RuntimeError):
    pass


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        f"(or `make -C {os.path.join(_HERE, 'csrc')}`).  There is no fallback path.")
lib = C.CDLL(LIB_PATH)

vp, i32, i64, f32, sz = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_size_t


class SplitW(C.Structure):
    _fields_ = [("hi", vp), ("lo", vp)]


class LayerW(C.Structure):
    _fields_ = [("qkv", SplitW), ("qkv_b", vp), ("o", SplitW), ("o_b", vp), ("ln1_g", vp), ("ln1_b", vp),
                ("f1", SplitW), ("f1_b", vp), ("f2", SplitW), ("f2_b", vp), ("ln2_g", vp), ("ln2_b", vp)]


class W2vConfig(C.Structure):
    _fields_ = [("hidden", i32), ("layers", i32), ("heads", i32), ("ffn", i32), ("n_conv", i32),
                ("conv_dim", i32 * MAX_CONV), ("conv_kernel", i32 * MAX_CONV), ("conv_stride", i32 * MAX_CONV),
                ("pos_kernel", i32), ("pos_groups", i32), ("eps", f32)]


class W2vWeights(C.Structure):
    _fields_ = [("conv0_w", vp), ("gn_g", vp), ("gn_b", vp), ("conv_w", SplitW * MAX_CONV), ("fp_ln_g", vp),
                ("fp_ln_b", vp), ("fp_w", SplitW), ("fp_b", vp), ("pos_w", SplitW), ("pos_b", vp), ("enc_ln_g", vp),
                ("enc_ln_b", vp), ("layers", C.POINTER(LayerW))]


class XlmrConfig(C.Structure):
    _fields_ = [("hidden", i32), ("layers", i32), ("heads", i32), ("ffn", i32), ("vocab", i32), ("max_pos", i32),
                ("pad_id", i32), ("eps", f32)]


class XlmrWeights(C.Structure):
    _fields_ = [("word_emb", vp), ("pos_emb", vp), ("type_emb", vp), ("emb_ln_g", vp), ("emb_ln_b", vp),
                ("layers", C.POINTER(LayerW))]


def _sig(name, restype, *argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = list(argtypes)
    return fn


_sig("ser_last_error_string", C.c_char_p)
_sig("ser_abi_version", i32)
_sig("ser_stream_create_cu_masked", i32, C.POINTER(C.c_uint32), i32, C.POINTER(vp))
_sig("ser_stream_destroy", i32, vp)
_sig("ser_prof_gemm_start", i32)
_sig("ser_prof_gemm_stop", i32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_longlong))
_sig("ser_split_bf16", i32, vp, vp, vp, i64, vp)
_sig("ser_gemm_bf16_nt", i32, vp, vp, i32, vp, vp, i32, i32, i32, i32, vp, i32, vp, i32, vp, vp, vp, i32, vp)
_sig("ser_layernorm", i32, vp, vp, vp, vp, f32, i32, i32, vp, vp, vp, vp)
_sig("ser_self_attention", i32, vp, vp, vp, i32, i32, i32, vp, vp, vp)
_sig("ser_wav2vec2_workspace_bytes", sz, C.POINTER(W2vConfig), i32, i32, i32)
_sig("ser_wav2vec2_out_len", i32, C.POINTER(W2vConfig), i32)
_sig("ser_wav2vec2_forward", i32, C.POINTER(W2vConfig), C.POINTER(W2vWeights), vp, i32, i32, i32, vp, vp, sz, vp)
_sig("ser_xlmr_workspace_bytes", sz, C.POINTER(XlmrConfig), i32, i32, i32)
_sig("ser_xlmr_forward", i32, C.POINTER(XlmrConfig), C.POINTER(XlmrWeights), vp, vp, i32, i32, i32, vp, vp, sz, vp)
_sig("ser_encoders_forward", i32, C.POINTER(W2vConfig), C.POINTER(W2vWeights), vp, i32, i32, C.POINTER(XlmrConfig),
     C.POINTER(XlmrWeights), vp, vp, i32, i32, i32, vp, vp, vp, sz, vp, sz, vp)


def check(rc, what=""):
    if rc != 0:
        msg = lib.ser_last_error_string()
        raise SerHipError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "libser_hip needs contiguous device tensors"
    return t.data_ptr()


def cu_masked_stream(bits):
    """torch ExternalStream on a HIP stream limited to the CUs listed in `bits` (iterable of CU indices)."""
    words = [0] * 8
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    arr = (C.c_uint32 * 8)(*words)
    out = vp()
    check(lib.ser_stream_create_cu_masked(arr, 8, C.byref(out)), "ser_stream_create_cu_masked")
    return torch.cuda.ExternalStream(out.value)


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


# ---- thin op-level wrappers (used by tests and by weight packing) ---------------------------------

def split_bf16(x, want_lo=True):
    """fp32 device tensor -> (hi, lo) bf16 planes computed by the library's kernel."""
    x = x.contiguous()
    assert x.dtype == torch.float32
    hi = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    lo = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device) if want_lo else None
    check(lib.ser_split_bf16(ptr(x), ptr(hi), ptr(lo), x.numel(), stream_ptr()), "ser_split_bf16")
    return hi, lo


IL_GROUP = 32        # interleaved split planes: [hi 0..31 | lo 0..31 | hi 32..63 | ...] (csrc/ser_common.h)


def split_bf16_il(x):
    """fp32 device tensor [..., D] (D % 32 == 0) -> ONE bf16 tensor [..., 2 D] holding both planes interleaved in groups
    of 32.  The library recognises the layout by `lo == hi + 32 elements` (see `il_ptrs`)."""
    x = x.contiguous()
    assert x.dtype == torch.float32 and x.shape[-1] % IL_GROUP == 0
    out = torch.empty(x.shape[:-1] + (2 * x.shape[-1],), dtype=torch.bfloat16, device=x.device)
    check(lib.ser_split_bf16(ptr(x), out.data_ptr(), out.data_ptr() + 2 * IL_GROUP, x.numel(), stream_ptr()), "ser_split_bf16")
    return out


_sig("ser_split_bf16_t", i32, vp, i32, i32, i64, vp, vp, i32, vp)


def split_bf16_t(x, want_lo=True, pad=64, out=None):
    """fp32 [R, C] device tensor (rows may be strided: a view like x[j::s]) -> split planes of x^T with the R axis zero-padded to
    a multiple of `pad`: want_lo: ONE interleaved bf16 tensor [C, 2 Rp]; else the hi plane [C, Rp].  out: a contiguous row block
    of a larger planes tensor to write into (stacking the transposes of several views under each other)."""
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    R, Cc = x.shape
    Rp = (R + pad - 1) // pad * pad
    if out is None:
        out = torch.empty(Cc, (2 if want_lo else 1) * Rp, dtype=torch.bfloat16, device=x.device)
    else:
        assert out.dtype == torch.bfloat16 and out.is_contiguous() and tuple(out.shape) == (Cc, (2 if want_lo else 1) * Rp)
    check(lib.ser_split_bf16_t(x.data_ptr(), R, Cc, x.stride(0), out.data_ptr(), out.data_ptr() + 2 * IL_GROUP if want_lo else None, Rp,
                               stream_ptr()), "ser_split_bf16_t")
    return out, Rp


_sig("ser_split_bf16_both", i32, vp, i32, i32, i64, vp, vp, vp, vp, i32, vp)


_sig("ser_split_bf16_both_multi", i32, vp, i32, i64, vp)
_sig("ser_split_bf16_both_colsum", i32, vp, i32, i32, i64, vp, vp, vp, vp, i32, vp, vp)


def split_bf16_both(x, straight_lo=True, t_lo=True, pad=64, colpart=False):
    """fp32 [R, C] contiguous-row device tensor (C % 32 == 0) -> (planes of x, planes of x^T, Rp) in one pass: each either ONE
    interleaved bf16 tensor ([R, 2 C] / [C, 2 Rp]) or, with its `lo` flag off, the hi plane alone ([R, C] / [C, Rp])."""
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1 and x.shape[1] % IL_GROUP == 0
    R, Cc = x.shape
    Rp = (R + pad - 1) // pad * pad
    s = torch.empty(R, (2 if straight_lo else 1) * Cc, dtype=torch.bfloat16, device=x.device)
    t = torch.empty(Cc, (2 if t_lo else 1) * Rp, dtype=torch.bfloat16, device=x.device)
    if colpart:           # + column sums per block of 32 rows: [Rp / 32, C] (finish with ser_colsum)
        part = torch.empty(Rp // 32, Cc, dtype=torch.float32, device=x.device)
        check(lib.ser_split_bf16_both_colsum(x.data_ptr(), R, Cc, x.stride(0), s.data_ptr(), s.data_ptr() + 2 * IL_GROUP if straight_lo else None,
                                             t.data_ptr(), t.data_ptr() + 2 * IL_GROUP if t_lo else None, Rp, part.data_ptr(), stream_ptr()),
              "ser_split_bf16_both_colsum")
        return s, t, Rp, part
    check(lib.ser_split_bf16_both(x.data_ptr(), R, Cc, x.stride(0), s.data_ptr(), s.data_ptr() + 2 * IL_GROUP if straight_lo else None,
                                  t.data_ptr(), t.data_ptr() + 2 * IL_GROUP if t_lo else None, Rp, stream_ptr()), "ser_split_bf16_both")
    return s, t, Rp


def il_ptrs(t):
    """(hi, lo) pointer pair of an interleaved tensor."""
    return t.data_ptr(), t.data_ptr() + 2 * IL_GROUP


def il_planes(t):
    """Interleaved tensor [..., 2 D] -> (hi, lo) views [..., D] (tests)."""
    v = t.reshape(t.shape[:-1] + (t.shape[-1] // (2 * IL_GROUP), 2, IL_GROUP))
    D = t.shape[-1] // 2
    return v[..., 0, :].reshape(t.shape[:-1] + (D,)), v[..., 1, :].reshape(t.shape[:-1] + (D,))


def gemm_bf16x3_il(a_il, w_il, bias=None, act=ACT_NONE, residual=None, out_f32=True, out_split=False):
    """Three-product NT GEMM on interleaved planes: a_il [M, 2K], w_il [N, 2K] -> (c fp32 [M,N] or None, c_il [M,2N] or None)."""
    M, K = a_il.shape[0], a_il.shape[1] // 2
    N = w_il.shape[0]
    dev = a_il.device
    c = torch.empty(M, N, dtype=torch.float32, device=dev) if out_f32 else None
    ci = torch.empty(M, 2 * N, dtype=torch.bfloat16, device=dev) if out_split else None
    ah, al = il_ptrs(a_il)
    wh, wl = il_ptrs(w_il)
    ch, cl = il_ptrs(ci) if ci is not None else (None, None)
    check(lib.ser_gemm_bf16_nt(ah, al, K, wh, wl, K, M, N, K, ptr(bias), act, ptr(residual), N, ptr(c), ch, cl, N, stream_ptr()),
          "ser_gemm_bf16_nt")
    return c, ci


def gemm_bf16_nt(a_hi, a_lo, w_hi, w_lo, bias=None, act=ACT_NONE, residual=None, out_f32=True, out_split=False):
    M, K = a_hi.shape
    N = w_hi.shape[0]
    dev = a_hi.device
    c = torch.empty(M, N, dtype=torch.float32, device=dev) if out_f32 else None
    ch = torch.empty(M, N, dtype=torch.bfloat16, device=dev) if out_split else None
    cl = torch.empty(M, N, dtype=torch.bfloat16, device=dev) if out_split and a_lo is not None else None
    check(lib.ser_gemm_bf16_nt(ptr(a_hi), ptr(a_lo), K, ptr(w_hi), ptr(w_lo), K, M, N, K, ptr(bias), act,
                               ptr(residual), N, ptr(c), ptr(ch), ptr(cl), N, stream_ptr()), "ser_gemm_bf16_nt")
    return c, ch, cl


def layernorm(x, gamma, beta, eps, x2=None, out_split=False, want_lo=True):
    rows, D = x.shape
    y = torch.empty_like(x)
    yh = torch.empty(rows, D, dtype=torch.bfloat16, device=x.device) if out_split else None
    yl = torch.empty(rows, D, dtype=torch.bfloat16, device=x.device) if out_split and want_lo else None
    check(lib.ser_layernorm(ptr(x), ptr(x2), ptr(gamma), ptr(beta), eps, rows, D, ptr(y), ptr(yh), ptr(yl),
                            stream_ptr()), "ser_layernorm")
    return y, yh, yl


def self_attention(qkv_hi, qkv_lo, key_mask, B, S, heads):
    H = heads * 64
    ch = torch.empty(B * S, H, dtype=torch.bfloat16, device=qkv_hi.device)
    cl = torch.empty(B * S, H, dtype=torch.bfloat16, device=qkv_hi.device) if qkv_lo is not None else None
    check(lib.ser_self_attention(ptr(qkv_hi), ptr(qkv_lo), ptr(key_mask), B, S, heads, ptr(ch), ptr(cl), stream_ptr()),
          "ser_self_attention")
    return ch, cl

# ---- trainable head (fp32) -------------------------------------------------------------------------
_sig("ser_gemm_f32", i32, vp, i64, i64, vp, i64, i64, i32, i32, i32, vp, i32, vp, i32, vp, i32, i32, vp)
_sig("ser_gemm_f32_np", i32, vp, i64, i64, vp, i64, i64, i32, i32, i32, vp, i32, vp, i32, vp, i32, i32, i32, vp)
_sig("ser_linear_fwd", i32, vp, vp, vp, i32, vp, i32, vp, i32, i32, i32, vp)
_sig("ser_linear_wgrad_group_workspace_bytes", sz, C.POINTER(i32), i32)
_sig("ser_linear_wgrad_group", i32, C.POINTER(vp), C.POINTER(i32), i32, i32, vp, sz, vp)
_sig("ser_linear_wgrad_batch", i32, C.POINTER(vp), C.POINTER(i32), i32, i32, i32, vp)
_sig("ser_linear_fwd_ln2", i32, vp, vp, vp, i32, vp, vp, vp, vp, f32, vp, vp, vp, vp, i32, i32, i32, vp)
_sig("ser_linear_dgrad", i32, vp, vp, vp, vp, i32, i32, i32, i32, vp)
_sig("ser_set_head_backward_products", i32, i32)
_sig("ser_get_head_backward_products", i32)
_sig("ser_resample_out_len", i32, i32, i32, i32)
_sig("ser_resample", i32, vp, i32, i32, i32, i32, i32, f32, vp, vp)
_sig("ser_add_noise_snr", i32, vp, i32, i32, vp, C.c_ulonglong, vp, vp, vp)
_sig("ser_dropout", i32, vp, C.c_longlong, vp, C.c_uint, f32, vp, vp)
_sig("ser_adamw_multi", i32, vp, vp, vp, vp, i32, vp, f32, f32, f32, vp)
_sig("ser_adamw_multi_gated", i32, vp, vp, vp, vp, vp, i32, vp, f32, f32, f32, vp)
_sig("ser_gemm_tile_hint", i32, C.c_longlong, i32, i32, i32)
_sig("ser_gemm_tile_hint_mode", i32, C.c_longlong, i32, i32, i32, i32)
_sig("ser_debug_set_gemm_bm", i32, i32)
_sig("ser_stack_supported", i32, i32, i32, i32)
_sig("ser_stack_scratch_bytes", sz, i32)
_sig("ser_stack_fwd", i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, vp, vp, C.c_uint, f32, vp)
_sig("ser_stack_bwd", i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, C.c_uint, f32, vp, vp)
_sig("ser_stack_ln_param_bwd", i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp)
_sig("ser_linear_wgrad_pair", i32, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, i32, i32, i32, i32, vp)
_sig("ser_layernorm2_fwd", i32, vp, vp, vp, vp, vp, f32, i32, i32, vp, vp, vp, vp)
_sig("ser_layernorm2_bwd", i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, i32, vp)
_sig("ser_linear_wgrad_workspace_bytes", sz, i32, i32, i32)
_sig("ser_linear_wgrad", i32, vp, vp, vp, vp, i32, i32, i32, i32, vp, sz, vp)
_sig("ser_layernorm_fwd", i32, vp, vp, vp, vp, f32, i32, i32, vp, vp, vp, vp, vp)
_sig("ser_layernorm_bwd_workspace_bytes", sz, i32, i32)
_sig("ser_layernorm_bwd", i32, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, i32, vp, vp)
_sig("ser_colsum", i32, vp, i32, i32, i32, vp, i32, vp)
_sig("ser_act_fwd", i32, vp, i32, i64, vp, vp)
_sig("ser_act_bwd", i32, vp, vp, i32, i64, vp, vp)
_sig("ser_act_drop_fwd", i32, vp, i32, i64, vp, vp, C.c_uint, f32, vp)
_sig("ser_act_drop_bwd", i32, vp, vp, i32, i64, vp, vp, C.c_uint, f32, vp)
_sig("ser_axpby", i32, vp, f32, f32, i64, vp, vp)
_sig("ser_scale_dev", i32, vp, vp, i64, vp)
_sig("ser_xattn_fwd", i32, vp, i32, vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, vp, vp, i32, vp, C.c_uint, f32, vp, vp)
_sig("ser_xattn_bwd", i32, vp, i32, vp, i32, vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, vp, vp, i32, vp, i32, vp, i32, vp, C.c_uint, f32, vp, vp)
_sig("ser_pool_fwd", i32, vp, vp, vp, i32, i32, i32, vp, vp, vp)
_sig("ser_pool_bwd", i32, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp)
_sig("ser_fusion_mix_fwd", i32, vp, vp, vp, vp, i32, i32, vp, vp)
_sig("ser_fusion_mix_bwd", i32, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp)
_sig("ser_train_loss", i32, vp, vp, vp, vp, vp, i32, i32, i32, f32, f32, f32, f32, f32, f32, f32, i32, vp, vp, vp, vp, vp, vp, vp)
_sig("ser_openmax", i32, vp, vp, vp, vp, vp, i32, i32, i32, f32, f32, vp, vp)
_sig("ser_adamw", i32, vp, vp, vp, vp, i64, vp, f32, f32, f32, f32, f32, vp)
