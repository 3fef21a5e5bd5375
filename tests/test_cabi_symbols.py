"""CPU-only: the C-ABI library builds for gfx950, loads, and exports every entry point that
include/ser_hip.h declares (no compute call is made: there is no GPU here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "ser_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ser_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_hot_path():
    names = declared_functions()
    for must in ("ser_wav2vec2_forward", "ser_xlmr_forward", "ser_gemm_bf16_nt", "ser_self_attention", "ser_layernorm",
                 "ser_linear_fwd", "ser_linear_dgrad", "ser_linear_wgrad", "ser_xattn_fwd", "ser_xattn_bwd", "ser_pool_fwd",
                 "ser_pool_bwd", "ser_fusion_mix_fwd", "ser_fusion_mix_bwd", "ser_train_loss", "ser_openmax", "ser_adamw"):
        assert must in names, must


def test_library_exports_every_declared_symbol():
    import ser_amd  # noqa: F401
    from ser_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, f"declared in ser_hip.h but not exported: {missing}"
    assert lib.ser_abi_version() >= 1


def test_host_only_entry_points_work_without_gpu():
    import ser_amd  # noqa: F401
    from ser_amd import _lib as L
    cfg = L.W2vConfig()
    cfg.hidden, cfg.layers, cfg.heads, cfg.ffn, cfg.n_conv = 768, 12, 12, 3072, 7
    for i, (d, k, s) in enumerate(zip([512] * 7, [10, 3, 3, 3, 3, 2, 2], [5, 2, 2, 2, 2, 2, 2])):
        cfg.conv_dim[i], cfg.conv_kernel[i], cfg.conv_stride[i] = d, k, s
    cfg.pos_kernel, cfg.pos_groups, cfg.eps = 128, 16, 1e-5
    # frame counts of SURVEY section 8(a): 1 s -> 49, 3 s -> 149, 4 s -> 199, 10 s -> 499
    for sec, frames in ((1, 49), (3, 149), (4, 199), (10, 499)):
        assert L.lib.ser_wav2vec2_out_len(ctypes.byref(cfg), 16000 * sec) == frames
    assert L.lib.ser_wav2vec2_out_len(ctypes.byref(cfg), 100) <= 0            # shorter than the receptive field
    nb = L.lib.ser_wav2vec2_workspace_bytes(ctypes.byref(cfg), 16, 64000, L.PREC_BF16)
    nx = L.lib.ser_wav2vec2_workspace_bytes(ctypes.byref(cfg), 16, 64000, L.PREC_BF16X3)
    assert 2e8 < nb < nx < 4e9
    assert L.lib.ser_linear_wgrad_workspace_bytes(16, 512, 512) == 0
    assert L.lib.ser_linear_wgrad_workspace_bytes(3184, 256, 768) > 13 * 256 * 768 * 4


def test_product_refuses_to_run_without_gpu_tensors():
    """No silent CPU fallback: CPU tensors are rejected before any launch."""
    import pytest
    import torch
    import ser_amd  # noqa: F401
    from ser_amd import _ops as O
    with pytest.raises(AssertionError):
        O.linear_fwd(torch.randn(4, 16), torch.randn(8, 16))
