"""CPU: the C-ABI shared library loads and exports every symbol include/szn.h declares (no compute calls)."""
import ctypes
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "szn.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(szn_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_expected_surface():
    syms = header_symbols()
    for must in ("szn_conv2d_fwd", "szn_conv2d_dgrad", "szn_conv2d_wgrad", "szn_gemm_proj_fwd", "szn_maxpool2x2_ceil_fwd",
                 "szn_bilinear_up32_crop_fwd", "szn_deconv64s32_wgrad", "szn_cosine_loss_fwd", "szn_cosine_loss_bwd",
                 "szn_mse_loss_fwd", "szn_ce2d_fwd", "szn_embed_argmax", "szn_confusion_hist", "szn_fused_head",
                 "szn_adam_step", "szn_sgd_momentum_step", "szn_version", "szn_device_info", "szn_last_error"):
        assert must in syms


def test_library_exports_every_header_symbol():
    import __graft_entry__ as g
    from zeroshotsemanticsegmentation_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        g.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = header_symbols()
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    # the ctypes binding covers exactly the header
    assert sorted(_lib.SIGNATURES) == syms
    loaded = _lib.load()
    assert loaded.szn_version() >= 100
    assert isinstance(loaded.szn_last_error(), bytes)


def test_product_path_has_no_cpu_fallback():
    import pytest
    import torch
    from zeroshotsemanticsegmentation_amd import _lib, utils
    with pytest.raises(_lib.SznError):
        utils.cosine_loss(torch.zeros(1, 4, 2, 2), torch.zeros(1, 2, 2, dtype=torch.int64), torch.zeros(3, 4))
    with pytest.raises(_lib.SznError):
        utils.infer_lbl(torch.zeros(1, 4, 2, 2), torch.zeros(3, 4))
    # nothing under the package imports the oracle
    pkg = os.path.join(ROOT, "zeroshotsemanticsegmentation_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            assert "oracle" not in open(os.path.join(pkg, fn)).read().replace("the oracle", "").replace("CPU oracle", ""), fn
