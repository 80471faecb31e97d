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


def test_conv1_1_wgrad_read_set_is_host_logic():
    """szn_conv1_1_wgrad_reads (what conv1_2's dgrad may leave unwritten under the constant-border hint) is plain host arithmetic:
    the fused 16-bit kernel reads the rows whose 3x3 windows meet the image and whole 32-pixel column segments around them; fp32 and
    shapes the fused kernel does not take read everything"""
    from zeroshotsemanticsegmentation_amd import _lib
    lib = _lib.load()
    rect = (ctypes.c_int * 4)()
    assert lib.szn_conv1_1_wgrad_reads(_lib.SZN_BF16, 8, 512, 512, 100, rect) == 1
    assert tuple(rect) == (98, 612, 96, 640)
    assert lib.szn_conv1_1_wgrad_reads(_lib.SZN_F16, 2, 300, 500, 100, rect) == 1
    assert tuple(rect) == (98, 400, 96, 608)                       # map 498 x 698; columns [98, 600) -> segments [96, 608)
    assert lib.szn_conv1_1_wgrad_reads(_lib.SZN_F32, 8, 512, 512, 100, rect) == 0
    assert tuple(rect) == (0, 710, 0, 710)
    assert lib.szn_conv1_1_wgrad_reads(_lib.SZN_BF16, 1, 64, 64, 1, rect) == 0
    assert tuple(rect) == (0, 64, 0, 64)                            # pad 1: the map is 64 x 64, every window meets the image


def test_every_knob_has_a_case():
    """no environment knob without a test case -- the library's table (szn_knob_count / szn_knob_name), the package's
    os.environ reads and the C sources' getenv calls against tests/test_gpu_knobs.py's groups"""
    import re
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_knobs import ELSEWHERE, GROUPS, PY_KNOBS
    from zeroshotsemanticsegmentation_amd import _lib as L
    lib = L.load()
    names = {lib.szn_knob_name(i).decode() for i in range(lib.szn_knob_count())}
    assert lib.szn_knob_name(lib.szn_knob_count()) is None and len(names) == lib.szn_knob_count()
    covered = set()
    for env in GROUPS.values():
        covered |= set(env)
    assert names <= covered, "library knobs without a test case: %s" % sorted(names - covered)
    # the Python side: every SZN_* variable the package reads is in PY_KNOBS (and through it in a group or in the test file named above)
    found = set()
    pkg = os.path.join(ROOT, "zeroshotsemanticsegmentation_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            found |= set(re.findall(r'environ[^\n]*?"(SZN_[A-Z0-9_]+)"', open(os.path.join(pkg, fn)).read()))
    assert found == PY_KNOBS, (sorted(found - PY_KNOBS), sorted(PY_KNOBS - found))
    assert PY_KNOBS - ELSEWHERE <= covered, sorted(PY_KNOBS - ELSEWHERE - covered)
    # and no C source reads the environment behind szn_knob's back (the ablation switches exist in `make ABLATE=1` builds only)
    src = os.path.join(pkg, "csrc")
    for fn in os.listdir(src):
        if fn.endswith((".hip", ".h")):
            for ln in open(os.path.join(src, fn)).read().splitlines():
                if "getenv(" in ln:
                    assert fn == "szn_elementwise.hip" and "getenv(name)" in ln or fn == "szn_common.h", (fn, ln.strip())


