"""Every environment knob that survives in the tree, under test.

The library reads an environment variable only through szn_knob(), which refuses names that are not in its table (szn_knob_count /
szn_knob_name enumerate it); the Python side's switches are listed in PY_KNOBS below and checked against a grep of the package.  Each
GROUP sets non-default values for a few knobs that do not interact and runs two bf16 training steps (B = 3, 512 x 512, E = 20, K = 33)
in a child process (the switches are read once per process); loss, class map and every layer's gradient must agree with the default
run -- different kernels / summation orders / hints, same step.  A knob that appears in no group fails tests/test_abi.py::test_every_knob_has_a_case (CPU)."""
import os
import re
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

HUGE = "1000000000"
GROUPS = {
    "default": {},
    # dispatch thresholds: every specialised kernel family off -> the generic tile kernels run the whole step
    "generic kernels only": {"SZN_REGW_MINTILES": HUGE, "SZN_WIDE_MINTILES": HUGE, "SZN_WGT_MINTILES": HUGE, "SZN_WGW_MINTILES": HUGE},
    # the round 1-3 256-wide kernels instead of conv_igemm_8ph, without the rows form
    "wide kernels of rounds 1-3": {"SZN_WIDE_8PH": "0", "SZN_WIDE_ROWS": "0"},
    "tap-major K order": {"SZN_8PH_KORD": "0"},
    "LDS-staged epilogues": {"SZN_WIDE_8PH": "0", "SZN_WIDE_DIRECT": "0", "SZN_IGEMM_DIRECT": "0"},
    # full maps with the tile-skipping hints (rounds 3-4), and fully dense
    "hints instead of the band": {"SZN_BAND_CROP": "0"},
    "dense": {"SZN_BAND_CROP": "0", "SZN_CONST_BORDER": "0", "SZN_DGRAD_BORDER": "0", "SZN_WGT_CB": "0"},
    # fc6's weight gradient + Adam: the two-block form forced on / off, no stagger, launch-order tiles
    "fc6 wgrad: half tiles": {"SZN_WGW_HALF": "1", "SZN_WGH_STAGGER": "0", "SZN_WGW_XCD": "0"},
    "fc6 wgrad: full tiles": {"SZN_WGW_HALF": "0", "SZN_WGW_STAGGER": "0"},
    # engine switches
    "atomics + second stream": {"SZN_DETERMINISTIC": "0", "SZN_WGRAD_STREAM": "1"},
    "separate optimizer pass": {"SZN_FUSED_ADAM": "0", "SZN_WGRAD_STREAM": "0"},
    "early optimizer pass": {"SZN_EARLY_ADAM": "1"},
    "fc7 fused too, CU-masked fc6": {"SZN_FUSED_ADAM_LAYERS": "fc6,fc7", "SZN_FC6_CUMASK": "128:low"},
}
# switches of the Python side (grep of the package below) and where each is exercised when not in GROUPS
PY_KNOBS = {"SZN_BAND_CROP", "SZN_CONST_BORDER", "SZN_DGRAD_BORDER", "SZN_DETERMINISTIC", "SZN_WGRAD_STREAM", "SZN_FUSED_ADAM",
            "SZN_FUSED_ADAM_LAYERS", "SZN_EARLY_ADAM", "SZN_FC6_CUMASK",
            # data-parallel configuration (tests/test_gpu_ddp_single_gpu.py, test_gpu_rccl_world1.py, test_gpu_wire.py, test_ddp_gloo.py)
            "SZN_GRAD_COMM", "SZN_WIRE_DIRECT", "SZN_SHARDED_OPT", "SZN_RCCL_HIPRI", "SZN_FORCE_COMM", "SZN_RESERVED_CUS",
            # harness / trainer plumbing (tests/test_gpu_bench_contract.py, test_gpu_train_cli.py, tools/ab_lib.sh)
            "SZN_TEST_ONE_GPU", "SZN_LIB_PATH", "SZN_VERBOSE_VAL", "SZN_KEEP_GRADS"}
ELSEWHERE = {"SZN_GRAD_COMM", "SZN_WIRE_DIRECT", "SZN_SHARDED_OPT", "SZN_RCCL_HIPRI", "SZN_FORCE_COMM", "SZN_RESERVED_CUS", "SZN_TEST_ONE_GPU",
             "SZN_LIB_PATH", "SZN_VERBOSE_VAL", "SZN_KEEP_GRADS"}


def _child(out):
    from zeroshotsemanticsegmentation_amd import _lib as L
    from zeroshotsemanticsegmentation_amd import engine, models, synth
    dev = torch.device("cuda", 0)
    E, K, H, B = 20, 33, 512, 3          # (three images: conv3_x reaches the 240 tiles from which the 256-wide kernels take a layer)
    m = models.FCN32s(E)
    m.load_synthetic(1337, device=dev)
    m.eval()
    prec = torch.float32 if os.environ.get("KNOBTEST_FP32") == "1" else torch.bfloat16      # (test-only switch: the envelope run, see default_run)
    ts = engine.TrainStep(m, synth.make_embeddings(K, E), optimizer="adam", lr=1e-4, precision=prec, fused_head=True, keep_grads=True)
    x = torch.from_numpy(synth.make_images(B, H, H, seed=5)).to(dev)
    t = torch.from_numpy(synth.make_labels(B, H, H, K, seed=6)).to(dev)
    kernels = set()
    orig = L.call

    def spy(name, *a):
        orig(name, *a)
        kernels.add(L.last_kernel())
        kernels.add(L.prev_kernel())
    engine.L.call = models.L.call = spy
    try:
        loss, pred = ts.step(x, t)
        loss = float(loss)                            # (the step returns a view of its loss buffer: read it before the next step)
        loss2, _ = ts.step(x, t)                      # (the second step runs on the weights the first one's optimizer wrote)
    finally:
        engine.L.call = models.L.call = orig
    torch.cuda.synchronize()
    grads = {}
    for n in ts.layers:
        o, cnt = ts.woff[n]
        grads[n] = ts.flat_gw[o:o + cnt].float().cpu()
    torch.save({"loss": float(loss), "loss2": float(loss2), "pred": pred.cpu(), "grads": grads, "kernels": sorted(kernels),
                "w": ts.flat_w[::997].float().cpu()}, out)


def _run(tag, fast_tmp):
    out = os.path.join(fast_tmp, re.sub(r"\W+", "_", tag) + ".pt")
    env = {k: v for k, v in os.environ.items() if not k.startswith("SZN_") or k == "SZN_LIB_PATH"}
    env.update({"KNOBTEST_FP32": "1"} if tag == "fp32" else GROUPS[tag])
    p = subprocess.run([sys.executable, os.path.abspath(__file__), out], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, (tag, p.stderr[-3000:])
    return torch.load(out)


@pytest.fixture(scope="module")
def default_run(tmp_path_factory):
    import shutil
    import tempfile
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    d = tempfile.mkdtemp(prefix="szn_knobs_", dir=base)
    try:
        ref = _run("default", d)
        # the envelope of THIS configuration: the same step in fp32.  A bf16 pass whose forward kernels sum in another order ends up with activations
        # that differ from the default pass by independent bf16 roundings -- the same kind and size of difference as bf16 against fp32 -- so its
        # gradients may differ from the default's by about as much as the default's differ from the fp32 step's (ReLU-gate / pooling-winner flips)
        f32 = _run("fp32", d)
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from test_gpu_headline_pin import rel_l2
        ref["envelope"] = {n: rel_l2(ref["grads"][n], f32["grads"][n]) for n in ref["grads"]}
        print("envelope (bf16 default vs fp32, relative L2 per layer): " + "  ".join("%s %.2e" % kv for kv in ref["envelope"].items()))
        yield d, ref
    finally:
        shutil.rmtree(d, ignore_errors=True)


FORWARD_CHANGES = {"generic kernels only", "wide kernels of rounds 1-3", "tap-major K order", "LDS-staged epilogues"}
EXPECT = {   # group -> kernels that must (+) / must not (-) have run
    "default": ("+conv_igemm_8ph", "+conv3x3_regw", "+wgrad_taps_reduce", "+band_remap_kernel", "+conv_wgrad_half_adam"),
    # (fc6's split-K forward and its dgrad GEMM on the forward layout ask for the 256-wide kernels by themselves, whatever the thresholds say)
    # (... and SZN_WGT_MINTILES is "tiles per block": a huge value leaves the all-taps kernel one pixel split, it does not switch it off)
    "generic kernels only": ("-conv3x3_regw", "-conv3x3_wide_rows", "-conv_wgrad_wide_adam", "-conv_wgrad_half_adam", "+conv_igemm_v2",
                             "+conv_wgrad_v2"),
    "wide kernels of rounds 1-3": ("-conv_igemm_8ph", "+conv_igemm_wide"),
    "tap-major K order": ("+conv_igemm_8ph",),
    "LDS-staged epilogues": ("-conv_igemm_8ph",),
    "hints instead of the band": ("-band_remap_kernel",),
    "dense": ("-band_remap_kernel", "-border"),
    "fc6 wgrad: half tiles": ("+conv_wgrad_half_adam",),
    "fc6 wgrad: full tiles": ("+conv_wgrad_wide_adam", "-conv_wgrad_half_adam"),
    "atomics + second stream": ("-colsum_reduce_kernel",),
    "separate optimizer pass": ("-conv_wgrad_wide_adam", "-conv_wgrad_half_adam", "+adam_kernel"),
    "early optimizer pass": ("-conv_wgrad_wide_adam", "-conv_wgrad_half_adam", "+adam_kernel"),
    "fc7 fused too, CU-masked fc6": ("+conv_wgrad_half_adam",),
}


@pytest.mark.parametrize("tag", [g for g in GROUPS if g != "default"])
def test_step_under_non_default_knobs_equals_the_default_step(tag, default_run):
    d, ref = default_run
    got = _run(tag, d)
    print(tag, "kernels:", " ".join(got["kernels"]))
    for rule in EXPECT[tag]:
        hit = any(rule[1:] in k for k in got["kernels"])
        assert hit == (rule[0] == "+"), (tag, rule, got["kernels"])
    for rule in EXPECT["default"]:
        assert any(rule[1:] in k for k in ref["kernels"]), (rule, ref["kernels"])
    # same step: 16-bit activations differ in their last bit where another kernel / summation order produced them, which flips ReLU gates and
    # pooling winners downstream; a flip moves whole gradient elements and the flips of all layers behind a tensor add up towards the input
    # (tests/test_gpu_headline_pin.py measures the mechanism -- 3e-3 at score_fr .. 0.39 at conv1_1 between the fp32 and the bf16 pass -- and pins
    # the backward kernels themselves to 7e-3 against the oracle on ONE forward state).  Two bf16 passes through different kernels differ the same
    # way (measured here: conv1_1 0.2 - 0.64, everything from conv3_1 on below 0.1), so the bound per layer is that envelope x 2 and the direction
    # must hold: loss 1e-3, class map 0.995, gradients 2 x MEASURED relative L2 with cosine >= 0.7
    assert abs(got["loss"] - ref["loss"]) < 1e-3 * abs(ref["loss"]), (tag, got["loss"], ref["loss"])
    assert abs(got["loss2"] - ref["loss2"]) < 1e-3 * abs(ref["loss2"]), (tag, got["loss2"], ref["loss2"])
    assert got["loss2"] < got["loss"]
    assert float((got["pred"] == ref["pred"]).float().mean()) > 0.995
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_headline_pin import MEASURED, cosine, rel_l2
    rows = [(n, rel_l2(got["grads"][n], ref["grads"][n]), cosine(got["grads"][n], ref["grads"][n])) for n in ref["grads"]]
    print("%s: relative L2 / cosine of every layer's weight gradient against the default step: %s"
          % (tag, "  ".join("%s %.2e/%.4f" % r for r in rows)))
    # groups that leave the FORWARD kernels alone reproduce the forward state bit for bit (loss equal): their gradients are held to the headline
    # envelope; groups that change a forward kernel's summation order are held to THIS configuration's own bf16-against-fp32 envelope (default_run:
    # measured 0.03 at score_fr .. 0.6 at conv1_1 at B = 3, E = 20, eval mode -- the flips average over fewer pixels than in the headline test) x 2,
    # plus the direction; the kernels themselves are compared bit for bit in tests/test_gpu_conv.py
    same_forward = got["loss"] == ref["loss"]
    assert same_forward == (tag not in FORWARD_CHANGES), (tag, got["loss"], ref["loss"])
    env = ref["envelope"]
    bad = [r for r in rows if not ((r[1] <= 2.0 * MEASURED[r[0]] if same_forward else r[1] <= min(2.0 * env[r[0]], 0.9)) and r[2] >= 0.7)]
    assert not bad, (tag, bad, env)
    assert float((got["w"] - ref["w"]).abs().max()) < 4.5e-4          # two Adam steps at lr 1e-4: a tiny gradient whose sign flips moves 2 lr per step


if __name__ == "__main__":
    _child(sys.argv[1])
