"""CPU: host-side logic that mirrors the reference's train.py / configs.py / utils.py metrics (no GPU needed)."""
import json
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zeroshotsemanticsegmentation_amd import configs, synth, train, utils  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")


def test_configs_equal_reference():
    ref = json.load(open(os.path.join(G, "configs.json")))
    assert {str(k): v for k, v in configs.configurations.items()} == ref
    assert sorted(configs.configurations) == [1, 2, 4, 14, 15, 16, 17, 18, 19]


def args(*argv):
    return train.build_parser().parse_args(list(argv))


def test_update_and_validate_cfg():
    cfg = train.update_cfg_with_args(configs.configurations[18], args('-c', '18', '-e', '300', '-lr', '0.001', '-tu', '3,4'))
    assert cfg['embed_dim'] == 300 and cfg['fcn_lr'] == 0.001 and cfg['train_unseen'] == [3, 4]
    assert cfg['one_hot_embed'] is None and cfg['forced_unseen'] is None and cfg['load_fcn_path'] is None
    assert configs.configurations[18]['embed_dim'] == 20          # the table itself is not mutated
    train.validate_cfg(cfg)
    # reference quirks kept: -ve 0 cannot override (truthiness), -se is parsed but never applied
    cfg = train.update_cfg_with_args(configs.configurations[18], args('-c', '18', '-ve', '0', '-se', '3'))
    assert cfg['fcn_epochs'] == 59 and cfg['seenmask_epochs'] == 10
    with pytest.raises(Exception):      # test mode needs -r
        train.validate_cfg(train.update_cfg_with_args(configs.configurations[19], args('-c', '19')))
    with pytest.raises(Exception):      # seenmask phase without train_unseen
        c = dict(train.update_cfg_with_args(configs.configurations[4], args('-c', '4')), seenmask_epochs=2)
        train.validate_cfg(c)
    with pytest.raises(Exception):      # cos loss without an embedding space
        c = dict(train.update_cfg_with_args(configs.configurations[1], args('-c', '1')), fcn_loss='cos')
        train.validate_cfg(c)
    with pytest.raises(SystemExit):     # -e choices as in the reference
        args('-e', '33')


def test_log_dir_name(tmp_path):
    import datetime
    cfg = train.update_cfg_with_args(configs.configurations[14], args('-c', '14'))
    now = datetime.datetime(2018, 4, 21, 16, 37, 51)
    d = train.get_log_dir('8_2_10', 14, cfg, str(tmp_path), now=now)
    # the reference's own run name for cfg 14 (configs.py:83, load_fcn_path of cfg 15)
    assert os.path.basename(d) == configs.configurations[15]['load_fcn_path']
    assert os.path.isdir(d)


def test_get_parameters_groups():
    from zeroshotsemanticsegmentation_amd import models
    m = models.FCN32s(n_class=20)
    ws = list(train.get_parameters(m, bias=False))
    bs = list(train.get_parameters(m, bias=True))
    sm = list(train.get_parameters(m, seenmask=True))
    assert len(ws) == 16 and len(bs) == 16 and len(sm) == 3
    assert sum(p.numel() for p in ws) + sum(p.numel() for p in bs) == 134342484      # SURVEY A2, E = 20
    ids = {id(p) for p in ws + bs}
    assert id(m.upscore.weight) not in ids and id(m.seenmask_score.weight) not in ids
    assert [tuple(p.shape) for p in sm] == [(2, 4096, 1, 1), (2,), (2, 2, 64, 64)]
    m.extra = nn.Linear(2, 2)
    with pytest.raises(ValueError):
        list(train.get_parameters(m))
    del m.extra
    train.freeze_for_seenmask(m)
    assert [n for n, p in m.named_parameters() if p.requires_grad] == ['seenmask_score.weight', 'seenmask_score.bias',
                                                                      'seenmask_upscore.weight']
    # state_dict surface of the reference
    keys = list(m.state_dict().keys())
    assert keys[:2] == ['conv1_1.weight', 'conv1_1.bias'] and 'upscore.weight' in keys and len(keys) == 36
    assert tuple(m.upscore.weight.shape) == (20, 20, 64, 64) and m.upscore.bias is None
    assert torch.equal(m.upscore.weight[3, 3], m.seenmask_upscore.weight[1, 1]) and float(m.upscore.weight[0, 1].abs().sum()) == 0
    with pytest.raises(Exception):
        m(torch.zeros(1, 3, 4, 4), mode='nope')


def test_copy_params_from_vgg16():
    from zeroshotsemanticsegmentation_amd import models
    vgg = models.VGG16(pretrained=False)
    m = models.FCN32s(n_class=5)
    m.copy_params_from_vgg16(vgg)
    assert torch.equal(m.conv3_2.weight, vgg.features[12].weight)
    assert torch.equal(m.fc6.weight.reshape(4096, -1), vgg.classifier[0].weight)
    assert torch.equal(m.fc7.bias, vgg.classifier[3].bias)
    with pytest.raises(IOError):
        models.VGG16(pretrained=True, data_dir='/nonexistent')


def test_metrics_host_path_against_golden():
    g = np.load(os.path.join(G, "g6_metrics.npz"))
    lt, lp = list(g["lt"]), list(g["lp"])
    np.testing.assert_allclose(utils.label_accuracy_score(lt, lp, 33), g["metrics"], rtol=1e-12, equal_nan=True)
    np.testing.assert_allclose(np.array(utils.label_accuracy_score(lt, lp, 33, unseen=[16, 18])), g["metrics3"], rtol=1e-12,
                               equal_nan=True)
    lt2, lp2 = [g["lt_adv0"], g["lt_adv1"]], [g["lp_adv0"], g["lp_adv1"]]
    np.testing.assert_allclose(utils.label_accuracy_score(lt2, lp2, 33), g["metrics_adv"], rtol=1e-12, equal_nan=True)


def test_synth_is_deterministic_and_shaped():
    a, b = synth.make_params(20), synth.make_params(20)
    assert all(np.array_equal(a[k], b[k]) for k in a) and a["fc6.weight"].shape == (4096, 512, 7, 7)
    x = synth.make_images(2, 8, 9)
    assert x.shape == (2, 3, 8, 9) and x.dtype == np.float32 and -123.0 <= x.min() and x.max() <= 151.0
    lbl = synth.make_labels(2, 70, 40, 21)
    assert lbl.min() == -1 and lbl.max() <= 20 and 0.02 < (lbl == -1).mean() < 0.09
    e = synth.make_embeddings(59, 300)
    n = np.linalg.norm(e, axis=1)
    assert e.shape == (59, 300) and abs(n.max() - 1.0) < 1e-6 and n.min() > 0.62
    assert synth.unseen_bits([0, 12, 63]) == (1 | (1 << 12) | (1 << 63))


def test_class_sets_beyond_64_classes():
    """szn_class_set packing (bit k % 64 of word k // 64) and the oracle's word-array class sets at K = 150 against a float64
    numpy restatement of utils.py:159-205 / 104-154 (the oracle is what the -m gpu tests of the *_k entry points compare with)"""
    from oracle import szn_oracle as O
    from zeroshotsemanticsegmentation_amd import _lib as L
    cs = L.class_set([0, 63, 64, 130, 255])._obj
    assert list(cs.w) == [1 | (1 << 63), 1, 1 << 2, 1 << 63]
    assert L.class_set([]) is None and L.class_set(None) is None
    with pytest.raises(L.SznError):
        L.class_set([256])
    K, E, H, W = 150, 20, 16, 16
    unseen = [3, 63, 64, 70, 127, 128, 149]
    emb = synth.make_embeddings(K, E, seed=3)
    score = synth.uniform(77, (2, E, H, W), -1, 1)
    sm = synth.uniform(78, (2, 2, H, W), -1, 1)
    target = synth.make_labels(2, H, W, K, seed=79, block=2)
    s = score.transpose(0, 2, 3, 1).reshape(-1, E).astype(np.float64)
    e = emb.astype(np.float64)
    sim = (s @ e.T) / (np.linalg.norm(s, axis=1, keepdims=True) * np.linalg.norm(e, axis=1)[None])
    top = np.sort(sim, axis=1)
    clear = ((top[:, -1] - top[:, -2]) > 1e-5).reshape(2, H, W)
    assert np.array_equal(O.infer_lbl(score, emb)[clear], sim.argmax(1).reshape(2, H, W)[clear])
    un = np.zeros(K, bool)
    un[unseen] = True
    ps = np.where(un[None], 0.0, sim).argmax(1).reshape(2, H, W)       # seen-only matrix: unseen rows zeroed, still competing
    pu = np.where(un[None], sim, 0.0).argmax(1).reshape(2, H, W)
    # (top-2 margins of the two masked problems are not the full problem's: compare where both variants are far from ties)
    def far(m):
        t = np.sort(m, axis=1)
        return ((t[:, -1] - t[:, -2]) > 1e-5).reshape(2, H, W)
    ok = far(np.where(un[None], 0.0, sim)) & far(np.where(un[None], sim, 0.0))
    want = np.where(sm[:, 1] > sm[:, 0], ps, pu)                       # utils.py:197-198: mask argmax 0 -> unseen
    assert np.array_equal(O.infer_lbl_szn(score, sm, emb, unseen)[ok], want[ok])
    want = np.where(np.isin(target, unseen), pu, ps)                   # utils.py:190-191
    assert np.array_equal(O.infer_lbl_forced_unseen(score, target, emb, unseen)[ok], want[ok])
    lp = synth.make_labels(2, H, W, K, seed=80, block=1, ignore_frac=0.0)
    hist = O.confusion_hist(target, lp, K, unseen=unseen)
    v = target >= 0
    full = np.bincount(K * target[v] + lp[v], minlength=K * K).reshape(K, K)
    vu = v & np.isin(target, unseen)
    assert np.array_equal(hist[0], full)
    assert np.array_equal(hist[2], np.bincount(K * target[vu] + lp[vu], minlength=K * K).reshape(K, K))
    assert np.array_equal(hist[1], full - hist[2])
    assert np.isin(target[v], [k for k in unseen if k >= 64]).any()


def test_seenmask_binary_target_rule():
    from zeroshotsemanticsegmentation_amd import trainer_seenmask
    g = np.load(os.path.join(G, "g8_seenmask_step.npz"))

    class DS(object):
        class_names = ['c%d' % i for i in range(33)]

    class Loader(object):
        dataset = DS()

    tr = trainer_seenmask.Trainer.__new__(trainer_seenmask.Trainer)
    tr.n_class, tr.device = 33, torch.device("cpu")
    seen = [x for x in range(33) if x not in list(g["unseen"])]
    tr._seen_lut = torch.zeros(34, dtype=torch.int64)
    tr._seen_lut[torch.tensor(seen)] = 1
    got = tr.binary_target(torch.from_numpy(g["target"]))
    assert np.array_equal(got.numpy(), g["bin_target"])          # -1 -> 0 ("unseen"), not ignored


def test_packaged_embeddings_equal_golden_capture():
    """the K x E matrices shipped as package data are the byte-identical re-saves captured from the reference (G9)"""
    import glob
    pk = os.path.join(ROOT, "zeroshotsemanticsegmentation_amd", "data")
    files = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "embeddings_*.npy")))
    assert len(files) == 18        # 2 datasets x every width of the reference CLI (train.py:31)
    for f in files:
        assert open(f, "rb").read() == open(os.path.join(pk, os.path.basename(f)), "rb").read()


def test_vgg16_pretrained_file_round_trip(tmp_path):
    from zeroshotsemanticsegmentation_amd import models
    """A6: `VGG16(pretrained=True, data_dir)` reads `<data_dir>/models/vgg16_from_caffe.pth` (the file the reference downloads,
    models.py:195-210).  The real caffe weights are not available here; a state dict in that file's layout (torchvision VGG16
    keys: features.N.weight/bias, classifier.{0,3,6}.weight/bias with the (4096, 512*7*7) Linear of fc6) must load and land
    in the FCN32s layers exactly as the reference's copy loop places it (models.py:183-193)."""
    g = torch.Generator().manual_seed(5)
    src = models.VGG16(pretrained=False)
    sd = {k: torch.randn(v.shape, generator=g) * 0.01 for k, v in src.state_dict().items()}
    assert sd["classifier.0.weight"].shape == (4096, 512 * 7 * 7) and "features.28.bias" in sd
    (tmp_path / "models").mkdir()
    torch.save(sd, str(tmp_path / "models" / "vgg16_from_caffe.pth"))
    vgg = models.VGG16(pretrained=True, data_dir=str(tmp_path))
    m = models.FCN32s(n_class=20)
    m.copy_params_from_vgg16(vgg)
    conv_keys = [k for k in sd if k.startswith("features.") and k.endswith(".weight")]
    ours = [n for n, _, _, _ in synth.CONV_LAYERS[:13]]
    assert len(conv_keys) == 13
    for k, n in zip(conv_keys, ours):
        assert torch.equal(getattr(m, n).weight, sd[k]) and torch.equal(getattr(m, n).bias, sd[k[:-6] + "bias"]), n
    # fc6 = the first Linear viewed as (4096, 512, 7, 7): channel-major flattening of the 7x7 window
    assert torch.equal(m.fc6.weight, sd["classifier.0.weight"].view(4096, 512, 7, 7))
    assert torch.equal(m.fc7.weight, sd["classifier.3.weight"].view(4096, 4096, 1, 1))
    assert torch.equal(m.fc6.bias, sd["classifier.0.bias"]) and torch.equal(m.fc7.bias, sd["classifier.3.bias"])
    # the kernels read the weights in OHWI order: the copy must keep channels_last storage
    assert m.fc6.weight.is_contiguous(memory_format=torch.channels_last)


def test_band_cut_tables_are_consistent():
    """models._band_cut / _BandPlan (round 5: the constant band removed from the conv blocks of the pad-100 network, models.py:43): for every
    image size the four index maps are mutually consistent -- crop keeps exactly the rows crop_bwd maps back, every pooled row is either kept
    or a copy of a kept pure row, uncrop_bwd is the transpose of uncrop -- and the cuts respect the purity margins they were derived from."""
    import numpy as np
    from zeroshotsemanticsegmentation_amd import models as M

    for H in (1, 31, 64, 96, 150, 321, 500, 512, 768):
        n = H + 198
        reg = M._cb_conv1_1(H, 100)
        blocks = [("conv1_2", 1), ("conv2_1", 2), ("conv3_1", 3)]
        for name, L in blocks:
            plan = M._BandPlan(reg, reg, n, n, "cpu", L)
            cuts = M._band_cut(reg, n, L)
            r0, r1, c0, c1 = reg
            for a, e, rep, first in cuts:
                assert a % 2 == 0 and e % 2 == 0 and e - a >= 4
                if e <= r0:          # top / left band: the rows next to the cut stay pure through L layers, the representative too
                    assert a - 2 >= c0 + L and e <= r0 - L and rep == a // 2 - 1
                else:                # bottom / right band
                    assert a >= r1 + L and e + 1 < min(c1, n) - L and rep == e // 2
            if plan.ok:
                ty = {k: v[0].numpy() for k, v in plan.tabs.items()}
                kept = ty["crop"][:, 0]
                assert len(kept) == plan.Hc and np.all(np.diff(kept) > 0)
                back = ty["crop_bwd"]
                for i, y in enumerate(kept):
                    assert tuple(back[y]) == (i, 1)
                assert int((back[:, 1] == 0).sum()) == n - plan.Hc
                up, ub = ty["uncrop"], ty["uncrop_bwd"]
                A = np.zeros((plan.Hp, plan.Hpc))
                for p in range(plan.Hp):
                    A[p, up[p, 0]] = 1
                At = np.zeros((plan.Hpc, plan.Hp))
                for q in range(plan.Hpc):
                    At[q, ub[q, 0]:ub[q, 0] + ub[q, 1]] = 1
                assert np.array_equal(A.T, At)
                assert plan.Hpc == (plan.Hc + 1) // 2 and plan.Hp == (n + 1) // 2
            # the block's own layers + its pool: regions of the next block's input
            for _ in range(L):
                reg = M._cb_conv3x3(reg, n)
            reg = M._cb_pool(reg, n)
            n = (n + 1) // 2
    # the bench shape: 12 rows per side at 1/4 resolution, 40 at 1/2, 92 at full resolution
    n, reg = 710, M._cb_conv1_1(512, 100)
    assert [(a, e) for a, e, _, _ in M._band_cut(reg, n, 1)] == [(4, 96), (614, 706)]


def test_cu_mask_spec_parsing(monkeypatch):
    """SZN_FC6_CUMASK = "<n>[:low|:even]": which steps ask for the masked stream (models.masked_stream_wanted; off by default, small steps only,
    never with SZN_WGRAD_STREAM=0) -- host logic, no GPU"""
    from zeroshotsemanticsegmentation_amd import _lib as L
    from zeroshotsemanticsegmentation_amd import models
    monkeypatch.delenv("SZN_FC6_CUMASK", raising=False)
    monkeypatch.delenv("SZN_WGRAD_STREAM", raising=False)
    assert models._cumask_spec() == (0, "low") and not models.masked_stream_wanted(512 * 512)
    monkeypatch.setenv("SZN_FC6_CUMASK", "128:low")
    assert models._cumask_spec() == (128, "low")
    assert models.masked_stream_wanted(512 * 512) and models.masked_stream_wanted(2 * 512 * 512)
    assert not models.masked_stream_wanted(8 * 512 * 512)
    monkeypatch.setenv("SZN_WGRAD_STREAM", "0")
    assert not models.masked_stream_wanted(512 * 512)
    monkeypatch.delenv("SZN_WGRAD_STREAM")
    monkeypatch.setenv("SZN_FC6_CUMASK", "96:even")
    assert models._cumask_spec() == (96, "even")
    monkeypatch.setenv("SZN_FC6_CUMASK", "-5")
    assert models._cumask_spec()[0] == 0
    monkeypatch.setenv("SZN_FC6_CUMASK", "half")
    with pytest.raises(L.SznError):
        models._cumask_spec()


def test_gather_form_of_a_transposed_band_map():
    """models._gather_form (host logic): folding the multi-source runs in place (rows, then columns) and reading ONE source per pooled pixel
    afterwards is the transposed band map -- checked with integers (exact) against the {start, count} tables it was made from, for the
    un-crop map of a block and for the composed map of two adjacent blocks"""
    import torch
    from zeroshotsemanticsegmentation_amd import models
    dev = torch.device("cpu")
    reg, n = (48, 307, 1, 354), 355
    band = models._BandPlan(reg, reg, n, n, dev, 2)
    r = reg
    for _ in range(2):
        r = models._cb_conv3x3(r, n)
    rp = models._cb_pool(r, n)
    nplan = models._BandPlan(rp, rp, band.Hp, band.Wp, dev, 3)
    assert band.ok and nplan.ok
    ft = band.fused_with(nplan)
    cases = [(band.host["uncrop_bwd"], band.gather_uncrop, (band.Hp, band.Wp))]
    ty = [[int(a), int(b)] for a, b in ft["bwd_y"][0].tolist()]
    tx = [[int(a), int(b)] for a, b in ft["bwd_x"][1].tolist()]
    cases.append(((ty, tx), ft["bwd_gather"], (nplan.Hc, nplan.Wc)))
    rng = np.random.RandomState(0)
    for (ty, tx), gf, (Hs, Ws) in cases:
        src = rng.randint(-50, 50, size=(Hs, Ws)).astype(np.int64)
        want = np.array([[src[ys:ys + yc, xs:xs + xc].sum() for xs, xc in tx] for ys, yc in ty])
        d = src.copy()
        assert gf["runs_y"] is not None and gf["runs_x"] is not None
        for s0, c in gf["runs_y"].tolist():
            d[s0] = d[s0:s0 + c].sum(axis=0)
        for s0, c in gf["runs_x"].tolist():
            d[:, s0] = d[:, s0:s0 + c].sum(axis=1)
        t1y, t1x = (t.tolist() for t in gf["tabs"])
        assert all(c in (0, 1) for _, c in t1y + t1x)
        got = np.array([[d[ys, xs] if (yc and xc) else 0 for xs, xc in t1x] for ys, yc in t1y])
        assert np.array_equal(got, want)
        assert max(c for _, c in ty) > 1 and max(c for _, c in tx) > 1
