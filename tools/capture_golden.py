#!/usr/bin/env python
"""Capture golden input/output vectors from the REFERENCE implementation (/root/reference).

Runs only in the build container (the reference never travels to the GPU box).  It imports the
reference's models.py / utils.py with import-time stubs for the packages that are not installed
(fcn, gdown, torchvision), feeds them the deterministic inputs of zeroshotsemanticsegmentation_amd.synth
and writes small .npz fixtures to tests/golden/.  No reference source is copied: only data.

    python tools/capture_golden.py            # regenerates every fixture (~2 min on 8 cores)
"""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)

for name in ("fcn", "fcn.data", "fcn.utils", "gdown", "torchvision", "torchvision.models"):
    if name not in sys.modules:
        sys.modules[name] = types.ModuleType(name)
sys.path.insert(0, REF)

import torch  # noqa: E402
import models as ref_models  # noqa: E402  (reference)
import utils as ref_utils  # noqa: E402  (reference)

from zeroshotsemanticsegmentation_amd import synth  # noqa: E402

torch.set_num_threads(8)


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %-34s %8.1f KB" % (name + ".npz", os.path.getsize(path) / 1024.0))


def load_embed(dataset, E):
    arr = ref_utils.load_obj(os.path.join(REF, "datasets", dataset, "embeddings", "norm_embed_arr_%d" % E))
    return np.ascontiguousarray(np.asarray(arr, dtype=np.float32))


def build_ref_model(n_class, seed=1337):
    m = ref_models.FCN32s(n_class=n_class)
    params = synth.make_params(n_class, seed)
    sd = m.state_dict()
    for k, v in params.items():
        assert tuple(sd[k].shape) == v.shape, (k, sd[k].shape, v.shape)
        sd[k].copy_(torch.from_numpy(v))
    m.load_state_dict(sd)
    return m


def stats(t):
    a = t.detach().double()
    return np.array([a.sum().item(), a.abs().sum().item(), (a * a).sum().item()], dtype=np.float64)


def subsample(t):
    """deterministic small probe of an activation (1,C,H,W): channels ::max(C//4,1), rows/cols ::7"""
    a = t.detach()[0]
    return a[:: max(a.shape[0] // 4, 1), ::7, ::7].contiguous().numpy().copy()


# ---------------------------------------------------------------------------------------- G1
def g1_upsampling():
    w2 = ref_models.get_upsampling_weight(2, 2, 64).numpy()
    w3 = ref_models.get_upsampling_weight(3, 3, 4).numpy()
    w5 = ref_models.get_upsampling_weight(2, 2, 5).numpy()
    save("g1_upsampling_weight", filt64=w2[0, 0], offdiag64=w2[0, 1], k4=w3, k5=w5)


# ---------------------------------------------------------------------------------------- G2 / G3
STAGES = ["pool1", "pool2", "pool3", "pool4", "pool5", "relu6", "relu7", "drop6", "drop7", "score_fr",
          "seenmask_score", "upscore", "seenmask_upscore"]


def run_forward(m, x, train, seed=None):
    acts = {}
    hooks = []
    for s in STAGES:
        hooks.append(getattr(m, s).register_forward_hook(lambda mod, i, o, s=s: acts.__setitem__(s, (i[0].detach().clone(), o.detach().clone()))))
    m.train(train)
    if seed is not None:
        torch.manual_seed(seed)
    with torch.no_grad():
        f, s = m(torch.from_numpy(x), mode="both")
    for h in hooks:
        h.remove()
    return f, s, acts


def dropout_mask(inp, out):
    """per-(n,c) Dropout2d factor recovered from a hooked (input, output) pair: 0 or 2"""
    n, c = inp.shape[:2]
    num = (out * inp).reshape(n, c, -1).sum(-1)
    den = (inp * inp).reshape(n, c, -1).sum(-1)
    fac = torch.where(den > 0, num / den.clamp_min(1e-30), torch.full_like(den, 2.0))
    return fac.round().float().numpy()


def g2_g3_forward(m):
    for (H, W) in [(1, 1), (32, 32), (33, 47)]:
        x = synth.make_images(1, H, W, seed=2000 + H)
        f, s, acts = run_forward(m, x, train=False)
        d = dict(x=x, f=f.numpy(), s=s.numpy(), score_fr=acts["score_fr"][1].numpy(),
                 seenmask_score=acts["seenmask_score"][1].numpy(),
                 upscore_uncropped_stats=stats(acts["upscore"][1]))
        for st in ["pool1", "pool2", "pool3", "pool4", "pool5", "relu6", "relu7"]:
            d[st + "_stats"] = stats(acts[st][1])
            d[st + "_probe"] = subsample(acts[st][1])
            d[st + "_shape"] = np.array(acts[st][1].shape)
        save("g2_forward_eval_%dx%d" % (H, W), **d)
    # train mode: torch's own Dropout2d draws the masks; they are recovered from the hooks and stored
    H, W = 32, 32
    x = synth.make_images(1, H, W, seed=2000 + H)
    f, s, acts = run_forward(m, x, train=True, seed=4242)
    m6 = dropout_mask(*acts["drop6"])
    m7 = dropout_mask(*acts["drop7"])
    assert set(np.unique(m6)) <= {0.0, 2.0} and set(np.unique(m7)) <= {0.0, 2.0}
    save("g3_forward_train_32x32", x=x, f=f.numpy(), s=s.numpy(), mask6=m6, mask7=m7,
         score_fr=acts["score_fr"][1].numpy(), relu7_stats=stats(acts["relu7"][1]))
    m.eval()


# ---------------------------------------------------------------------------------------- G4
def g4_losses():
    for (ds, K, E, H, W) in [("pascal", 21, 20, 32, 32), ("context", 33, 20, 24, 40), ("context", 33, 300, 12, 12),
                             ("pascal", 21, 300, 9, 14)]:
        emb = load_embed(ds, E)
        assert emb.shape == (K, E)
        score = synth.uniform(5000 + E + H, (1, E, H, W), -1.5, 1.5)
        target = synth.make_labels(1, H, W, K, seed=5100 + E + H, block=4, ignore_frac=0.1)
        lbl0 = np.where(target < 0, 0, target)                    # context_dataset.py:128-141
        tembed = np.ascontiguousarray(emb[lbl0[0]].transpose(2, 0, 1)[None])   # (1,E,H,W)
        out = dict(score=score, target=target, embed=emb)
        for name, fn in (("cos", ref_utils.cosine_loss), ("mse", ref_utils.mse_loss)):
            s = torch.from_numpy(score).clone().requires_grad_(True)
            loss = fn(s, torch.from_numpy(target), torch.from_numpy(tembed))
            loss.backward()
            out[name + "_loss"] = np.array(loss.item(), dtype=np.float64)
            out[name + "_dscore"] = s.grad.numpy()
        save("g4_embed_losses_%s_E%d" % (ds, E), **out)
    for (C, H, W, avg, n) in [(21, 16, 20, False, 1), (2, 32, 32, True, 1), (2, 8, 8, True, 3)]:
        score = synth.uniform(5200 + C + n, (n, C, H, W), -3.0, 3.0)
        target = synth.make_labels(n, H, W, C, seed=5300 + C + n, block=2, ignore_frac=0.1 if C > 2 else 0.0)
        s = torch.from_numpy(score).clone().requires_grad_(True)
        loss = ref_utils.cross_entropy2d(s, torch.from_numpy(target), size_average=avg)
        loss.backward()
        pred = s.data.max(1)[1].numpy()
        save("g4_ce2d_C%d_n%d" % (C, n), score=score, target=target, size_average=np.array(int(avg)),
             loss=np.array(loss.item(), dtype=np.float64), dscore=s.grad.numpy(), pred=pred.astype(np.int64))


def g4w_weighted_ce():
    """cross_entropy2d with class weights (reference utils.py:19,46: F.nll_loss(weight=...)); kept apart from g4 so that the
    existing fixtures stay byte-identical"""
    for (C, H, W, avg, n) in [(21, 16, 20, False, 1), (2, 8, 8, True, 3)]:
        score = synth.uniform(6200 + C + n, (n, C, H, W), -3.0, 3.0)
        target = synth.make_labels(n, H, W, C, seed=6300 + C + n, block=2, ignore_frac=0.1)
        weight = synth.uniform(6400 + C, (C,), 0.25, 2.0)
        if C > 2:
            weight[3] = 0.0                                   # a class that does not count at all
        s = torch.from_numpy(score).clone().requires_grad_(True)
        loss = ref_utils.cross_entropy2d(s, torch.from_numpy(target), weight=torch.from_numpy(weight), size_average=avg)
        loss.backward()
        save("g4_ce2d_weighted_C%d_n%d" % (C, n), score=score, target=target, weight=weight, size_average=np.array(int(avg)),
             loss=np.array(loss.item(), dtype=np.float64), dscore=s.grad.numpy())


# ---------------------------------------------------------------------------------------- G5
def masked(emb, rows):
    out = np.zeros_like(emb)
    out[rows] = emb[rows]
    return out


def sims_margin(score, emb):
    """top-2 margin of the similarity the reference maximises, computed in float64 (for filtering near-ties)"""
    E = score.shape[1]
    s = score[0].reshape(E, -1).T.astype(np.float64)
    e = emb.astype(np.float64)
    en = np.linalg.norm(e, axis=1)
    en[en == 0] = 1
    sim = (s @ e.T) / (np.linalg.norm(s, axis=1, keepdims=True) * en[None])
    top = np.sort(sim, axis=1)
    return (top[:, -1] - top[:, -2]).reshape(score.shape[2:]).astype(np.float32)


def g5_infer():
    for (ds, K, E, H, W, train_unseen, val_unseen) in [("context", 33, 20, 32, 32, [0, 12], [16, 18]),
                                                       ("pascal", 21, 20, 24, 24, [1, 13], [6, 7, 10, 14, 15, 16, 17, 18, 19, 20]),
                                                       ("context", 33, 300, 16, 16, [0, 12], [16, 18])]:
        emb = load_embed(ds, E)
        unseen = train_unseen + val_unseen                       # train.py:139
        seen = [k for k in range(K) if k not in unseen]          # trainer_fcn.py:44
        seen_e, unseen_e = masked(emb, seen), masked(emb, unseen)    # trainer_fcn.py:56-58
        score = synth.uniform(6000 + E + K, (1, E, H, W), -1.0, 1.0)
        target = synth.make_labels(1, H, W, K, seed=6100 + E + K, block=4, ignore_frac=0.1)
        smask = synth.uniform(6200 + E + K, (1, 2, H, W), -1.0, 1.0)
        smask[0, :, 0, :4] = 0.25                                 # exact ties in the 2-channel argmax
        ts, tt = torch.from_numpy(score), torch.from_numpy(target)
        t = lambda a: torch.from_numpy(a)
        d = dict(score=score, target=target, seenmask=smask, embed=emb, unseen=np.array(unseen), seen=np.array(seen))
        d["pred_all"] = ref_utils.infer_lbl(ts, t(emb))
        d["pred_seen_only"] = ref_utils.infer_lbl(ts, t(seen_e))
        d["pred_unseen_only"] = ref_utils.infer_lbl(ts, t(unseen_e))
        d["pred_szn"] = ref_utils.infer_lbl_szn(ts, t(smask), t(seen_e), t(unseen_e))
        d["pred_forced"] = ref_utils.infer_lbl_forced_unseen(ts, tt, t(seen_e), t(unseen_e), unseen)
        d["margin_all"] = sims_margin(score, emb)
        d["margin_seen_only"] = sims_margin(score, seen_e)
        d["margin_unseen_only"] = sims_margin(score, unseen_e)
        for k in list(d):
            if k.startswith("pred_"):
                assert d[k].dtype == np.int64 and d[k].shape == (1, H, W)
        save("g5_infer_%s_E%d" % (ds, E), **d)


# ---------------------------------------------------------------------------------------- G6
def g6_metrics():
    K = 33
    lt = [synth.make_labels(1, 20, 24, K, seed=7000 + i, block=4, ignore_frac=0.1)[0] for i in range(3)]
    lp = [synth.make_labels(1, 20, 24, K, seed=7100 + i, block=3, ignore_frac=0.0)[0] for i in range(3)]
    lp = [np.where(synth.uniform01(7200 + i, 480).reshape(20, 24) < 0.6, np.where(a < 0, 0, a), b) for i, (a, b) in enumerate(zip(lt, lp))]
    with np.errstate(all="ignore"):
        m_all = ref_utils.label_accuracy_score(lt, lp, K)
        m3 = ref_utils.label_accuracy_score(lt, lp, K, unseen=[16, 18])
        # adversarial: classes absent from GT (NaN paths) and from predictions
        lt2 = [np.full((4, 5), 3, dtype=np.int64), np.array([[-1, 7, 7, 40, 2]] * 4, dtype=np.int64)]
        lp2 = [np.full((4, 5), 5, dtype=np.int64), np.array([[0, 7, 1, 2, 2]] * 4, dtype=np.int64)]
        m_adv = ref_utils.label_accuracy_score(lt2, lp2, K)
        m_adv3 = ref_utils.label_accuracy_score(lt2, lp2, K, unseen=[7, 9])
        hist = sum(ref_utils._fast_hist(a.flatten(), b.flatten(), K) for a, b in zip(lt, lp))
    save("g6_metrics", lt=np.stack(lt), lp=np.stack(lp), metrics=np.array(m_all), metrics3=np.array(m3),
         lt_adv0=lt2[0], lt_adv1=lt2[1], lp_adv0=lp2[0], lp_adv1=lp2[1], metrics_adv=np.array(m_adv),
         metrics_adv3=np.array(m_adv3), hist=hist.astype(np.int64))


# ---------------------------------------------------------------------------------------- G7 / G8
PROBE_PARAMS = ["conv1_1.weight", "conv1_1.bias", "conv1_2.weight", "conv3_2.weight", "conv5_3.bias", "fc6.weight",
                "fc7.weight", "fc7.bias", "score_fr.weight", "score_fr.bias"]


def param_groups(m):
    """train.py:302-331 restated for this capture: Conv2d weights | Conv2d biases, seenmask layers excluded"""
    ws, bs = [], []
    for name, mod in m.named_modules():
        if name in ("seenmask_score", "seenmask_upscore"):
            continue
        if isinstance(mod, torch.nn.Conv2d):
            ws.append(mod.weight)
            bs.append(mod.bias)
    return ws, bs


def probe_idx(n, cnt=64):
    return (np.arange(cnt, dtype=np.int64) * 2654435761 % n).astype(np.int64)


def g7_train_step():
    K, E, H, W = 33, 20, 32, 32
    emb = load_embed("context", E)
    x = synth.make_images(1, H, W, seed=2032)
    target = synth.make_labels(1, H, W, K, seed=7300, block=8, ignore_frac=0.05)
    lbl0 = np.where(target < 0, 0, target)
    tembed = np.ascontiguousarray(emb[lbl0[0]].transpose(2, 0, 1)[None])
    for optname in ("adam", "sgd"):
        m = build_ref_model(E)
        m.eval()                       # dropout off: the step is deterministic
        ws, bs = param_groups(m)
        if optname == "adam":          # train.py:130-133
            lr = 1e-5
            opt = torch.optim.Adam([{"params": ws}, {"params": bs, "lr": lr * 2}], lr=lr)
        else:                          # train.py:126-129
            lr = 1e-10
            opt = torch.optim.SGD([{"params": ws}, {"params": bs, "lr": lr * 2, "weight_decay": 0}], lr=lr,
                                  momentum=0.99, weight_decay=0.0005)
        out = dict(x=x, target=target, embed=emb, lr=np.array(lr))
        before = {k: v.detach().clone() for k, v in m.state_dict().items()}
        for it in range(2):
            score = m(torch.from_numpy(x), mode="fcn")
            loss = ref_utils.cosine_loss(score, torch.from_numpy(target), torch.from_numpy(tembed))
            pred = ref_utils.infer_lbl(score, torch.from_numpy(emb))
            opt.zero_grad()
            loss.backward()
            if it == 0:
                out["loss0"] = np.array(loss.item(), dtype=np.float64)
                out["score0"] = score.detach().numpy()
                out["pred0"] = pred
                out["margin0"] = sims_margin(score.detach().numpy(), emb)
                out["score_fr_wgrad_sum"] = np.array(m.score_fr.weight.grad.double().sum().item())   # trainer_fcn.py:161
                out["upscore_wgrad_sum"] = np.array(m.upscore.weight.grad.double().sum().item())      # trainer_fcn.py:162
                for k in PROBE_PARAMS:
                    g = dict(m.named_parameters())[k].grad
                    out["grad_stats/" + k] = stats(g)
                    out["grad_probe/" + k] = g.flatten()[torch.from_numpy(probe_idx(g.numel()))].numpy()
            else:
                out["loss1"] = np.array(loss.item(), dtype=np.float64)
            opt.step()
            if it == 0:
                for k in PROBE_PARAMS:
                    p = dict(m.named_parameters())[k].detach()
                    idx = torch.from_numpy(probe_idx(p.numel()))
                    out["delta_probe/" + k] = (p.flatten()[idx].double() - before[k].flatten()[idx].double()).numpy()
        for k in PROBE_PARAMS:
            p = dict(m.named_parameters())[k].detach()
            idx = torch.from_numpy(probe_idx(p.numel()))
            out["delta2_probe/" + k] = (p.flatten()[idx].double() - before[k].flatten()[idx].double()).numpy()
        save("g7_train_step_%s" % optname, **out)


def g8_seenmask_step():
    K, E, H, W = 33, 20, 32, 32
    unseen = [0, 12]                                              # cfg 18 train_unseen (configs.py:117)
    x = synth.make_images(1, H, W, seed=2032)
    target = synth.make_labels(1, H, W, K, seed=7400, block=8, ignore_frac=0.05)
    m = build_ref_model(E)
    m.eval()
    for p in m.parameters():                                      # train.py:166-171
        p.requires_grad = False
    for p in m.seenmask_score.parameters():
        p.requires_grad = True
    for p in m.seenmask_upscore.parameters():
        p.requires_grad = True
    params = list(m.seenmask_score.parameters()) + list(m.seenmask_upscore.parameters())
    opt = torch.optim.Adam(params, lr=1e-3)                       # train.py:174-175
    seen = [k for k in range(K) if k not in unseen]
    bin_target = np.in1d(target.ravel(), seen).reshape(target.shape).astype(np.int64)   # trainer_seenmask.py:55-56
    before = {k: v.detach().clone() for k, v in m.state_dict().items()}
    score = m(torch.from_numpy(x), mode="seenmask")
    loss = ref_utils.cross_entropy2d(score, torch.from_numpy(bin_target), size_average=True)
    pred = score.data.max(1)[1].numpy()
    opt.zero_grad()
    loss.backward()
    out = dict(x=x, target=target, bin_target=bin_target, unseen=np.array(unseen), loss=np.array(loss.item(), dtype=np.float64),
               score=score.detach().numpy(), pred=pred.astype(np.int64),
               dW_score=m.seenmask_score.weight.grad.numpy(), db_score=m.seenmask_score.bias.grad.numpy(),
               dW_up_stats=stats(m.seenmask_upscore.weight.grad),
               dW_up_probe=m.seenmask_upscore.weight.grad[:, :, ::9, ::9].contiguous().numpy())
    opt.step()
    out["delta_W_score_probe"] = (m.seenmask_score.weight.detach().double() - before["seenmask_score.weight"].double()).flatten()[:256].numpy()
    out["delta_W_up_probe"] = (m.seenmask_upscore.weight.detach().double() - before["seenmask_upscore.weight"].double())[:, :, ::9, ::9].contiguous().numpy()
    save("g8_seenmask_step", **out)


# ---------------------------------------------------------------------------------------- G9
def g9_embeddings():
    import hashlib
    for ds, K in (("pascal", 21), ("context", 33)):
        # every width the reference CLI accepts (train.py:31: -e {2,5,10,20,21,50,100,200,300})
        for E in (2, 5, 10, 20, 21, 50, 100, 200, 300):
            emb = load_embed(ds, E)
            assert emb.shape == (K, E) and emb.dtype == np.float32
            np.save(os.path.join(OUT, "embeddings_%s_%d.npy" % (ds, E)), emb)
            np.save(os.path.join(ROOT, "zeroshotsemanticsegmentation_amd", "data", "embeddings_%s_%d.npy" % (ds, E)), emb)
            print("embeddings_%s_%d sha256[:16]=%s" % (ds, E, hashlib.sha256(emb.tobytes()).hexdigest()[:16]))


def g0_configs():
    """the numbered experiment dicts (reference configs.py) as data, for the host-side parity test"""
    import json
    import configs as ref_configs  # reference
    with open(os.path.join(OUT, "configs.json"), "w") as f:
        json.dump({str(k): v for k, v in ref_configs.configurations.items()}, f, indent=1, sort_keys=True)
    print("wrote configs.json")


# ---------------------------------------------------------------------------------------- G10
def make_tiny_dataset(root, ids, imgs, ctx_lbls, voc_lbls):
    """lay a tiny synthetic dataset out on disk in the reference's directory layout (used by the capture here and by
    tests/test_datasets.py on the fixture's arrays).  Images are written as PNG bytes under the .jpg names (PIL opens by
    content), so the decode is lossless."""
    import PIL.Image
    import scipy.io
    d = os.path.join
    for sub in ("pascal/VOCdevkit/VOC2012/JPEGImages", "pascal/VOCdevkit/VOC2012/SegmentationClass",
                "pascal/benchmark_RELEASE/dataset/img", "pascal/benchmark_RELEASE/dataset/cls", "context/33_context_labels"):
        os.makedirs(d(root, "data", sub), exist_ok=True)
    for sub in ("datasets/context", "datasets/pascal"):
        os.makedirs(d(root, sub), exist_ok=True)
    for i, did in enumerate(ids):
        for sub in ("pascal/VOCdevkit/VOC2012/JPEGImages", "pascal/benchmark_RELEASE/dataset/img"):
            PIL.Image.fromarray(imgs[i]).save(d(root, "data", sub, did + ".jpg"), format="PNG")
        PIL.Image.fromarray(ctx_lbls[i].astype(np.uint8)).save(d(root, "data/context/33_context_labels", did + ".png"))
        PIL.Image.fromarray(voc_lbls[i].astype(np.uint8)).save(d(root, "data/pascal/VOCdevkit/VOC2012/SegmentationClass", did + ".png"))
        seg = np.empty((1,), dtype=[("Segmentation", object)])
        seg[0]["Segmentation"] = voc_lbls[i].astype(np.uint8)
        scipy.io.savemat(d(root, "data/pascal/benchmark_RELEASE/dataset/cls", did + ".mat"), {"GTcls": seg})
    for ds in ("context", "pascal"):
        for split in ("train", "val"):
            with open(d(root, "datasets", ds, split + ".txt"), "w") as f:
                f.write("\n".join(ids) + "\n")


def g10_datasets():
    """the dataset -> trainer contract (context_dataset.py:53-150, pascal_dataset.py:43-145): which images each split keeps
    under the zero-shot filters, and the exact (img, (lbl, lbl_vec)) tuple of __getitem__, on a tiny on-disk dataset"""
    import tempfile
    H, W, n = 10, 12, 8
    ids = ["2008_%06d" % i for i in range(n)]
    rng = np.random.RandomState(1337)
    imgs = rng.randint(0, 256, size=(n, H, W, 3)).astype(np.uint8)
    # context labels are 1-based PNG values (0 = unlabelled); class sets chosen so that every filter rule fires
    ctx_sets = [[1, 5], [1, 13], [17, 19], [3, 4, 0], [2, 6], [13, 17], [8], [21, 33]]
    voc_sets = [[0, 3], [1, 15], [17, 19], [255, 2], [13, 5], [6, 0], [0], [20, 255]]
    ctx = np.stack([np.array(s)[rng.randint(0, len(s), size=(H, W))] for s in ctx_sets]).astype(np.int32)
    voc = np.stack([np.array(s)[rng.randint(0, len(s), size=(H, W))] for s in voc_sets]).astype(np.int32)
    out = {"ids": np.array(ids), "imgs": imgs, "ctx_png": ctx, "voc_png": voc}
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as root:
        make_tiny_dataset(root, ids, imgs, ctx, voc)
        for ds in ("context", "pascal"):
            os.symlink(os.path.join(REF, "datasets", ds, "embeddings"), os.path.join(root, "datasets", ds, "embeddings"))
        os.chdir(root)
        try:
            import context_dataset as ref_ctx  # reference
            import pascal_dataset as ref_voc  # reference
            for tag, cls, tu, vu in (("ctx", ref_ctx.PascalContext, [0, 12], [16, 18]), ("voc", ref_voc.PascalVOC, [1, 13], [17, 19])):
                for split in ("train", "train_seen", "val"):
                    dset = cls(split=split, transform=True, embed_dim=20, data_dir="data", train_unseen=tu, val_unseen=vu)
                    kept = [os.path.basename(f["img"])[:-4] for f in dset.files]
                    out["%s_%s_kept" % (tag, split)] = np.array(kept)
                    if kept:
                        img, (lbl, vec) = dset[0]
                        out["%s_%s_img0" % (tag, split)] = img.numpy()
                        out["%s_%s_lbl0" % (tag, split)] = lbl.numpy()
                        out["%s_%s_vec0" % (tag, split)] = vec.numpy()
                # no embeddings, no transform: raw arrays
                dset = cls(split="val", transform=False, embed_dim=None, data_dir="data")
                img, lbl = dset[len(dset) - 1]
                out["%s_raw_img" % tag] = np.asarray(img)
                out["%s_raw_lbl" % tag] = np.asarray(lbl)
                out["%s_raw_id" % tag] = np.array(os.path.basename(dset.files[-1]["img"])[:-4])
        finally:
            os.chdir(cwd)
    save("g10_datasets", **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    only = set(sys.argv[1:])
    run = lambda tag: not only or tag in only
    if run("g0"): g0_configs()
    if run("g1"): g1_upsampling()
    if run("g9"): g9_embeddings()
    if run("g4"): g4_losses()
    if run("g4w"): g4w_weighted_ce()
    if run("g5"): g5_infer()
    if run("g6"): g6_metrics()
    if run("g2") or run("g3"):
        m = build_ref_model(20)
        g2_g3_forward(m)
    if run("g7"): g7_train_step()
    if run("g8"): g8_seenmask_step()
    if run("g10"): g10_datasets()


if __name__ == "__main__":
    main()
