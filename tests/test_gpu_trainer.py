"""GPU: the Trainer return contracts (reference trainer_fcn.py:83-147, trainer_seenmask.py:50-70) against the oracle:
`Trainer.forward` -> (score, loss, lbl_pred numpy int64, lbl_true cpu tensor) with plain / forced-unseen inference,
`Trainer.forward_szn` (both heads + seen-mask-stitched inference), the seen-mask trainer's binary target rule, and the
device-side validation metrics against the host metric code on the same predictions."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import szn_oracle as O  # noqa: E402
from zeroshotsemanticsegmentation_amd import models, optim, trainer_fcn, trainer_seenmask, utils  # noqa: E402
from zeroshotsemanticsegmentation_amd.synthetic_dataset import SyntheticSegmentation  # noqa: E402

E, K, H, W = 20, 33, 48, 56
UNSEEN, VAL_UNSEEN = [0, 12, 16, 18], [16, 18]


def make(tmp, forced=False):
    m = models.FCN32s(E)
    m.load_synthetic(1337, device=torch.device("cuda"))
    ds = SyntheticSegmentation(split="val", n_images=3, size=(H, W), n_class=K, embed_dim=E, seed=5)
    loader = torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False)
    ws = [getattr(m, n).weight for n in models._OPT_LAYERS]
    bs = [getattr(m, n).bias for n in models._OPT_LAYERS]
    opt = optim.FusedAdam([{"params": ws}, {"params": bs, "lr": 2e-5}], lr=1e-5)
    t = trainer_fcn.Trainer(cuda=True, model=m, optimizer=opt, train_loader=loader, val_loader=loader, log_dir=str(tmp),
                            dataset="context", max_epoch=1, tb_writer=None, pixel_embeddings=E, loss_func="cos", unseen=UNSEEN,
                            val_unseen=VAL_UNSEEN, forced_unseen=forced)
    return m, loader, t


@pytest.mark.parametrize("forced", [False, True])
def test_trainer_forward_contract_vs_oracle(tmp_path, forced):
    m, loader, t = make(tmp_path, forced)
    m.eval()
    emb = t.embeddings.cpu().numpy()
    data, target = next(iter(loader))
    with torch.no_grad():
        score, loss, lbl_pred, lbl_true = t.forward(data, target)
    assert tuple(score.shape) == (1, E, H, W) and isinstance(lbl_pred, np.ndarray) and lbl_pred.dtype == np.int64
    assert lbl_pred.shape == (1, H, W) and isinstance(lbl_true, torch.Tensor) and not lbl_true.is_cuda
    sn, lbl = score.cpu().numpy(), target[0].numpy()
    oloss, _, _ = O.cosine_loss(sn, lbl, embed=emb, want_grad=False)
    assert abs(float(loss) - float(oloss)) < 1e-5
    want = O.infer_lbl_forced_unseen(sn, lbl, emb, UNSEEN) if forced else O.infer_lbl(sn, emb)
    assert np.array_equal(lbl_pred, want) and np.array_equal(lbl_true.numpy(), lbl)
    # both heads + stitched inference (reference :123-147)
    with torch.no_grad():
        fs, loss2, pred_szn, _ = t.forward_szn(data, target)
        f2, s2 = m(data.cuda(), mode="both")
    assert torch.equal(fs, f2) and abs(float(loss2) - float(oloss)) < 1e-5
    assert np.array_equal(pred_szn, O.infer_lbl_szn(f2.cpu().numpy(), s2.cpu().numpy(), emb, UNSEEN))


def test_validate_device_metrics_equal_host_metrics(tmp_path):
    """Trainer.validate accumulates the {all, seen, unseen} histograms on the GPU; the logged metrics must equal the host
    metric code (utils.label_accuracy_score on numpy label maps, reference utils.py:131-154) on the same predictions"""
    m, loader, t = make(tmp_path)
    metrics = t.validate()
    row = open(os.path.join(str(tmp_path), "val_log.csv")).read().strip().split("\n")[1].split(",")
    m.eval()
    lts, lps, losses = [], [], []
    with torch.no_grad():
        for data, target in loader:
            _, loss, pred, lt = t._predict_device(data, target, False)       # the path validate() runs (fused head)
            lts.append(lt[0].cpu().numpy()); lps.append(pred[0].cpu().numpy()); losses.append(float(loss))
    want, seen_m, unseen_m = utils.label_accuracy_score(lts, lps, K, unseen=VAL_UNSEEN)
    np.testing.assert_allclose(np.array(metrics), np.array(want), rtol=1e-12, equal_nan=True)
    got_row = np.array([float(v) for v in row[2:15]])
    np.testing.assert_allclose(got_row, np.array([np.mean(losses)] + list(want) + list(seen_m) + list(unseen_m)), rtol=1e-6,
                               equal_nan=True)
    ck = torch.load(os.path.join(str(tmp_path), "checkpoint"), map_location="cpu", weights_only=False)
    assert set(ck) >= {"epoch", "iteration", "arch", "optim_state_dict", "model_state_dict", "best_mean_iu"}


def test_seenmask_trainer_forward_contract(tmp_path):
    m, loader, _ = make(tmp_path)
    m.eval()
    head = list(m.seenmask_score.parameters()) + list(m.seenmask_upscore.parameters())
    st = trainer_seenmask.Trainer(cuda=True, model=m, optimizer=optim.FusedAdam(head, lr=1e-3), train_loader=loader,
                                  val_loader=loader, log_dir=str(tmp_path), dataset="context", max_epoch=1, tb_writer=None,
                                  checkpoint={}, unseen=[0, 12])
    data, target = next(iter(loader))
    with torch.no_grad():
        score, loss, pred, lbl_true = st.forward(data, target)
    lbl = target[0].numpy()
    seen = [k for k in range(K) if k not in (0, 12)]
    want_t = np.isin(lbl, seen).astype(np.int64)                       # -1 (unlabelled) -> 0, not ignored (reference :55-56)
    assert np.array_equal(lbl_true.numpy(), want_t) and tuple(score.shape) == (1, 2, H, W)
    oloss, _, opred = O.cross_entropy2d(score.cpu().numpy(), want_t, size_average=True, want_grad=False)
    assert abs(float(loss) - float(oloss)) < 1e-5 * max(1.0, abs(float(oloss)))
    assert np.array_equal(pred, opred) and pred.dtype == np.int64


def test_embed_predict_equals_unfused_head(tmp_path):
    """FCN32s.embed_predict (what Trainer.validate uses for plain embedding inference) against forward + utils on the
    materialised score: same loss to rounding, same class assignment except near-ties"""
    m, loader, t = make(tmp_path)
    m.eval()
    data, target = next(iter(loader))
    lbl = target[0]
    with torch.no_grad():
        score = m(data.cuda(), mode="fcn")
        loss_u = utils.cosine_loss(score, lbl.cuda(), t.embeddings)
        pred_u = utils.infer_lbl_device(score, t.embeddings)
    loss, pred = m.embed_predict(data.cuda(), t.embeddings, lbl)
    assert abs(float(loss) - float(loss_u)) < 2e-6 and pred.dtype == torch.int64 and tuple(pred.shape) == (1, H, W)
    bad = pred != pred_u
    assert float(bad.float().mean()) < 2e-3
    l2, p2 = m.embed_predict(data.cuda(), t.embeddings)
    assert l2 is None and torch.equal(p2, pred)
