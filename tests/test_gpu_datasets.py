"""GPU: the dataset -> device path: uint8 upload + on-device BGR / mean transform bit-identical to
the reference's host transform, the Trainer fed with the reference's dense `(lbl, lbl_vec)` tuple, and `train.py` end to
end on a real-layout dataset directory (the tiny fixture of tests/golden/g10_datasets.npz)."""
import glob
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers_datasets import make_tiny_dataset  # noqa: E402
from zeroshotsemanticsegmentation_amd import datasets, models, optim, train, trainer_fcn, utils  # noqa: E402

G = np.load(os.path.join(ROOT, "tests", "golden", "g10_datasets.npz"))


@pytest.fixture()
def tiny(fast_tmp, monkeypatch):
    import pathlib
    make_tiny_dataset(fast_tmp, [str(s) for s in G["ids"]], G["imgs"], G["ctx_png"], G["voc_png"])
    monkeypatch.chdir(fast_tmp)
    return pathlib.Path(fast_tmp)


def test_image_to_device_bit_identical_to_reference_transform():
    imgs = G["imgs"]                                                     # (8, 10, 12, 3) uint8 RGB
    want = (imgs[..., ::-1].astype(np.float64) - datasets.MEAN_BGR).transpose(0, 3, 1, 2).astype(np.float32)
    got = utils.image_to_device(torch.from_numpy(imgs)).cpu().numpy()
    assert got.dtype == np.float32 and np.array_equal(got, want)
    assert np.array_equal(utils.image_to_device(imgs[0]).cpu().numpy()[0], G["ctx_val_img0"])     # the reference's own output
    big = np.random.RandomState(3).randint(0, 256, size=(2, 375, 500, 3)).astype(np.uint8)        # a PASCAL-sized image
    want = (big[..., ::-1].astype(np.float64) - datasets.MEAN_BGR).transpose(0, 3, 1, 2).astype(np.float32)
    assert np.array_equal(utils.image_to_device(torch.from_numpy(big).cuda()).cpu().numpy(), want)
    with pytest.raises(Exception):
        utils.image_to_device(torch.zeros(4, 4, 3))                      # not uint8


def test_trainer_accepts_dense_reference_tuple_and_native_samples(tiny):
    E = 20
    m = models.FCN32s(E)
    m.load_synthetic(1337, device=torch.device("cuda"))
    m.eval()
    ws = [getattr(m, n).weight for n in models._OPT_LAYERS]
    bs = [getattr(m, n).bias for n in models._OPT_LAYERS]
    opt = optim.FusedAdam([{"params": ws}, {"params": bs, "lr": 2e-5}], lr=1e-5)
    losses = {}
    for native in (False, True):
        d = datasets.PascalContext(split="train_seen", transform=True, embed_dim=E, data_dir="data", train_unseen=[0, 12],
                                   val_unseen=[16, 18], native=native)
        loader = torch.utils.data.DataLoader(d, batch_size=1, shuffle=False)
        t = trainer_fcn.Trainer(cuda=True, model=m, optimizer=opt, train_loader=loader, val_loader=loader,
                                log_dir=str(tiny / ("log%d" % native)), dataset="context", max_epoch=1, tb_writer=None,
                                pixel_embeddings=E, loss_func="cos", unseen=[0, 12, 16, 18], val_unseen=[16, 18])
        data, target = next(iter(loader))
        if not native:
            assert isinstance(target, (list, tuple)) and tuple(target[1].shape) == (1, E, 10, 12)      # the dense lbl_vec volume
        with torch.no_grad():
            score, loss, pred, lbl_true = t.forward(data, target)
        assert tuple(score.shape) == (1, E, 10, 12) and pred.shape == (1, 10, 12) and pred.dtype == np.int64
        losses[native] = float(loss)
    # dense per-pixel target volume (reference form) and label + on-device gather give the same loss
    assert abs(losses[False] - losses[True]) < 1e-6


def test_train_cli_on_real_layout_dataset(tiny, capsys):
    train.main(['-c', '18', '-ve', '1', '-dir', 'data', '-n', 'real', '--workers', '0'])
    log = glob.glob(os.path.join('data', 'logs', 'real_CFG_18_*'))[0]
    rows = open(os.path.join(log, 'train_log.csv')).read().strip().split('\n')
    assert len(rows) == 1 + len(G["ctx_train_seen_kept"])                   # one epoch over the train_seen split
    counts = open(os.path.join(log, 'counts.csv')).read().strip().split('\n')[1].split(',')
    assert [int(c) for c in counts] == [len(G["ctx_train_seen_kept"]), len(G["ctx_train_kept"]) - len(G["ctx_train_seen_kept"]),
                                        len(G["ctx_val_kept"])]
    srows = open(os.path.join(log, 'seenmask_train_log.csv')).read().strip().split('\n')
    assert len(srows) == 1 + 10 * len(G["ctx_train_kept"])                   # 10 seen-mask epochs over the train split
    assert all(np.isfinite(float(r.split(',')[2])) for r in rows[1:] + srows[1:])
    assert os.path.exists(os.path.join('data', 'context', 'label_presence.json'))


def _fresh_step(E, emb):
    from zeroshotsemanticsegmentation_amd import engine
    m = models.FCN32s(E).load_synthetic(1337).cuda().eval()
    return m, engine.TrainStep(m, emb, optimizer="adam", lr=1e-5, precision=torch.float32, fused_head=True)


def test_padded_batch_equals_mean_of_single_image_steps(tiny):
    """real-data batches > 1 (datasets.pad_collate): B = 2 == the two B = 1 steps averaged -- exactly, when the images have the
    same size (nothing is padded: the tiny dataset), and with padding, for the loss / valid-pixel bookkeeping: the padded
    pixels carry label -1 and are ignored by the loss, the class assignment and the histogram"""
    E = 20
    emb = trainer_fcn.load_embeddings("context", E)
    d = datasets.PascalContext(split="val", data_dir="data", native=True)
    loader = torch.utils.data.DataLoader(d, batch_size=2, shuffle=False, collate_fn=datasets.pad_collate)
    img, lbl = next(iter(loader))
    x, t = utils.image_to_device(img, torch.device("cuda")), lbl.cuda()
    m2, ts2 = _fresh_step(E, emb)
    loss2, _ = ts2.step(x, t)
    g2 = ts2.flat_gw.clone()
    singles = []
    for k in range(2):
        m1, ts1 = _fresh_step(E, emb)
        l1, _ = ts1.step(x[k:k + 1], t[k:k + 1])
        singles.append((float(l1), ts1.flat_gw.clone()))
    assert abs(float(loss2) - 0.5 * (singles[0][0] + singles[1][0])) < 1e-6
    want = 0.5 * (singles[0][1] + singles[1][1])
    assert float((g2 - want).abs().max()) < 1e-5 * float(want.abs().max())
    # ragged: a 10x12 and a 7x9 image -> (2,10,12) batch; per-image valid counts are the images' own pixel counts
    small_i, small_l = img[1][:7, :9].clone(), lbl[1][:7, :9].clone()
    bi, bl = datasets.pad_collate([(img[0], lbl[0]), (small_i, small_l)])
    assert tuple(bi.shape) == (2, 10, 12, 3) and int((bl[1] >= 0).sum()) == int((small_l >= 0).sum())
    m3, ts3 = _fresh_step(E, emb)
    ts3.hist.zero_()
    loss3, pred3 = ts3.step(utils.image_to_device(bi, torch.device("cuda")), bl.cuda())
    assert np.isfinite(float(loss3))
    assert ts3.stats[:, 1].cpu().tolist() == [float((lbl[0] >= 0).sum()), float((small_l >= 0).sum())]
    assert int(ts3.hist[0].sum()) == int((lbl[0] >= 0).sum()) + int((small_l >= 0).sum())
    # the same padded image alone gives the same per-image loss: the batch loss is the mean of its images' losses
    m4, ts4 = _fresh_step(E, emb)
    l4, _ = ts4.step(utils.image_to_device(bi[1:2], torch.device("cuda")), bl[1:2].cuda())
    assert abs(float(loss3) - 0.5 * (singles[0][0] + float(l4))) < 1e-6


def test_train_cli_batch2_on_real_layout_dataset(tiny):
    """train.py accepts --batch-size > 1 on real-layout data (pad_collate); phase 1 and the fused phase 2"""
    train.main(['-c', '18', '-ve', '1', '-dir', 'data', '-n', 'realb2', '--workers', '0', '--batch-size', '2'])
    log = glob.glob(os.path.join('data', 'logs', 'realb2_CFG_18_*'))[0]
    rows = open(os.path.join(log, 'train_log.csv')).read().strip().split('\n')
    n_seen = len(G["ctx_train_seen_kept"])
    assert len(rows) == 1 + (n_seen + 1) // 2
    srows = open(os.path.join(log, 'seenmask_train_log.csv')).read().strip().split('\n')
    assert len(srows) == 1 + 10 * ((len(G["ctx_train_kept"]) + 1) // 2)
    assert all(np.isfinite(float(r.split(',')[2])) for r in rows[1:] + srows[1:])
