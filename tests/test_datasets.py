"""CPU: the dataset readers (zeroshotsemanticsegmentation_amd/datasets.py) against the reference's own dataset classes run on
the same tiny on-disk dataset (tests/golden/g10_datasets.npz, captured by tools/capture_golden.py g10): which images every
split keeps under the zero-shot filters, the exact __getitem__ tuple, the raw (native) form, and the label-presence cache."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers_datasets import make_tiny_dataset  # noqa: E402
from zeroshotsemanticsegmentation_amd import datasets  # noqa: E402

G = np.load(os.path.join(ROOT, "tests", "golden", "g10_datasets.npz"))
CASES = [("ctx", datasets.PascalContext, [0, 12], [16, 18]), ("voc", datasets.PascalVOC, [1, 13], [17, 19])]


@pytest.fixture()
def tiny(tmp_path, monkeypatch):
    make_tiny_dataset(str(tmp_path), [str(s) for s in G["ids"]], G["imgs"], G["ctx_png"], G["voc_png"])
    monkeypatch.chdir(tmp_path)                      # split lists are read relative to the CWD, like the reference
    return tmp_path


def ids_of(dset):
    return [os.path.basename(f["img"])[:-4] for f in dset.files]


@pytest.mark.parametrize("tag,cls,tu,vu", CASES)
def test_splits_and_getitem_match_reference(tiny, tag, cls, tu, vu):
    assert len(cls.class_names) == (33 if tag == "ctx" else 21)
    for split in ("train", "train_seen", "val"):
        d = cls(split=split, transform=True, embed_dim=20, data_dir="data", train_unseen=tu, val_unseen=vu)
        assert ids_of(d) == [str(s) for s in G["%s_%s_kept" % (tag, split)]], (tag, split)
        img, (lbl, vec) = d[0]
        assert img.dtype == torch.float32 and lbl.dtype == torch.int64 and vec.dtype == torch.float32
        assert np.array_equal(img.numpy(), G["%s_%s_img0" % (tag, split)])           # bit-identical transform
        assert np.array_equal(lbl.numpy(), G["%s_%s_lbl0" % (tag, split)])
        assert np.array_equal(vec.numpy(), G["%s_%s_vec0" % (tag, split)])
    d = cls(split="val", transform=False, embed_dim=None, data_dir="data")
    img, lbl = d[len(d) - 1]
    assert ids_of(d)[-1] == str(G["%s_raw_id" % tag])
    assert np.array_equal(np.asarray(img), G["%s_raw_img" % tag]) and np.array_equal(np.asarray(lbl), G["%s_raw_lbl" % tag])
    with pytest.raises(Exception):
        cls(split="test", data_dir="data")


@pytest.mark.parametrize("tag,cls,tu,vu", CASES)
def test_native_samples_and_presence_cache(tiny, tag, cls, tu, vu, monkeypatch):
    d = cls(split="train_seen", embed_dim=20, data_dir="data", train_unseen=tu, val_unseen=vu, native=True)
    img, lbl = d[0]
    assert img.dtype == torch.uint8 and tuple(img.shape[-1:]) == (3,) and lbl.dtype == torch.int64
    # raw sample + the reference transform == the reference's own output; the label is the same map
    want_lbl = G["%s_train_seen_lbl0" % tag]
    assert np.array_equal(lbl.numpy(), want_lbl)
    bgr = img.numpy()[:, :, ::-1].astype(np.float64) - datasets.MEAN_BGR
    assert np.array_equal(bgr.transpose(2, 0, 1).astype(np.float32), G["%s_train_seen_img0" % tag])
    # second construction answers every split from the cache: no label file is decoded
    cache = os.path.join("data", cls.name, "label_presence.json")
    assert os.path.exists(cache)
    calls = []
    orig = cls._read_label
    monkeypatch.setattr(cls, "_read_label", lambda self, f: calls.append(f) or orig(self, f))
    for split in ("train", "train_seen"):
        d2 = cls(split=split, data_dir="data", train_unseen=tu, val_unseen=vu, native=True)
        assert ids_of(d2) == [str(s) for s in G["%s_%s_kept" % (tag, split)]]
    assert calls == []
    # a modified label file invalidates its entry only
    f0 = d.files[0]["lbl"]
    os.utime(f0, (1, 1))
    cls(split="train_seen", data_dir="data", train_unseen=tu, val_unseen=vu, native=True)
    assert calls == [f0]


def test_loader_batches_native_samples(tiny):
    d = datasets.PascalContext(split="val", data_dir="data", native=True)
    loader = torch.utils.data.DataLoader(d, batch_size=2, shuffle=False)
    img, lbl = next(iter(loader))
    assert tuple(img.shape) == (2, 10, 12, 3) and img.dtype == torch.uint8 and tuple(lbl.shape) == (2, 10, 12)


def test_presence_cache_is_versioned(tiny, monkeypatch):
    """a cache written under another schema (label-decoding rules / bit layout / class list) is discarded, not reused"""
    import json
    cls = datasets.PascalContext
    cls(split="train_seen", data_dir="data", train_unseen=[0, 12], val_unseen=[16, 18], native=True)
    cache = os.path.join("data", cls.name, "label_presence.json")
    c = json.load(open(cache))
    assert c["_schema"].startswith("%d:" % datasets.PRESENCE_SCHEMA) and len(c["files"]) == len(G["ids"])
    # poison every mask and pretend it was written by an older schema: the split must still come out right
    for k in c["files"]:
        c["files"][k][2] = 0
    c["_schema"] = "1:" + c["_schema"].split(":")[1]
    json.dump(c, open(cache, "w"))
    d = cls(split="train_seen", data_dir="data", train_unseen=[0, 12], val_unseen=[16, 18], native=True)
    assert ids_of(d) == [str(s) for s in G["ctx_train_seen_kept"]]
    assert json.load(open(cache))["_schema"] == d._schema()
    # the pre-versioning flat format is not trusted either
    json.dump({k: [1, 1, 0] for k in c["files"]}, open(cache, "w"))
    d = cls(split="train_seen", data_dir="data", train_unseen=[0, 12], val_unseen=[16, 18], native=True)
    assert ids_of(d) == [str(s) for s in G["ctx_train_seen_kept"]]


def test_pad_collate_ragged_batch():
    """PASCAL images differ in size (context_dataset.py:143-150): batches > 1 are padded with mean-colour pixels labelled datasets.PAD_LABEL = -2 (ignored by both phases; -1 = "unlabelled" COUNTS in phase 2)"""
    rs = np.random.RandomState(0)
    sizes = [(10, 12), (7, 15), (13, 5)]
    batch = [(torch.from_numpy(rs.randint(0, 256, size=(h, w, 3)).astype(np.uint8)),
              torch.from_numpy(rs.randint(0, 33, size=(h, w)).astype(np.int64))) for h, w in sizes]
    img, lbl = datasets.pad_collate(batch)
    assert tuple(img.shape) == (3, 13, 15, 3) and img.dtype == torch.uint8
    assert tuple(lbl.shape) == (3, 13, 15) and lbl.dtype == torch.int64
    for k, (h, w) in enumerate(sizes):
        assert torch.equal(img[k, :h, :w], batch[k][0]) and torch.equal(lbl[k, :h, :w], batch[k][1])
        assert int((lbl[k] >= 0).sum()) == h * w                      # every padded pixel is ignored
        pad = torch.ones(13, 15, dtype=torch.bool)
        pad[:h, :w] = False
        assert datasets.PAD_LABEL < -1 and (lbl[k][pad] == datasets.PAD_LABEL).all()
        assert (img[k][pad] == torch.tensor(datasets.MEAN_RGB_U8, dtype=torch.uint8)).all()
    # the padding colour is the mean: at most half a grey level away from zero after the transform
    bgr = np.array(datasets.MEAN_RGB_U8[::-1], np.float64) - datasets.MEAN_BGR
    assert np.abs(bgr).max() <= 0.5
    # the reference's (label, label_embedding) tuple form keeps the label
    img2, lbl2 = datasets.pad_collate([(b[0], (b[1], torch.zeros(20, *b[1].shape))) for b in batch])
    assert torch.equal(lbl2, lbl)
    with pytest.raises(ValueError):
        datasets.pad_collate([(torch.zeros(3, 4, 4), torch.zeros(4, 4, dtype=torch.int64))])


def test_loader_with_pad_collate(tiny):
    d = datasets.PascalContext(split="val", data_dir="data", native=True)
    loader = torch.utils.data.DataLoader(d, batch_size=3, shuffle=False, collate_fn=datasets.pad_collate)
    img, lbl = next(iter(loader))
    assert tuple(img.shape) == (3, 10, 12, 3) and tuple(lbl.shape) == (3, 10, 12)
