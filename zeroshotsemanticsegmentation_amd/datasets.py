"""datasets.py -- PASCAL-VOC (SBD) and PASCAL-Context-33 readers with the reference's constructor / __getitem__ contract
(/root/reference/pascal_dataset.py:8-146, context_dataset.py:14-160), re-designed around the GPU data path:

  * `native=True` (what train.py uses): __getitem__ returns the RAW sample -- uint8 RGB (H,W,3) image and int64 label
    (H,W) with -1 = ignore.  The BGR / mean subtraction runs on the GPU (utils.image_to_device -> szn_image_u8_to_bgr_f32:
    3 B/px over PCIe instead of 12) and the per-pixel target embedding is gathered on the GPU from the K x E matrix
    (the reference materialises a dense (E,H,W) f32 `lbl_vec` per image on the host: 315 MB at 512x512, E = 300, and
    copies it to the device, trainer_fcn.py:93-95);
  * `native=False`: exactly the reference's outputs -- `transform=True` gives the f32 (3,H,W) BGR-minus-mean image and an
    int64 label, `embed_dim` adds the dense `lbl_vec` as `(lbl, lbl_vec)` -- so existing callers keep working;
  * split filtering (reference: opens EVERY label file at construction, context_dataset.py:75-94, pascal_dataset.py:62-89)
    runs once: the set of classes present in each label file is cached in `<data_dir>/<dataset>/label_presence.json`
    keyed by file size + mtime, and each split is a bit test against it.

Directory layout and split lists are the reference's: `<data_dir>/pascal/benchmark_RELEASE/dataset/{img,cls}`,
`<data_dir>/pascal/VOCdevkit/VOC2012/{JPEGImages,SegmentationClass}`, `<data_dir>/context/33_context_labels`, image ids
from `datasets/<dataset>/<split>.txt` relative to the CWD (override: split_dir=).  Downloading is out of scope (no network).
"""
import json
import os
import os.path as osp

import numpy as np
import torch

from . import utils

MEAN_BGR = np.array(utils.MEAN_BGR)

VOC_CLASSES = ['background', 'aeroplane', 'bicycle', 'bird', 'boat', 'bottle', 'bus', 'car', 'cat', 'chair', 'cow',
               'diningtable', 'dog', 'horse', 'motorbike', 'person', 'potted plant', 'sheep', 'sofa', 'train', 'tv/monitor']
CONTEXT33_CLASSES = ['aeroplane', 'bicycle', 'bird', 'boat', 'bottle', 'bus', 'car', 'cat', 'chair', 'cow', 'diningtable',
                     'dog', 'horse', 'motorbike', 'person', 'pottedplant', 'sheep', 'sofa', 'train', 'tvmonitor', 'sky',
                     'grass', 'ground', 'road', 'building', 'tree', 'water', 'mountain', 'wall', 'floor', 'track',
                     'keyboard', 'ceiling']
INVALID_BIT = 63          # "the label file contains invalid / unlabelled pixels" in the presence mask
# version of everything a cached presence mask depends on besides the file itself: the label decoding rules of the
# _read_label hooks (context: png - 1; voc: 255 -> -1) and the bit layout above.  Bump it when any of them changes: a cache
# written under another version (or for another class list) is discarded, never reused.
PRESENCE_SCHEMA = 2


def _open_image(path):
    import PIL.Image
    return np.array(PIL.Image.open(path), dtype=np.uint8)


def _open_png_label(path):
    import PIL.Image
    return np.array(PIL.Image.open(path), dtype=np.int32)


class _SegmentationDataset(torch.utils.data.Dataset):
    class_names = None
    mean_bgr = MEAN_BGR
    name = None

    def __init__(self, split='train', transform=False, embed_dim=None, one_hot_embed=False, data_dir='data', train_unseen=(),
                 val_unseen=(), native=False, split_dir=None, embed_arr=None, cache=True):
        self.split = split
        self._transform = transform
        self.embed_dim = embed_dim
        self.one_hot_embed = one_hot_embed
        self.data_dir = data_dir
        self.train_unseen = list(train_unseen)
        self.val_unseen = list(val_unseen)
        self.native = native
        if split not in ('train', 'train_seen', 'val'):
            raise Exception("unexpected split for %s dataset" % self.name)
        if (embed_dim or one_hot_embed) and not native:
            self.embed_arr = self._load_embeddings(embed_arr)
        self._cache_path = osp.join(data_dir, self.name, 'label_presence.json') if cache else None
        self._presence = self._load_cache()
        dirty = False
        split_dir = split_dir or osp.join('datasets', self.name)
        split_file = osp.join(split_dir, '%s.txt' % ('train' if split == 'train_seen' else split))
        banned = self._banned_mask()
        self.files = []
        for did in open(split_file):
            did = did.strip()
            if not did:
                continue
            img_file, lbl_file = self._paths(did)
            if banned:
                mask, fresh = self._presence_of(lbl_file)
                dirty |= fresh
                if mask & banned:
                    continue
            self.files.append({'img': img_file, 'lbl': lbl_file})
        if dirty:
            self._save_cache()

    # ---- per-dataset hooks -----------------------------------------------------------------------------------------
    def _paths(self, did):
        raise NotImplementedError

    def _read_label(self, lbl_file):
        """int32 (H,W) in the trainer's numbering: classes 0..K-1, -1 = invalid / ignore"""
        raise NotImplementedError

    def _banned_mask(self):
        raise NotImplementedError

    # ---- split filtering with a presence cache -----------------------------------------------------------------------
    @staticmethod
    def _bits(classes):
        m = 0
        for c in classes:
            m |= 1 << (INVALID_BIT if c < 0 else int(c))
        return m

    def _schema(self):
        import hashlib
        return "%d:%s" % (PRESENCE_SCHEMA, hashlib.sha1("|".join(self.class_names).encode()).hexdigest()[:12])

    def _load_cache(self):
        if self._cache_path and osp.exists(self._cache_path):
            try:
                c = json.load(open(self._cache_path))
            except (OSError, ValueError):
                return {}
            if isinstance(c, dict) and c.get("_schema") == self._schema() and isinstance(c.get("files"), dict):
                return c["files"]
        return {}

    def _save_cache(self):
        if not self._cache_path:
            return
        try:
            os.makedirs(osp.dirname(self._cache_path), exist_ok=True)
            tmp = self._cache_path + '.tmp%d' % os.getpid()
            json.dump({"_schema": self._schema(), "files": self._presence}, open(tmp, 'w'))
            os.replace(tmp, self._cache_path)
        except OSError:
            pass                                      # read-only dataset directory: scan again next time

    def _presence_of(self, lbl_file):
        st = os.stat(lbl_file)
        key = osp.relpath(lbl_file, self.data_dir)
        rec = self._presence.get(key)
        if rec and rec[0] == st.st_size and rec[1] == int(st.st_mtime):
            return rec[2], False
        mask = self._bits(np.unique(self._read_label(lbl_file)).tolist())
        self._presence[key] = [st.st_size, int(st.st_mtime), mask]
        return mask, True

    # ---- samples -------------------------------------------------------------------------------------------------------
    def _load_embeddings(self, embed_arr):
        if embed_arr is not None:
            return np.asarray(embed_arr, dtype=np.float32)
        if self.one_hot_embed:
            pkl = 'datasets/%s/embeddings/one_hot_%d_dim' % (self.name, len(self.class_names))
            if osp.exists(pkl + '.pkl'):
                return np.asarray(utils.load_obj(pkl), dtype=np.float32)
            return np.eye(len(self.class_names), dtype=np.float32)
        from .trainer_fcn import load_embeddings
        return load_embeddings(self.name, self.embed_dim)

    def __len__(self):
        return len(self.files)

    def __getitem__(self, index):
        f = self.files[index]
        img = _open_image(f['img'])
        lbl = self._read_label(f['lbl'])
        if self.native:
            return torch.from_numpy(np.ascontiguousarray(img)), torch.from_numpy(lbl.astype(np.int64))
        lbl_vec = None
        if self.embed_dim:
            idx = np.where(lbl == -1, 0, lbl)                          # row 0 stands in for ignore pixels
            lbl_vec = torch.from_numpy(np.ascontiguousarray(self.embed_arr[idx].transpose(2, 0, 1)))
        if self._transform:
            img, lbl = self.transform(img, lbl)
        return (img, (lbl, lbl_vec)) if self.embed_dim else (img, lbl)

    def transform(self, img, lbl):
        img = img[:, :, ::-1].astype(np.float64) - self.mean_bgr      # RGB -> BGR, minus mean, in float64 like the reference
        img = torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1))).float()
        return img, torch.from_numpy(np.ascontiguousarray(lbl)).long()

    def untransform(self, img, lbl):
        img = img.numpy().transpose(1, 2, 0) + self.mean_bgr
        return img.astype(np.uint8)[:, :, ::-1], lbl.numpy() if hasattr(lbl, 'numpy') else lbl


class PascalContext(_SegmentationDataset):
    """33-class PASCAL-Context (context_dataset.py:14-160): label PNGs are 1-based, 0 = unlabelled -> -1 after `lbl - 1`.
    EVERY split drops images that contain unlabelled pixels; 'train' also drops val_unseen classes, 'train_seen' all unseen."""
    class_names = CONTEXT33_CLASSES
    name = 'context'

    def _paths(self, did):
        return (osp.join(self.data_dir, 'pascal/VOCdevkit/VOC2012', 'JPEGImages/%s.jpg' % did),
                osp.join(self.data_dir, 'context/33_context_labels/%s.png' % did))

    def _read_label(self, lbl_file):
        return _open_png_label(lbl_file) - 1

    def _banned_mask(self):
        extra = {'train': self.val_unseen, 'train_seen': self.train_unseen + self.val_unseen, 'val': []}[self.split]
        return self._bits([-1] + list(extra))


class PascalVOC(_SegmentationDataset):
    """21-class PASCAL-VOC: train / train_seen from the SBD .mat files, val from the VOC2012 PNGs, 255 -> -1
    (pascal_dataset.py:8-146).  'train' drops images with val_unseen classes, 'train_seen' with any unseen class, 'val' none.
    (The reference's val branch tests a label left over from the previous loop iteration, pascal_dataset.py:76-89; a val
    split is never filtered here.)"""
    class_names = VOC_CLASSES
    name = 'pascal'

    def _paths(self, did):
        if self.split in ('train', 'train_seen'):
            d = osp.join(self.data_dir, 'pascal/benchmark_RELEASE/dataset')
            return osp.join(d, 'img/%s.jpg' % did), osp.join(d, 'cls/%s.mat' % did)
        d = osp.join(self.data_dir, 'pascal/VOCdevkit/VOC2012')
        return osp.join(d, 'JPEGImages/%s.jpg' % did), osp.join(d, 'SegmentationClass/%s.png' % did)

    def _read_label(self, lbl_file):
        if lbl_file.endswith('.mat'):
            import scipy.io
            lbl = scipy.io.loadmat(lbl_file)['GTcls'][0]['Segmentation'][0].astype(np.int32)
        else:
            lbl = _open_png_label(lbl_file)
        lbl[lbl == 255] = -1
        return lbl

    def _banned_mask(self):
        extra = {'train': self.val_unseen, 'train_seen': self.train_unseen + self.val_unseen, 'val': []}[self.split]
        return self._bits(list(extra))


MEAN_RGB_U8 = tuple(int(round(v)) for v in utils.MEAN_BGR[::-1])       # the uint8 colour closest to the mean: ~0 after the transform


PAD_LABEL = -2     # label of the pixels pad_collate adds: ignored by every loss / metric kernel, in BOTH phases (see pad_collate)


def pad_collate(batch):
    """collate_fn for NATIVE samples (uint8 (H,W,3) RGB image, int64 (H,W) label) of different sizes -- PASCAL images differ in
    size (context_dataset.py:143-150 never resizes; the reference therefore trains at batch size 1, train.py:82-84).  Every
    sample is padded at the bottom / right to the largest height / width of the batch: image pixels with the mean colour (zero
    after the BGR - mean transform, i.e. what conv1_1's own zero padding continues with), labels with PAD_LABEL = -2.  Phase 1:
    every loss, class-assignment and histogram kernel ignores negative labels (utils.py:33,60,84: `target >= 0`); per-image losses
    keep their own valid-pixel counts, so a padded batch is the mean of its images' losses.  Phase 2 (trainer_seenmask.py:55-56)
    turns the UNLABELLED value -1 into target 0 = "unseen" and counts it -- padding must not be counted, hence its own value:
    szn_seenmask_head and Trainer.binary_target leave labels below -1 out of the loss, the gradients and the confusion counts.  -> ((B,Hm,Wm,3) uint8, (B,Hm,Wm)
    int64).  Samples that carry the reference's (label, label_embedding) tuple keep the label only (the embedding is gathered
    on the GPU)."""
    imgs, lbls = [], []
    for img, lbl in batch:
        if isinstance(lbl, (tuple, list)):
            lbl = lbl[0]
        img, lbl = torch.as_tensor(img), torch.as_tensor(lbl)
        if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3:
            raise ValueError("pad_collate expects native samples: uint8 (H,W,3) images (datasets built with native=True)")
        imgs.append(img)
        lbls.append(lbl.long())
    Hm, Wm = max(i.shape[0] for i in imgs), max(i.shape[1] for i in imgs)
    out_i = torch.empty(len(imgs), Hm, Wm, 3, dtype=torch.uint8)
    out_i[:] = torch.tensor(MEAN_RGB_U8, dtype=torch.uint8)
    out_l = torch.full((len(imgs), Hm, Wm), PAD_LABEL, dtype=torch.int64)
    for k, (img, lbl) in enumerate(zip(imgs, lbls)):
        h, w = img.shape[:2]
        out_i[k, :h, :w] = img
        out_l[k, :h, :w] = lbl
    return out_i, out_l


def download(data_dir):
    """the reference fetches SBD / VOC2012 / the 33-class context labels over HTTP (pascal_dataset.py:148-172,
    context_dataset.py:162-183); this environment has no network, so the data must already be under data_dir"""
    raise IOError("no network access: place the datasets under %s (layout in the module docstring)" % data_dir)
