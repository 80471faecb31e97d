"""synthetic_dataset.py -- a dataset with the output contract of the reference's PascalVOC / PascalContext classes
(pascal_dataset.py:106-136, context_dataset.py:116-141) fed by the deterministic generator of synth.py.

There are no PASCAL images in this environment (no network); real-data loading is row F1 of SURVEY.md section 8-f
(next).  __getitem__ -> (img (3,H,W) f32 BGR minus mean, lbl (H,W) int64 with -1 = ignore) or, with embeddings,
(img, (lbl, lbl)) -- the dense per-pixel embedding volume of the reference (315 MB per 512x512 image at E = 300) is
replaced by the label itself: the target embedding is gathered on the GPU from the K x E matrix.
"""
import numpy as np
import torch

from . import synth

PASCAL_CLASSES = np.array(['background', 'aeroplane', 'bicycle', 'bird', 'boat', 'bottle', 'bus', 'car', 'cat', 'chair',
                           'cow', 'diningtable', 'dog', 'horse', 'motorbike', 'person', 'potted plant', 'sheep', 'sofa',
                           'train', 'tv/monitor'])


class SyntheticSegmentation(torch.utils.data.Dataset):
    mean_bgr = synth.MEAN_BGR

    def __init__(self, split='train', n_images=8, size=(512, 512), n_class=21, embed_dim=0, unseen=(), seed=1337,
                 class_names=None):
        self.split, self.n_images, self.size, self.embed_dim, self.seed = split, n_images, size, embed_dim, seed
        self.class_names = class_names if class_names is not None else (
            PASCAL_CLASSES if n_class == 21 else np.array(['class%d' % i for i in range(n_class)]))
        self.n_class = len(self.class_names)
        # 'train_seen' style splits only contain seen classes (the reference filters images, context_dataset.py:75-94)
        self.classes = [k for k in range(self.n_class) if k not in set(unseen)]
        self.offset = {'train': 0, 'train_seen': 0, 'val': 100000}.get(split, 0)

    def __len__(self):
        return self.n_images

    def __getitem__(self, index):
        H, W = self.size
        s = self.seed + self.offset + index
        img = torch.from_numpy(synth.make_images(1, H, W, seed=s)[0])
        lbl = torch.from_numpy(synth.make_labels(1, H, W, self.n_class, seed=s, classes=self.classes)[0])
        if self.embed_dim:
            return img, (lbl, lbl)
        return img, lbl

    def untransform(self, img, lbl):
        img = img.numpy().transpose(1, 2, 0) + self.mean_bgr
        return img.astype(np.uint8)[:, :, ::-1], lbl.numpy() if hasattr(lbl, 'numpy') else lbl
