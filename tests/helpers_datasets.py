"""Test glue: lay the tiny dataset of tests/golden/g10_datasets.npz out on disk in the reference's directory layout
(the same layout tools/capture_golden.py used when it ran the reference's dataset classes on these arrays)."""
import os

import numpy as np


def make_tiny_dataset(root, ids, imgs, ctx_lbls, voc_lbls):
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
            PIL.Image.fromarray(imgs[i]).save(d(root, "data", sub, did + ".jpg"), format="PNG")    # lossless bytes, .jpg name
        PIL.Image.fromarray(ctx_lbls[i].astype(np.uint8)).save(d(root, "data/context/33_context_labels", did + ".png"))
        PIL.Image.fromarray(voc_lbls[i].astype(np.uint8)).save(d(root, "data/pascal/VOCdevkit/VOC2012/SegmentationClass", did + ".png"))
        seg = np.empty((1,), dtype=[("Segmentation", object)])
        seg[0]["Segmentation"] = voc_lbls[i].astype(np.uint8)
        scipy.io.savemat(d(root, "data/pascal/benchmark_RELEASE/dataset/cls", did + ".mat"), {"GTcls": seg})
    for ds in ("context", "pascal"):
        for split in ("train", "val"):
            with open(d(root, "datasets", ds, split + ".txt"), "w") as f:
                f.write("\n".join(ids) + "\n")
