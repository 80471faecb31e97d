"""torch-CPU restatement of the SZN training step -- TEST / BASELINE INFRASTRUCTURE ONLY (never the product path).

The reference executes its hot loop with stock torch ops (models.py:114-160, utils.py:75-102,159-185, train.py:126-133);
its Python files cannot travel to the GPU box, so this is the same step restated on torch's CPU kernels, for bench.py's
`cpu_baseline` leg ("the build's torch-CPU restatement of the identical step") and as a second,
independent checker.  One deliberate difference from the reference's execution (not from its math): the fixed bilinear
`upscore` ConvTranspose2d(E, E, 64, stride 32) whose weight is non-zero on the channel diagonal only (models.py:11-24,109-112)
runs as a depthwise (groups = E) transposed convolution -- identical output and input gradient, without the dense
E x E x 64 x 64 weight gradient the reference computes and discards (train.py:324-327; 112 s of its 121 s at E = 300).
Pinned against tests/golden/g7_train_step_adam.npz in tests/test_oracle_golden.py.
"""
import time

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

_CFG = [("conv1_1", 3, 64, 100), ("conv1_2", 64, 64, 1), "P", ("conv2_1", 64, 128, 1), ("conv2_2", 128, 128, 1), "P",
        ("conv3_1", 128, 256, 1), ("conv3_2", 256, 256, 1), ("conv3_3", 256, 256, 1), "P",
        ("conv4_1", 256, 512, 1), ("conv4_2", 512, 512, 1), ("conv4_3", 512, 512, 1), "P",
        ("conv5_1", 512, 512, 1), ("conv5_2", 512, 512, 1), ("conv5_3", 512, 512, 1), "P"]


def _bilinear_1d(k=64):
    factor = (k + 1) // 2
    center = factor - 1 if k % 2 == 1 else factor - 0.5
    return 1.0 - np.abs(np.arange(k, dtype=np.float64) - center) / factor


class FCN32sTorch(nn.Module):
    def __init__(self, n_class):
        super().__init__()
        self.n_class = n_class
        for item in _CFG:
            if item != "P":
                name, ci, co, pad = item
                setattr(self, name, nn.Conv2d(ci, co, 3, padding=pad))
        self.fc6 = nn.Conv2d(512, 4096, 7)
        self.fc7 = nn.Conv2d(4096, 4096, 1)
        self.score_fr = nn.Conv2d(4096, n_class, 1)
        self.seenmask_score = nn.Conv2d(4096, 2, 1)
        f = _bilinear_1d()
        filt = torch.from_numpy((f[:, None] * f[None, :]).astype(np.float32))
        self.register_buffer("up_filt", filt.expand(n_class, 1, 64, 64).contiguous())
        up2 = torch.zeros(2, 2, 64, 64)
        up2[0, 0] = filt
        up2[1, 1] = filt
        self.seenmask_upscore = nn.Parameter(up2)

    def load_numpy(self, params):
        sd = self.state_dict()
        for k, v in params.items():
            key = "seenmask_upscore" if k == "seenmask_upscore.weight" else k
            if key in sd:
                sd[key].copy_(torch.from_numpy(np.ascontiguousarray(v)))
        return self

    def forward(self, x, mode="fcn", masks=None):
        H, W = x.shape[2:]
        h = x
        for item in _CFG:
            h = F.max_pool2d(h, 2, 2, ceil_mode=True) if item == "P" else F.relu(getattr(self, item[0])(h))
        h = F.relu(self.fc6(h))
        h = h * masks[0][:, :, None, None] if masks is not None else h
        h = F.relu(self.fc7(h))
        h = h * masks[1][:, :, None, None] if masks is not None else h
        out = []
        if mode in ("fcn", "both"):
            f = F.conv_transpose2d(self.score_fr(h), self.up_filt, stride=32, groups=self.n_class)
            out.append(f[:, :, 19:19 + H, 19:19 + W].contiguous())
        if mode in ("seenmask", "both"):
            s = F.conv_transpose2d(self.seenmask_score(h), self.seenmask_upscore, stride=32)
            out.append(s[:, :, 19:19 + H, 19:19 + W].contiguous())
        return out[0] if len(out) == 1 else tuple(out)


class FCN8sTorch(FCN32sTorch):
    """FCN8s skip head on the same trunk -- the PUBLIC pytorch-fcn FCN8s definition (score_pool3 / score_pool4 1x1 convs,
    upscore2 / upscore_pool4 ConvTranspose2d(E,E,4,s2), upscore8 ConvTranspose2d(E,E,16,s8), crops 5 / 9 / 31), with the
    transposed convolutions fixed to their bilinear initialisation (models.py:11-24,109-112) and run depthwise.
    PARITY UNPINNED: /root/reference has no FCN8s (its models.py:27 is FCN32s only), so there is no reference output to
    capture; this class is the checker for zeroshotsemanticsegmentation_amd.models.FCN8s and nothing more."""

    def __init__(self, n_class):
        super().__init__(n_class)
        self.score_pool3 = nn.Conv2d(256, n_class, 1)
        self.score_pool4 = nn.Conv2d(512, n_class, 1)
        for name, k in (("up2_filt", 4), ("up8_filt", 16)):
            f = _bilinear_1d(k)
            filt = torch.from_numpy((f[:, None] * f[None, :]).astype(np.float32))
            self.register_buffer(name, filt.expand(n_class, 1, k, k).contiguous())

    def forward(self, x, mode="fcn", masks=None):
        H, W = x.shape[2:]
        h = x
        pools = []
        for item in _CFG:
            if item == "P":
                h = F.max_pool2d(h, 2, 2, ceil_mode=True)
                pools.append(h)
            else:
                h = F.relu(getattr(self, item[0])(h))
        h = F.relu(self.fc6(h))
        h = h * masks[0][:, :, None, None] if masks is not None else h
        h = F.relu(self.fc7(h))
        h = h * masks[1][:, :, None, None] if masks is not None else h
        out = []
        if mode in ("fcn", "both"):
            E = self.n_class
            up2 = F.conv_transpose2d(self.score_fr(h), self.up2_filt, stride=2, groups=E)
            sp4 = self.score_pool4(pools[3])[:, :, 5:5 + up2.shape[2], 5:5 + up2.shape[3]]
            up4 = F.conv_transpose2d(up2 + sp4, self.up2_filt, stride=2, groups=E)
            sp3 = self.score_pool3(pools[2])[:, :, 9:9 + up4.shape[2], 9:9 + up4.shape[3]]
            f = F.conv_transpose2d(up4 + sp3, self.up8_filt, stride=8, groups=E)
            out.append(f[:, :, 31:31 + H, 31:31 + W].contiguous())
        if mode in ("seenmask", "both"):
            s = F.conv_transpose2d(self.seenmask_score(h), self.seenmask_upscore, stride=32)
            out.append(s[:, :, 19:19 + H, 19:19 + W].contiguous())
        return out[0] if len(out) == 1 else tuple(out)


def cosine_loss(score, target, embed):
    """utils.py:75-102 with the target embedding gathered by label; per-image loss, mean over images"""
    B, E = score.shape[:2]
    total = 0.0
    for b in range(B):
        m = target[b] >= 0
        s = score[b].permute(1, 2, 0)[m]                     # (N, E)
        t = embed[target[b][m]]
        cos = (s * t).sum(1) / (s.norm(dim=1) * t.norm(dim=1))
        n = m.sum()
        total = total + (n - cos.sum()) / n
    return total / B


def infer_lbl(score, embed):
    """utils.py:159-185"""
    B, E, H, W = score.shape
    s = score.permute(0, 2, 3, 1).reshape(-1, E)
    en = embed.norm(dim=1)
    en = torch.where(en == 0, torch.ones_like(en), en)
    sim = (s @ embed.t()) / (s.norm(dim=1, keepdim=True) * en[None])
    return sim.argmax(1).reshape(B, H, W)


def param_groups(model):
    ws = [p for n, p in model.named_parameters() if n.endswith(".weight") and not n.startswith("seenmask")]
    bs = [p for n, p in model.named_parameters() if n.endswith(".bias") and not n.startswith("seenmask")]
    return ws, bs


def timed_train_step(E, K, H, emb, x, target, steps=1, threads=None, arch="fcn32s"):
    """one (or more) full train steps on torch-CPU; returns seconds per phase of the LAST step"""
    if threads:
        torch.set_num_threads(threads)
    torch.manual_seed(1337)
    m = FCN8sTorch(E) if arch == "fcn8s" else FCN32sTorch(E)
    ws, bs = param_groups(m)
    opt = torch.optim.Adam([{"params": ws}, {"params": bs, "lr": 2e-5}], lr=1e-5)
    xt, tt, et = torch.from_numpy(x), torch.from_numpy(target), torch.from_numpy(emb)
    out = None
    for _ in range(steps):
        t0 = time.time()
        f = m(xt, "fcn")
        t1 = time.time()
        loss = cosine_loss(f, tt, et)
        t2 = time.time()
        with torch.no_grad():
            infer_lbl(f, et)
        t3 = time.time()
        opt.zero_grad()
        loss.backward()
        t4 = time.time()
        opt.step()
        t5 = time.time()
        out = {"fwd": t1 - t0, "loss": t2 - t1, "infer": t3 - t2, "bwd": t4 - t3, "adam": t5 - t4, "total": t5 - t0,
               "loss_value": float(loss)}
    return out
