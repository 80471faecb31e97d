"""GPU: does the training step actually learn?  A synthetic task whose labels are a function of the image (every 64x64 block is
painted in its class colour, plus noise): a few hundred fused steps (engine.TrainStep, bf16 and fp32) must drive the cosine loss
down and the train-time pixel accuracy of the nearest-embedding prediction far above chance.  This exercises the sign and scale
of every gradient, the two-group Adam wiring and the weight-image refresh end to end -- things a two-step comparison cannot."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zeroshotsemanticsegmentation_amd import engine, models, synth  # noqa: E402

K, E, H, BLK = 6, 20, 128, 64
COLORS = np.array([[-100, -100, -100], [120, -90, -90], [-90, 120, -90], [-90, -90, 120], [110, 110, -100], [-100, 110, 110]], np.float32)


def batch(seed, B=4):
    rs = np.random.RandomState(seed)
    lbl = rs.randint(0, K, size=(B, H // BLK, H // BLK)).repeat(BLK, 1).repeat(BLK, 2)
    img = COLORS[lbl].transpose(0, 3, 1, 2) + rs.randn(B, 3, H, H).astype(np.float32) * 10.0
    lbl = lbl.astype(np.int64)
    lbl[rs.rand(B, H, H) < 0.03] = -1
    return torch.from_numpy(np.ascontiguousarray(img, dtype=np.float32)).cuda(), torch.from_numpy(lbl).cuda()


@pytest.mark.parametrize("precision", [torch.bfloat16, torch.float32])
def test_fused_training_learns_a_colour_coded_task(precision):
    emb = synth.make_embeddings(K, E, seed=3)
    m = models.FCN32s(E)
    m.load_synthetic(1337, device=torch.device("cuda"))
    m.eval()                                              # no dropout: a deterministic learning curve
    ts = engine.TrainStep(m, emb, optimizer="adam", lr=5e-5, precision=precision, fused_head=True)
    first, accs = None, []
    for it in range(240):
        x, t = batch(it % 16)
        ts.hist.zero_()
        loss, pred = ts.step(x, t)
        if it % 40 == 39 or it == 0:
            h = ts.hist[0].double()
            accs.append(float(h.diag().sum() / h.sum()))
            if first is None:
                first = float(loss)
    last = float(loss)
    print("%s: loss %.4f -> %.4f, train pixel accuracy %s" % (precision, first, last, ["%.2f" % a for a in accs]))
    assert np.isfinite(last) and last < 0.25 * first
    assert accs[-1] > 0.9 and accs[-1] > accs[0] + 0.5           # chance = 1/6
