"""GPU: the CLI entry point end to end on the synthetic dataset -- phase 1 (fused TrainStep), validation with the
seen / unseen metric split, checkpoint with the reference's dict keys, phase 2 (seen-mask head, backbone frozen),
then the test_all mode on the saved checkpoint (full SZN inference)."""
import glob
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zeroshotsemanticsegmentation_amd import train  # noqa: E402


def test_train_cfg18_then_test_all(fast_tmp, capsys):
    d = fast_tmp
    # cfg 18: context, 20-d embeddings, cosine loss, Adam, seen-mask phase (10 epochs, -se is ignored like the reference)
    train.main(['-c', '18', '-ve', '1', '-dir', d, '-n', 'smoke', '--synthetic', '2', '64', '64', '--workers', '0'])
    logs = glob.glob(os.path.join(d, 'logs', 'smoke_CFG_18_*'))
    assert len(logs) == 1
    log = logs[0]
    for f in ('config.yaml', 'train_log.csv', 'val_log.csv', 'seenmask_train_log.csv', 'seenmask_val_log.csv', 'counts.csv',
              'checkpoint', 'best'):
        assert os.path.exists(os.path.join(log, f)), f
    rows = open(os.path.join(log, 'train_log.csv')).read().strip().split('\n')
    assert rows[0].startswith('epoch,iteration,train/loss') and len(rows) == 1 + 2          # 2 images, 1 epoch
    vrows = open(os.path.join(log, 'val_log.csv')).read().strip().split('\n')
    assert 'val/unseen/mean_iu' in vrows[0] and len(vrows) == 2
    srows = open(os.path.join(log, 'seenmask_train_log.csv')).read().strip().split('\n')
    assert len(srows) == 1 + 10 * 2                                                         # 10 epochs x 2 images
    ck = torch.load(os.path.join(log, 'best'), map_location='cpu', weights_only=False)
    assert set(ck) >= {'epoch', 'iteration', 'arch', 'optim_state_dict', 'model_state_dict', 'best_mean_iu'}
    assert ck['arch'] == 'FCN32s' and tuple(ck['model_state_dict']['upscore.weight'].shape) == (20, 20, 64, 64)
    assert len(ck['optim_state_dict']['param_groups']) == 2
    st = ck['optim_state_dict']['state']
    assert len(st) == 32 and all('exp_avg' in v for v in st.values())      # flat moments exported per parameter
    # seen-mask phase really trained only the head: losses finite and changing
    sl = [float(r.split(',')[2]) for r in srows[1:]]
    assert all(l == l for l in sl) and sl[0] != sl[-1]
    # test_all on the checkpoint (cfg 19 = test mode of cfg 18): full SZN inference path
    run = os.path.basename(log)
    train.main(['-c', '19', '-r', run, '-dir', d, '-n', 'smoke_test', '--synthetic', '2', '64', '64', '--workers', '0'])
    out = capsys.readouterr().out
    assert 'unseen mean_iu' in out and 'overall mean_iu' in out


def test_train_cli_fcn8s_arch(fast_tmp, capsys):
    """--arch fcn8s: the same CLI flow over the skip head (autograd path + fused per-tensor Adam: 17 Conv2d weight / bias pairs
    incl. score_pool3 / score_pool4), seen-mask phase, checkpoint, then test_all from the checkpoint"""
    d = fast_tmp
    train.main(['-c', '18', '-ve', '1', '-dir', d, '-n', 'f8', '--synthetic', '2', '64', '64', '--workers', '0', '--arch', 'fcn8s',
                '--precision', 'bf16'])
    log = glob.glob(os.path.join(d, 'logs', 'f8_CFG_18_*'))[0]
    rows = open(os.path.join(log, 'train_log.csv')).read().strip().split('\n')
    assert len(rows) == 1 + 2 and all(float(r.split(',')[2]) == float(r.split(',')[2]) for r in rows[1:])
    ck = torch.load(os.path.join(log, 'best'), map_location='cpu', weights_only=False)
    sd = ck['model_state_dict']
    assert ck['arch'] == 'FCN8s' and 'upscore.weight' not in sd
    assert tuple(sd['score_pool3.weight'].shape) == (20, 256, 1, 1) and tuple(sd['upscore8.weight'].shape) == (20, 20, 16, 16)
    assert len(ck['optim_state_dict']['state']) == 36                       # 18 Conv2d layers x (weight, bias)
    run = os.path.basename(log)
    train.main(['-c', '19', '-r', run, '-dir', d, '-n', 'f8_test', '--synthetic', '2', '64', '64', '--workers', '0', '--arch', 'fcn8s',
                '--precision', 'bf16'])
    out = capsys.readouterr().out
    assert 'unseen mean_iu' in out and 'overall mean_iu' in out
