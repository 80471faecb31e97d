"""trainer_seenmask.py -- phase-2 trainer: the binary seen/unseen mask head on a frozen backbone.

Same surface as /root/reference/trainer_seenmask.py (Trainer.__init__ :21-48, forward :50-70, train_epoch :72-102,
validate :104-166, train :168-172).  Target construction keeps the reference's rule: a pixel is "seen" (1) iff its
label is a seen class, so unlabelled pixels (-1) become 0 = "unseen" and are NOT ignored (:55-56).
"""
import datetime
import os
import os.path as osp

import numpy as np
import torch

from . import utils


def _now():
    return datetime.datetime.now(datetime.timezone(datetime.timedelta(hours=-5)))


class _NullWriter(object):
    def add_scalar(self, *a, **k): pass
    def add_image(self, *a, **k): pass


class Trainer(object):

    def __init__(self, cuda, model, optimizer, train_loader, val_loader, log_dir, dataset, max_epoch, tb_writer,
                 checkpoint, unseen, rank=0, fused_step=True):
        if not cuda:
            raise RuntimeError("this implementation runs on the GPU only (cuda=False has no CPU fallback)")
        self.cuda = cuda
        self.model = model
        self.optim = optimizer
        self.train_loader = train_loader
        self.val_loader = val_loader
        self.log_dir = log_dir
        self.dataset = dataset
        self.max_epoch = max_epoch
        self.tb_writer = tb_writer if tb_writer is not None else _NullWriter()
        self.checkpoint = checkpoint if checkpoint is not None else {}
        self.unseen = list(unseen)
        self.rank = rank
        self._step = None
        self._fused_step = fused_step

        self.epoch = 0
        self.iteration = 0
        self.best_mean_iu = 0
        self.n_class = len(self.train_loader.dataset.class_names)
        self.timestamp_start = _now()
        self.device = next(model.parameters()).device
        seen = [x for x in range(self.n_class) if x not in self.unseen]
        self._seen_lut = torch.zeros(self.n_class + 1, dtype=torch.int64, device=self.device)
        self._seen_lut[torch.tensor(seen, dtype=torch.int64, device=self.device)] = 1

        base = ['loss', 'pxl_acc', 'class_acc', 'mean_iu', 'fwavacc']
        self.train_log_headers = ['epoch', 'iteration'] + ['train/' + b for b in base] + ['elapsed_time']
        self.val_log_headers = ['epoch', 'iteration'] + ['val/' + b for b in base] + ['elapsed_time']
        if self.rank == 0:
            os.makedirs(self.log_dir, exist_ok=True)
            for fname, hdr in (('seenmask_train_log.csv', self.train_log_headers),
                               ('seenmask_val_log.csv', self.val_log_headers)):
                if not osp.exists(osp.join(self.log_dir, fname)):
                    with open(osp.join(self.log_dir, fname), 'w') as f:
                        f.write(','.join(hdr) + '\n')

    def binary_target(self, target):
        """np.in1d(target, seen) on the device: labels outside [0, n_class) (incl. -1) map to 0 (trainer_seenmask.py:55-56);
        batch padding (datasets.PAD_LABEL = -2, written by pad_collate around the smaller images of a ragged batch) stays
        negative, so cross_entropy2d's `target >= 0` mask and the metrics leave it out"""
        t = target.to(self.device)
        idx = torch.where((t >= 0) & (t < self.n_class), t, torch.full_like(t, self.n_class))
        b = self._seen_lut[idx]
        return torch.where(t < -1, torch.full_like(b, -1), b)

    def _forward_device(self, data, target):
        """-> (score, loss, pred (n,h,w) int64 device tensor, binary target device tensor)"""
        if isinstance(target, (tuple, list)):
            target = target[0]
        target = self.binary_target(target)
        if data.dtype == torch.uint8:              # native dataset samples: raw RGB (n,h,w,3); BGR - mean on the GPU
            data = utils.image_to_device(data, self.device)
        else:
            data = data.to(self.device, non_blocking=True)
        score = self.model(data, mode='seenmask')
        loss = utils.cross_entropy2d(score, target, size_average=True)
        return score, loss, utils.channel_argmax(score), target

    def forward(self, data, target):
        """-> (score, loss, lbl_pred numpy int64 (n,h,w), lbl_true cpu tensor)   [reference :50-70]"""
        score, loss, pred, target = self._forward_device(data, target)
        return score, loss, pred.cpu().numpy(), target.detach().cpu()

    def _metrics_device(self, loss, target, pred):
        """loss + the K x K histogram of one batch in ONE device-to-host copy -> (loss value, metrics)"""
        hist = utils.confusion_hist_device(target, pred, self.n_class)
        packed = torch.cat([loss.detach().reshape(1).double(), hist[0].reshape(-1).double()]).cpu().numpy()
        return float(packed[0]), utils._hist_to_metrics(packed[1:].reshape(self.n_class, self.n_class))

    def _fast_step(self):
        """engine.SeenmaskStep when the optimizer is the reference wiring (train.py:170-175: ONE Adam group holding
        seenmask_score.weight / .bias and seenmask_upscore.weight) and the class count fits the 64-bit seen mask; any other
        wiring keeps the autograd path below"""
        if self._step is not None or not self._fused_step:
            return self._step
        from .engine import SeenmaskStep
        from .optim import FusedAdam
        m = self.model
        want = {id(m.seenmask_score.weight), id(m.seenmask_score.bias), id(m.seenmask_upscore.weight)}
        groups = self.optim.param_groups
        if (not isinstance(self.optim, (FusedAdam, torch.optim.Adam)) or len(groups) != 1 or self.n_class > 64
                or {id(p) for p in groups[0]['params']} != want or groups[0].get('amsgrad')):
            return None
        g = groups[0]
        self._step = SeenmaskStep(m, self.n_class, self.unseen, lr=g['lr'], betas=tuple(g['betas']), eps=g['eps'],
                                  weight_decay=g.get('weight_decay', 0.0))
        return self._step

    def train_epoch(self):
        from .engine import allreduce_param_grads
        self.model.train()
        step = self._fast_step()
        if step is None and self.model._engine.dtype == torch.float16:
            # the autograd path rounds the unscaled ~1e-5-sized d(coarse) to IEEE half before the weight gradient (subnormals);
            # engine.SeenmaskStep keeps it in fp32
            raise RuntimeError("precision fp16 in phase 2 needs the fused seen-mask step (the reference's single Adam group over "
                               "seenmask_score / seenmask_upscore); use bf16 or fp32 with this optimizer")
        for batch_idx, (data, target) in enumerate(self.train_loader):
            if step is not None:
                if isinstance(target, (tuple, list)):
                    target = target[0]
                data = utils.image_to_device(data, self.device) if data.dtype == torch.uint8 else data.to(self.device, non_blocking=True)
                step.conf.zero_()
                # hyperparameters are the optimizer object's, read every step (an LR schedule or a manual decay on
                # optim.param_groups keeps working on the fused path)
                g = self.optim.param_groups[0]
                step.lr, step.betas, step.eps, step.wd = g['lr'], tuple(g['betas']), g['eps'], g.get('weight_decay', 0.0)
                loss, pred = step.step(data, target.to(self.device, non_blocking=True).long().contiguous())
                # one D2H per iteration: the loss and the 2 x 2 confusion counts the step accumulated on the device
                packed = torch.cat([loss.reshape(1).double(), step.conf.double()]).cpu().numpy()
                lossv = float(packed[0])
                hist = np.zeros((self.n_class, self.n_class))
                hist[:2, :2] = packed[1:].reshape(2, 2)
                metrics = utils._hist_to_metrics(hist)
            else:
                score, loss, pred, tgt = self._forward_device(data, target)
                self.optim.zero_grad()
                loss.backward()
                # data parallel phase 2: one small RCCL all-reduce of the 24,578 trainable gradient elements
                allreduce_param_grads([p for g in self.optim.param_groups for p in g['params']])
                self.optim.step()
                lossv, metrics = self._metrics_device(loss, tgt, pred)
            if np.isnan(lossv):
                raise ValueError('loss is nan while training')
            if self.rank == 0:
                print("Seenmask Train Epoch {:<5} | Iteration {:<5} | Loss {:5.5f}".format(int(self.epoch), int(batch_idx), lossv))
                with open(osp.join(self.log_dir, 'seenmask_train_log.csv'), 'a') as f:
                    elapsed = (_now() - self.timestamp_start).total_seconds()
                    f.write(','.join(map(str, [self.epoch, self.iteration, lossv] + list(metrics) + [elapsed])) + '\n')
                for name, v in zip(['loss', 'pxl_acc', 'class_acc', 'mean_iu', 'fwavacc'], [lossv] + list(metrics)):
                    self.tb_writer.add_scalar('seenmask/train/' + name, v, self.iteration)
            self.iteration += 1
        if step is not None:
            # the fused step owns the Adam moments (flat buffers): expose them as per-parameter state of the optimizer object
            # (views, torch.optim.Adam layout), so that checkpointing / inspecting `optim` after phase 2 sees what autograd +
            # optim.step() would have left there.  The three head Parameters stay views of the step's flat master buffer (the
            # update is written in place; seenmask_score.weight is a permuted, non-contiguous (2,F,1,1) view of its (2,F) rows).
            step.export_optimizer_state(self.optim)

    def validate(self):
        """reference :104-166; histogram and loss sum accumulated on the GPU, one read-back per epoch; validation images are
        split across the data-parallel ranks and the sums all-reduced"""
        import torch.distributed as dist
        self.model.eval()
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        hist = torch.zeros(3, self.n_class, self.n_class, dtype=torch.int64, device=self.device)
        acc = torch.zeros(2, dtype=torch.float64, device=self.device)
        with torch.no_grad():
            for batch_idx, (data, target) in enumerate(self.val_loader):
                # a loader that is not already sharded per rank (train.py shards it: each rank decodes only its own images)
                if world > 1 and not getattr(self.val_loader, 'szn_sharded', False) and batch_idx % world != self.rank:
                    continue
                if self._fused_step and self.n_class <= 256:
                    # loss + prediction straight from the 1/32 map (models.FCN32s.seenmask_predict): no (n,2,h,w) score
                    if isinstance(target, (tuple, list)):
                        target = target[0]
                    data = utils.image_to_device(data, self.device) if data.dtype == torch.uint8 else data.to(self.device, non_blocking=True)
                    loss, pred = self.model.seenmask_predict(data, target.to(self.device), self.n_class, self.unseen)
                    tgt = self.binary_target(target)
                else:
                    score, loss, pred, tgt = self._forward_device(data, target)
                acc[0] += loss.double()
                acc[1] += 1
                utils.confusion_hist_device(tgt, pred, self.n_class, None, hist)
        if world > 1:
            dist.all_reduce(hist)
            dist.all_reduce(acc)
        accn = acc.cpu().numpy()
        metrics = utils._hist_to_metrics(hist[0].cpu().numpy())
        val_loss = float(accn[0]) / max(int(accn[1]), 1)
        if self.rank == 0:
            with open(osp.join(self.log_dir, 'seenmask_val_log.csv'), 'a') as f:
                row = [self.epoch, self.iteration, val_loss] + list(metrics) + [_now() - self.timestamp_start]
                f.write(','.join(map(str, row)) + '\n')
            self.tb_writer.add_scalar('seenmask/val/loss', val_loss, self.epoch)
            for n, v in zip(['pxl_acc', 'class_acc', 'mean_iu', 'fwavacc'], metrics):
                self.tb_writer.add_scalar('seenmask/val/' + n, v, self.epoch)
                print('%s: %.3f' % (n, v))
        if metrics[2] > self.best_mean_iu:
            self.best_mean_iu = metrics[2]
        if self.rank == 0:
            # the phase-1 checkpoint dict with the updated weights overwrites 'best' every epoch (reference :165-166)
            self.checkpoint['model_state_dict'] = self.model.state_dict()
            torch.save(self.checkpoint, osp.join(self.log_dir, 'best'))
        return metrics

    def train(self):
        for epoch in range(self.max_epoch):
            self.epoch = epoch
            if hasattr(getattr(self.train_loader, 'sampler', None), 'set_epoch'):
                self.train_loader.sampler.set_epoch(epoch)       # DistributedSampler: a new shuffle per epoch
            self.train_epoch()
            self.validate()
