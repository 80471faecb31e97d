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
                 checkpoint, unseen, rank=0):
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
        """np.in1d(target, seen) on the device: labels outside [0, n_class) (incl. -1) map to 0"""
        t = target.to(self.device)
        idx = torch.where((t >= 0) & (t < self.n_class), t, torch.full_like(t, self.n_class))
        return self._seen_lut[idx]

    def forward(self, data, target):
        if isinstance(target, (tuple, list)):
            target = target[0]
        target = self.binary_target(target)
        data = data.to(self.device, non_blocking=True)
        score = self.model(data, mode='seenmask')
        loss = utils.cross_entropy2d(score, target, size_average=True)
        lbl_pred = utils.channel_argmax(score).cpu().numpy()
        return score, loss, lbl_pred, target.detach().cpu()

    def _allreduce_grads(self):
        """data parallel phase 2: one small RCCL all-reduce of the 24,578 trainable gradient elements"""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        ps = [p for g in self.optim.param_groups for p in g['params'] if p.grad is not None]
        flat = torch.cat([p.grad.reshape(-1) for p in ps])
        dist.all_reduce(flat)
        flat /= dist.get_world_size()
        off = 0
        for p in ps:
            p.grad.copy_(flat[off:off + p.numel()].view_as(p.grad))
            off += p.numel()

    def train_epoch(self):
        self.model.train()
        for batch_idx, (data, target) in enumerate(self.train_loader):
            score, loss, lbl_pred, lbl_true = self.forward(data, target)
            self.optim.zero_grad()
            loss.backward()
            self._allreduce_grads()
            self.optim.step()
            lossv = float(loss.item())
            metrics = utils.label_accuracy_score(lbl_true.numpy(), lbl_pred, self.n_class)
            if self.rank == 0:
                print("Seenmask Train Epoch {:<5} | Iteration {:<5} | Loss {:5.5f} | seenmask_score grad sum {:7.8f} | "
                      "seenmask_upscore grad sum {:7.8f} | score sum {:10.5f}".format(
                          int(self.epoch), int(batch_idx), lossv, float(self.model.seenmask_score.weight.grad.sum().item()),
                          float(self.model.seenmask_upscore.weight.grad.sum().item()), float(score.sum().item())))
                with open(osp.join(self.log_dir, 'seenmask_train_log.csv'), 'a') as f:
                    elapsed = (_now() - self.timestamp_start).total_seconds()
                    f.write(','.join(map(str, [self.epoch, self.iteration, lossv] + list(metrics) + [elapsed])) + '\n')
                for name, v in zip(['loss', 'pxl_acc', 'class_acc', 'mean_iu', 'fwavacc'], [lossv] + list(metrics)):
                    self.tb_writer.add_scalar('seenmask/train/' + name, v, self.iteration)
            self.iteration += 1

    def validate(self):
        self.model.eval()
        val_loss = 0
        lbl_trues, lbl_preds = [], []
        with torch.no_grad():
            for batch_idx, (data, target) in enumerate(self.val_loader):
                score, loss, lbl_pred, lbl_true = self.forward(data, target)
                val_loss += float(loss.item())
                if self.rank == 0:
                    print("Seenmask Test Epoch {:<5} | Iteration {:<5} | Loss {:5.5f} | Score Sum {:10.5f}".format(
                        int(self.epoch), int(batch_idx), float(loss.item()), float(score.sum().item())))
                for i in range(lbl_pred.shape[0]):
                    lbl_trues.append(lbl_true[i].numpy())
                    lbl_preds.append(lbl_pred[i])
        metrics = utils.label_accuracy_score(lbl_trues, lbl_preds, self.n_class)
        val_loss /= max(len(self.val_loader), 1)
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
            self.train_epoch()
            self.validate()
