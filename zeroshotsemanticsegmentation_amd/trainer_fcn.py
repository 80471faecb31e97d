"""trainer_fcn.py -- phase-1 trainer of the SZN path: FCN backbone + pixel-embedding (or softmax) head.

Same constructor / method surface and return contracts as /root/reference/trainer_fcn.py (Trainer.__init__ :21-81,
forward :83-120, forward_szn :123-147, train_epoch :149-180, validate :182-292, train :294-306), on top of the
HIP path: with an embedding loss 'cos' the hot loop runs engine.TrainStep (fused head, flat-buffer optimizer,
RCCL gradient all-reduce); other losses compose the same kernels through autograd.  Logging (CSV, optional
tensorboard writer, checkpoints with the reference's dict keys) is host-side and kept format-compatible;
JPEG tile visualisations (third-party `fcn` package in the reference) are out of scope and skipped.
"""
import datetime
import os
import os.path as osp
import shutil

import numpy as np
import torch

from . import engine as _engine
from . import utils

_DATA = osp.join(osp.dirname(osp.abspath(__file__)), "data")      # package data: .npy re-saves of the reference's pickles


def load_embeddings(dataset, dim):
    """K x E class-embedding matrix: `datasets/<ds>/embeddings/norm_embed_arr_<E>.pkl` relative to the CWD exactly
    like the reference (trainer_fcn.py:49), else the .npy re-saves shipped inside this package (data/)."""
    pkl = 'datasets/%s/embeddings/norm_embed_arr_%s' % (dataset, str(dim))
    if osp.exists(pkl + '.pkl'):
        return np.asarray(utils.load_obj(pkl), dtype=np.float32)
    npy = osp.join(_DATA, "embeddings_%s_%s.npy" % (dataset, str(dim)))
    if osp.exists(npy):
        return np.load(npy)
    raise IOError("no embedding matrix for dataset=%s dim=%s (looked for %s.pkl and %s)" % (dataset, dim, pkl, npy))


def _now():
    return datetime.datetime.now(datetime.timezone(datetime.timedelta(hours=-5)))      # 'US/Eastern' (no DST handling)


class _NullWriter(object):
    def add_scalar(self, *a, **k): pass
    def add_text(self, *a, **k): pass
    def add_image(self, *a, **k): pass


class Trainer(object):

    def __init__(self, cuda, model, optimizer, train_loader, val_loader, log_dir, dataset, max_epoch, tb_writer,
                 pixel_embeddings=None, loss_func=None, unseen=None, val_unseen=None, label_names=None,
                 forced_unseen=False, embed_arr=None, precision=torch.float32, fused_step=True, rank=0):
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
        self.pixel_embeddings = pixel_embeddings
        self.loss_func = loss_func
        self.unseen = list(unseen or [])            # all unseen classes (train_unseen + val_unseen)
        self.val_unseen = list(val_unseen or [])
        self.label_names = label_names
        self.forced_unseen = forced_unseen
        self.rank = rank

        self.epoch = 0
        self.iteration = 0
        self.best_mean_iu = 0
        self.n_class = len(self.train_loader.dataset.class_names)
        self.timestamp_start = _now()
        self.seen = [x for x in range(self.n_class) if x not in self.unseen]
        self.device = next(model.parameters()).device
        self.precision = precision
        self._step = None
        self._fused_step = fused_step
        self.verbose_val = os.environ.get("SZN_VERBOSE_VAL", "0") == "1"   # per-image prints cost a host sync each

        if self.pixel_embeddings:
            arr = np.asarray(embed_arr, dtype=np.float32) if embed_arr is not None else \
                load_embeddings(dataset, pixel_embeddings)
            self.embeddings = torch.from_numpy(arr).to(self.device).float()
            # zero-masked copies used for forced_unseen / full SZN testing (reference :56-64; built in float64)
            seen_arr, unseen_arr = np.zeros(arr.shape), np.zeros(arr.shape)
            seen_arr[self.seen, :] = arr[self.seen, :]
            unseen_arr[self.unseen, :] = arr[self.unseen, :]
            self.seen_embeddings = torch.from_numpy(seen_arr).to(self.device).float()
            self.unseen_embeddings = torch.from_numpy(unseen_arr).to(self.device).float()

        base = ['loss', 'pxl_acc', 'class_acc', 'mean_iu', 'fwavacc']
        self.train_log_headers = ['epoch', 'iteration'] + ['train/' + b for b in base] + ['elapsed_time']
        self.val_log_headers = ['epoch', 'iteration'] + ['val/' + b for b in base]
        if self.unseen:
            for grp in ('seen', 'unseen'):
                self.val_log_headers += ['val/%s/%s' % (grp, b) for b in base[1:]]
        self.val_log_headers += ['elapsed_time']
        if self.rank == 0:
            os.makedirs(self.log_dir, exist_ok=True)
            for fname, hdr in (('train_log.csv', self.train_log_headers), ('val_log.csv', self.val_log_headers)):
                if not osp.exists(osp.join(self.log_dir, fname)):
                    with open(osp.join(self.log_dir, fname), 'w') as f:
                        f.write(','.join(hdr) + '\n')

    # ---- forward passes ------------------------------------------------------------------------------------
    def _unpack(self, data, target):
        """targets are (label, label_embedding) tuples when embeddings are on (dataset contract, context_dataset.py:
        116-141); the dense per-pixel embedding is optional here: the label alone is enough (gathered on the GPU)."""
        target_embed = None
        if self.pixel_embeddings and isinstance(target, (tuple, list)):
            target, target_embed = target
        if data.dtype == torch.uint8:              # native dataset samples: raw RGB (n,h,w,3); BGR - mean on the GPU
            data = utils.image_to_device(data, self.device)
        else:
            data = data.to(self.device, non_blocking=True)
        target = target.to(self.device, non_blocking=True)
        if target_embed is not None:
            if target_embed.dim() == 4 and target_embed.is_floating_point():
                target_embed = target_embed.to(self.device, non_blocking=True)     # the reference's dense (n,E,h,w) volume
            else:
                target_embed = None            # label-only datasets: the row of the K x E matrix is gathered on the GPU
        return data, target, target_embed

    def _loss(self, score, target, target_embed):
        te = target_embed if target_embed is not None else (self.embeddings if self.pixel_embeddings else None)
        if self.loss_func == "cos":
            return utils.cosine_loss(score, target, te)
        if self.loss_func == "mse":
            return utils.mse_loss(score, target, te)
        if self.loss_func == "cross_entropy":
            return utils.cross_entropy2d(score, target, size_average=False)
        raise ValueError("unknown loss_func %r" % (self.loss_func,))

    def forward(self, data, target):
        """-> (score, loss, lbl_pred numpy int64 (n,h,w), lbl_true cpu tensor)   [reference :83-120]"""
        data, target, target_embed = self._unpack(data, target)
        score = self.model(data, mode='fcn')
        loss = self._loss(score, target, target_embed)
        if np.isnan(float(loss.item())):
            raise ValueError('loss is nan while training')
        if self.pixel_embeddings:
            if self.forced_unseen:
                lbl_pred = utils.infer_lbl_forced_unseen(score, target, self.seen_embeddings, self.unseen_embeddings,
                                                         self.unseen, self.cuda)
            else:
                lbl_pred = utils.infer_lbl(score, self.embeddings, self.cuda)
        else:
            lbl_pred = utils.channel_argmax(score).cpu().numpy()
        return score, loss, lbl_pred, target.detach().cpu()

    def forward_szn(self, data, target):
        """both heads + seen-mask-stitched inference   [reference :123-147]"""
        data, target, target_embed = self._unpack(data, target)
        fcn_score, seen_mask_score = self.model(data, mode='both')
        loss = self._loss(fcn_score, target, target_embed)
        lbl_pred = utils.infer_lbl_szn(fcn_score, seen_mask_score, self.seen_embeddings, self.unseen_embeddings, self.cuda)
        return fcn_score, loss, lbl_pred, target.detach().cpu()

    # ---- training --------------------------------------------------------------------------------------------
    def _fast_step(self):
        """engine.TrainStep when the configuration allows it: embedding cosine loss, no forced_unseen train metrics, and an
        optimizer with exactly the reference's two parameter groups (train.py:126-133: all Conv2d weights | all Conv2d
        biases).  Hyper-parameters are read per group from the optimizer object; any other wiring keeps the autograd path."""
        if self._step is not None or not (self._fused_step and self.pixel_embeddings and self.loss_func == "cos"):
            return self._step
        if self.forced_unseen:
            return None
        from .optim import FusedAdam, FusedSGD
        from .models import opt_layers
        _OPT_LAYERS = opt_layers(self.model)
        groups = self.optim.param_groups
        ws = [getattr(self.model, n).weight for n in _OPT_LAYERS]
        bs = [getattr(self.model, n).bias for n in _OPT_LAYERS]
        same = lambda a, b: len(a) == len(b) and {id(t) for t in a} == {id(t) for t in b}
        if len(groups) != 2 or not same(groups[0]['params'], ws) or not same(groups[1]['params'], bs):
            return None
        gw, gb = groups
        if isinstance(self.optim, (FusedAdam, torch.optim.Adam)):
            if gw.get('amsgrad') or gw['betas'] != gb['betas'] or gw['eps'] != gb['eps']:
                return None
            kw = dict(optimizer="adam", betas=tuple(gw['betas']), eps=gw['eps'], adam_weight_decay=gw.get('weight_decay', 0.0))
        elif isinstance(self.optim, (FusedSGD, torch.optim.SGD)):
            if gw.get('nesterov') or gw.get('dampening', 0) or gw.get('momentum', 0) != gb.get('momentum', 0):
                return None
            kw = dict(optimizer="sgd", momentum=gw.get('momentum', 0), weight_decay=gw.get('weight_decay', 0.0))
        else:
            return None
        self._step = _engine.TrainStep(self.model, self.embeddings, lr=gw['lr'], bias_lr=gb['lr'],
                                       bias_weight_decay=gb.get('weight_decay', 0.0), precision=self.precision,
                                       fused_head=True, keep_grads=os.environ.get("SZN_KEEP_GRADS", "0") == "1", **kw)
        # keep_grads=False (default; SZN_KEEP_GRADS=1 restores the reference's "`.grad` valid until zero_grad()"): the loop
        # calls zero_grad() next (train.py:170-175) and nothing reads the weight gradients in between, so the layers whose Adam
        # step rides in their weight-gradient kernel do not store theirs -- those `.grad`s are None, not stale
        self._step.import_optimizer_state(self.optim)       # resumed runs (train.py:135-136)
        return self._step

    def train_epoch(self):
        self.model.train()
        step = self._fast_step()
        if step is None and self.model._engine.dtype == torch.float16:
            # IEEE-half gradients need loss scaling, which only engine.TrainStep applies (d(coarse) is multiplied in fp32 before
            # it enters the 16-bit backward pass); the autograd paths (mse / cross_entropy losses, forced_unseen, non-reference
            # optimizer wiring) would push ~1e-7-sized activation gradients through fp16 unscaled and lose them silently
            raise RuntimeError("precision fp16 is only supported on the fused training step (embedding cosine loss, reference "
                               "optimizer wiring, no forced_unseen); use bf16 or fp32 for this configuration")
        for batch_idx, (data, target) in enumerate(self.train_loader):
            if step is not None:
                data, target, _ = self._unpack(data, target)
                step.hist.zero_()
                loss, pred = step.step(data, target)
                # one D2H per iteration: the loss and the K x K histogram the step already accumulated on the device
                # (the reference logs per-iteration metrics, trainer_fcn.py:164-174)
                packed = torch.cat([loss.reshape(1).double(), step.hist[0].reshape(-1).double()]).cpu().numpy()
                lossv = float(packed[0])
                if np.isnan(lossv):
                    raise ValueError('loss is nan while training')
                metrics = utils._hist_to_metrics(packed[1:].reshape(self.n_class, self.n_class))
                gsum = float('nan')                  # not read back on this path (a host sync per iteration for a debug print)
                ssum = float('nan')                  # the (B,E,H,W) score is never materialised on this path
            elif (hasattr(self.model, 'embed_loss') and self._fused_step and self.pixel_embeddings and self.loss_func == "cos"
                  and not self.forced_unseen and self.embeddings.shape[0] <= 256):
                # FCN8s: autograd chain with the fused-from-1/8-map head (no (n,E,h,w) score), per-tensor fused optimizer
                data, target, _ = self._unpack(data, target)
                loss, pred = self.model.embed_loss(data, self.embeddings, target)
                self.optim.zero_grad()
                loss.backward()
                _engine.allreduce_param_grads([p for g in self.optim.param_groups for p in g['params']])
                self.optim.step()
                hist = utils.confusion_hist_device(target, pred, self.n_class)[0]
                packed = torch.cat([loss.detach().reshape(1).double(), hist.reshape(-1).double()]).cpu().numpy()
                lossv = float(packed[0])
                if np.isnan(lossv):
                    raise ValueError('loss is nan while training')
                metrics = utils._hist_to_metrics(packed[1:].reshape(self.n_class, self.n_class))
                gsum = ssum = float('nan')
            else:
                score, loss, lbl_pred, lbl_true = self.forward(data, target)
                self.optim.zero_grad()
                loss.backward()
                # data parallel: mean of the rank gradients before the update (the fused path does this inside TrainStep)
                _engine.allreduce_param_grads([p for g in self.optim.param_groups for p in g['params']])
                self.optim.step()
                lossv = float(loss.item())
                metrics = utils.label_accuracy_score(lbl_true.numpy(), lbl_pred, self.n_class)
                gsum = float(self.model.score_fr.weight.grad.sum().item())
                ssum = float(score.sum().item())
            if self.rank == 0:
                print("FCN Train Epoch {:<5} | Iteration {:<5} | Loss {:5.5f} | score_fr grad sum {:15.0f} | "
                      "upscore grad sum {:15.0f} | score sum {:10.5f}".format(int(self.epoch), int(batch_idx), lossv, gsum,
                                                                               0.0, ssum))
                with open(osp.join(self.log_dir, 'train_log.csv'), 'a') as f:
                    elapsed = (_now() - self.timestamp_start).total_seconds()
                    f.write(','.join(map(str, [self.epoch, self.iteration, lossv] + list(metrics) + [elapsed])) + '\n')
                for name, v in zip(['loss', 'pxl_acc', 'class_acc', 'mean_iu', 'fwavacc'], [lossv] + list(metrics)):
                    self.tb_writer.add_scalar('fcn/train/' + name, v, self.iteration)
            self.iteration += 1

    def _predict_device(self, data, target, szn):
        """forward + loss + class assignment with everything left on the GPU -> (score, loss 0-dim, pred (n,h,w), target)"""
        data, target, target_embed = self._unpack(data, target)
        if (self.pixel_embeddings and self.loss_func == "cos" and not szn and not self.forced_unseen and target_embed is None
                and not self.verbose_val and self.embeddings.shape[0] <= 256):
            # plain embedding inference: loss + class assignment straight from the 1/32 map (no (n,E,h,w) score in HBM)
            loss, pred = self.model.embed_predict(data, self.embeddings, target)
            return None, loss, pred, target
        if szn:
            score, seen_mask_score = self.model(data, mode='both')
        else:
            score = self.model(data, mode='fcn')
        loss = self._loss(score, target, target_embed)
        if not self.pixel_embeddings:
            pred = utils.channel_argmax(score)
        elif szn:
            full = self.seen_embeddings + self.unseen_embeddings
            pred = utils.infer_lbl_device(score, full, mode=1, unseen=self.unseen, seenmask=seen_mask_score)
        elif self.forced_unseen:
            full = self.seen_embeddings + self.unseen_embeddings
            pred = utils.infer_lbl_device(score, full, mode=1, unseen=self.unseen, target=target)
        else:
            pred = utils.infer_lbl_device(score, self.embeddings)
        return score, loss, pred, target

    def validate(self, both_fcn_and_seenmask=False):
        """reference :182-292.  The {all, seen, unseen} K x K histograms and the loss sum are accumulated on the GPU
        (szn_confusion_hist on the device prediction) and read back ONCE per epoch; under data parallelism the validation
        images are split across the ranks and the histograms / loss sums are all-reduced before the metrics."""
        import torch.distributed as dist
        self.model.eval()
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        hist = torch.zeros(3, self.n_class, self.n_class, dtype=torch.int64, device=self.device)
        acc = torch.zeros(2, dtype=torch.float64, device=self.device)          # loss sum, image-batch count
        with torch.no_grad():
            for batch_idx, (data, target) in enumerate(self.val_loader):
                # a loader that is not already sharded per rank (train.py shards it: each rank decodes only its own images)
                if world > 1 and not getattr(self.val_loader, 'szn_sharded', False) and batch_idx % world != self.rank:
                    continue
                score, loss, pred, tgt = self._predict_device(data, target, both_fcn_and_seenmask)
                acc[0] += loss.double()
                acc[1] += 1
                utils.confusion_hist_device(tgt, pred, self.n_class, self.val_unseen if self.unseen else None, hist)
                if self.verbose_val and self.rank == 0:
                    print("Test Epoch {:<5} | Iteration {:<5} | Loss {:5.5f} | Score Sum {:10.5f}".format(
                        int(self.epoch), int(batch_idx), float(loss.item()), float(score.sum().item())))
        if world > 1:
            dist.all_reduce(hist)
            dist.all_reduce(acc)
        h = hist.cpu().numpy()
        accn = acc.cpu().numpy()
        if np.isnan(accn[0]):
            raise ValueError('loss is nan while validating')
        val_loss = float(accn[0])
        n_batches = max(int(accn[1]), 1)
        metrics = utils._hist_to_metrics(h[0])
        seen_metrics = unseen_metrics = None
        if self.unseen:
            seen_metrics, unseen_metrics = utils._hist_to_metrics(h[1]), utils._hist_to_metrics(h[2])
        val_loss /= n_batches                           # averaged over images like the reference (:246)
        names = ['pxl_acc', 'class_acc', 'mean_iu', 'fwavacc']
        if self.rank == 0:
            with open(osp.join(self.log_dir, 'val_log.csv'), 'a') as f:
                row = [self.epoch, self.iteration, val_loss] + list(metrics)
                if self.unseen:
                    row += list(seen_metrics) + list(unseen_metrics)
                row += [_now() - self.timestamp_start]
                f.write(','.join(map(str, row)) + '\n')
            self.tb_writer.add_scalar('fcn/val/loss', val_loss, self.epoch)
            for grp, ms in (('', metrics), ('seen/', seen_metrics), ('unseen/', unseen_metrics)):
                if ms is None:
                    continue
                for n, v in zip(names, ms):
                    self.tb_writer.add_scalar('fcn/val/%s%s' % (grp, n), v, self.epoch)
                    print('%s%s: %.3f' % ((grp.replace('/', ' ') or 'overall '), n, v))
        mean_iu = metrics[2]
        is_best = mean_iu > self.best_mean_iu
        if is_best:
            self.best_mean_iu = mean_iu
        if self._step is not None:
            self._step.gather_masters()          # sharded optimizer: a collective -- every rank, before rank 0 reads the parameters
        if self.rank == 0:
            if self._step is not None:
                self._step.export_optimizer_state(self.optim)   # flat moments -> per-parameter torch optimizer state
            torch.save({
                'epoch': self.epoch,
                'iteration': self.iteration,
                'arch': self.model.__class__.__name__,
                'optim_state_dict': self.optim.state_dict(),
                'model_state_dict': self.model.state_dict(),
                'best_mean_iu': self.best_mean_iu,
            }, osp.join(self.log_dir, 'checkpoint'))
            if is_best:
                shutil.copy(osp.join(self.log_dir, 'checkpoint'), osp.join(self.log_dir, 'best'))
        return metrics

    def train(self):
        for epoch in range(self.max_epoch):
            self.epoch = epoch
            if hasattr(getattr(self.train_loader, 'sampler', None), 'set_epoch'):
                self.train_loader.sampler.set_epoch(epoch)       # DistributedSampler: a new shuffle per epoch
            self.train_epoch()
            self.validate()
            # early stop once as many images were seen as in 50 epochs without zero-shot (reference :300-306)
            cur_iter = self.epoch * len(self.train_loader)
            if (self.dataset == 'pascal' and cur_iter > 425000) or (self.dataset == 'context' and cur_iter > 247000):
                break
