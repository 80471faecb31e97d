#!/usr/bin/env python
"""train.py -- CLI entry point of the SZN training path (mirrors /root/reference/train.py).

Same flags (-n -g -c -dir -tb -m -d -tu -vu -e -ve -lr -loss -o -se -slr -oh -fu -r; reference :20-42), same
configuration merge / validation (:202-251), same optimizer wiring (two parameter groups, :126-133,302-331) and
two-phase orchestration (FCN phase, then seen-mask phase with the backbone frozen, :161-194).

Added for this implementation (none change the reference flags):
  --synthetic N H W    deterministic synthetic dataset (no PASCAL data / network here); without it the PASCAL-VOC /
                       PASCAL-Context readers of datasets.py load <data_dir> in the reference's layout
  --batch-size B       images per GPU per step (reference: 1)
  --precision fp32|bf16
  --init synthetic|vgg path handling: without the caffe VGG16 file the backbone starts from synth weights
  torchrun: RANK / LOCAL_RANK / WORLD_SIZE are honoured (one process per GPU, RCCL gradient all-reduce).
"""
import argparse
import datetime
import os
import os.path as osp

import torch
import torch.nn as nn
import yaml

from . import engine as _engine
from . import models, trainer_fcn, trainer_seenmask
from .configs import configurations
from .optim import FusedAdam, FusedSGD
from .synthetic_dataset import SyntheticSegmentation


def build_parser():
    p = argparse.ArgumentParser()
    # default value args
    p.add_argument('-n', '--name', type=str, default=None, help='name of checkpoint folder')
    p.add_argument('-g', '--gpu', type=int, default=0, help='gpu number')
    p.add_argument('-c', '--config', type=int, default=1, choices=configurations.keys())
    p.add_argument('-dir', '--data_dir', type=str, default='data', help='path for storing dataset, logs, and models')
    p.add_argument('-tb', '--tb_dir', type=str, default=None, help='path to tensorboard directory (optional)')
    # override cfg args
    p.add_argument('-m', '--mode', type=str, choices=['train', 'test_fcn', 'test_all'])
    p.add_argument('-d', '--dataset', type=str, choices=['pascal', 'context'])
    p.add_argument('-tu', '--train_unseen', type=str, help='comma separated zero-shot train-split unseen classes')
    p.add_argument('-vu', '--val_unseen', type=str, help='comma separated zero-shot val-split unseen classes')
    p.add_argument('-e', '--embed_dim', type=int, choices=[2, 5, 10, 20, 21, 50, 100, 200, 300])
    p.add_argument('-ve', '--fcn_epochs', type=int)
    p.add_argument('-lr', '--fcn_learning_rate', type=float)
    p.add_argument('-loss', '--fcn_loss', type=str, choices=['cos', 'mse', 'cross_entropy'])
    p.add_argument('-o', '--fcn_optim', type=str, choices=['sgd', 'adam'])
    p.add_argument('-se', '--seenmask_epochs', type=int)
    p.add_argument('-slr', '--seenmask_learning_rate', type=float)
    # optional cfg args
    p.add_argument('-oh', '--one_hot_embed', action='store_true')
    p.add_argument('-fu', '--forced_unseen', action='store_true')
    p.add_argument('-r', '--resume', type=str, help='fcn model checkpoint path')
    # additions
    p.add_argument('--synthetic', type=int, nargs=3, metavar=('N', 'H', 'W'), help='synthetic dataset: N images of HxW')
    p.add_argument('--batch-size', type=int, default=1)
    p.add_argument('--precision', choices=['fp32', 'bf16', 'fp16'], default='fp32')
    p.add_argument('--arch', choices=['fcn32s', 'fcn8s'], default='fcn32s',
                   help="fcn32s = the reference's only backbone (train.py:103,105); fcn8s = the public FCN8s skip head "
                        "(not in the reference: BASELINE north_star wording, parity unpinned)")
    p.add_argument('--workers', type=int, default=2, help='DataLoader worker processes per loader (0 = load in the main process)')
    return p


def update_cfg_with_args(cfg, args):
    """reference :202-230 (truthiness rules kept: 0 cannot override; -se is parsed but never applied)"""
    cfg = dict(cfg)
    for key, val in (('mode', args.mode), ('dataset', args.dataset), ('embed_dim', args.embed_dim),
                     ('fcn_epochs', args.fcn_epochs), ('fcn_lr', args.fcn_learning_rate), ('fcn_loss', args.fcn_loss),
                     ('fcn_optim', args.fcn_optim), ('seenmask_lr', args.seenmask_learning_rate)):
        if val:
            cfg[key] = val
    for key, val in (('train_unseen', args.train_unseen), ('val_unseen', args.val_unseen)):
        if val:
            cfg[key] = [int(item) for item in val.split(',')]
    cfg['one_hot_embed'] = args.one_hot_embed if args.one_hot_embed else cfg.get('one_hot_embed')
    cfg['forced_unseen'] = args.forced_unseen if args.forced_unseen else cfg.get('forced_unseen')
    cfg['load_fcn_path'] = args.resume if args.resume else cfg.get('load_fcn_path')
    return cfg


# configuration sanity rules (the reference's checks, train.py:232-251, with its messages: the CLI contract): (broken?, message)
_ONE_HOT_WIDTH = {"pascal": 21, "context": 33}      # classes of the dataset = width of a one-hot class embedding
_CFG_RULES = (
    (lambda c: c['one_hot_embed'] and c['embed_dim'] != _ONE_HOT_WIDTH.get(c['dataset'], c['embed_dim']),
     'joint-embedding space must be size of one-hot embedding space'),
    (lambda c: not c['load_fcn_path'] and (c['mode'] in ('test_fcn', 'test_all') or c['fcn_epochs'] < 1),
     'must load model path via -r flag for test mode'),
    (lambda c: c['seenmask_epochs'] > 0 and not c['train_unseen'],
     "can't train the seenmask classifier without train_unseen specified"),
    (lambda c: c['embed_dim'] == 0 and c['fcn_loss'] in ('cos', 'mse'),
     "invalid loss function because pixel embedding dimensionality not defined"),
)


def validate_cfg(cfg):
    for broken, message in _CFG_RULES:
        if broken(cfg):
            raise Exception(message)


def get_log_dir(model_name, cfg_num, cfg, data_dir, now=None):
    """<data_dir>/logs/<name>CFG_<n>_<KEY>_<value>_..._TIME_<stamp>_ (reference :253-286)"""
    os.makedirs(data_dir, exist_ok=True)
    name = ('%s_' % model_name) if model_name else ''
    name += "CFG_%d_" % int(cfg_num)
    for k, v in cfg.items():
        if k in ['one_hot_embed', 'forced_unseen'] and not v:
            continue
        if k == 'load_fcn_path':
            continue
        if k in ['train_unseen', 'val_unseen']:
            name += '%s_%s_' % (k.upper(), str(bool(v)))
        else:
            name += '%s_%s_' % (k.upper(), str(v))
    now = now or datetime.datetime.now(datetime.timezone(datetime.timedelta(hours=-5)))
    name += 'TIME_%s_' % now.strftime('%Y%m%d-%H%M%S')
    log_dir = osp.join(data_dir, 'logs', name)
    os.makedirs(log_dir, exist_ok=True)
    return log_dir


def output_cfg(cfg, log_dir, writer):
    for k, v in cfg.items():
        print(k, v)
    with open(osp.join(log_dir, 'config.yaml'), 'w') as f:
        yaml.safe_dump(cfg, f, default_flow_style=False)
    if writer is not None:
        writer.add_text("cfg", '\n'.join(['%s: %s' % (k, str(v)) for k, v in cfg.items()]))


def get_parameters(model, bias=False, seenmask=False):
    """reference :302-331: Conv2d weights | Conv2d biases (seenmask layers excluded); ConvTranspose2d weights are the
    frozen bilinear kernels and yield nothing; unknown module types raise."""
    if seenmask:
        for p in model.seenmask_score.parameters():
            yield p
        for p in model.seenmask_upscore.parameters():
            yield p
        return
    skipped = (nn.ReLU, nn.MaxPool2d, nn.Dropout2d, nn.Sequential, models.FCN32s)
    for name, m in model.named_modules():
        if name in ['seenmask_score', 'seenmask_upscore']:
            continue
        if isinstance(m, nn.Conv2d):
            yield m.bias if bias else m.weight
        elif isinstance(m, nn.ConvTranspose2d):
            if bias:
                assert m.bias is None
        elif isinstance(m, skipped):
            continue
        else:
            raise ValueError('Unexpected module: %s' % str(m))


def make_fcn_optimizer(model, cfg):
    """two parameter groups: conv weights, conv biases at 2x lr (SGD: momentum .99, wd 5e-4 on weights only) :126-133"""
    if cfg['fcn_optim'] == "sgd":
        params = [{'params': list(get_parameters(model, bias=False))},
                  {'params': list(get_parameters(model, bias=True)), 'lr': cfg['fcn_lr'] * 2, 'weight_decay': 0}]
        return FusedSGD(params, lr=cfg['fcn_lr'], momentum=.99, weight_decay=0.0005)
    params = [{'params': list(get_parameters(model, bias=False))},
              {'params': list(get_parameters(model, bias=True)), 'lr': cfg['fcn_lr'] * 2}]
    return FusedAdam(params, lr=cfg['fcn_lr'])


def freeze_for_seenmask(model):
    """phase 2: everything frozen except seenmask_score (w, b) and seenmask_upscore (w) (reference :166-171)"""
    for param in model.parameters():
        param.requires_grad = False
    for p in model.seenmask_score.parameters():
        p.requires_grad = True
    for p in model.seenmask_upscore.parameters():
        p.requires_grad = True


def main(argv=None):
    args = build_parser().parse_args(argv)
    cfg = update_cfg_with_args(configurations[args.config], args)
    validate_cfg(cfg)
    if args.precision == 'fp16' and cfg['mode'] == 'train' and cfg['fcn_epochs'] > 0 and \
            (cfg['fcn_loss'] != 'cos' or not cfg['embed_dim'] or cfg['forced_unseen']):
        # loss scaling lives in the fused steps only (engine.TrainStep / SeenmaskStep); see trainer_fcn.Trainer.train_epoch
        raise Exception("--precision fp16 needs the fused training step: embedding configuration with fcn_loss 'cos' and no "
                        "forced_unseen (got loss %r, embed_dim %r, forced_unseen %r); use bf16 or fp32"
                        % (cfg['fcn_loss'], cfg['embed_dim'], cfg['forced_unseen']))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(args.gpu)))
    one_gpu = os.environ.get("SZN_TEST_ONE_GPU") == "1"       # test hook: every rank on device 0 over gloo (1-GPU boxes)
    if one_gpu:
        local_rank = 0
    if not torch.cuda.is_available():
        raise RuntimeError("no GPU visible: this implementation has no CPU path")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            _engine.init_process_group("nccl", device)      # RCCL on a high-priority stream (engine.init_process_group)
    torch.manual_seed(1337)
    torch.cuda.manual_seed(1337)

    # one log directory for the whole job: rank 0 stamps it (wall clock), the others receive the name
    log_dir = get_log_dir(args.name, args.config, cfg, args.data_dir) if rank == 0 else None
    if world > 1:
        box = [log_dir]
        dist.broadcast_object_list(box, src=0)
        log_dir = box[0]
    tb_writer = None
    if args.tb_dir and rank == 0:
        try:
            from tensorboardX import SummaryWriter
            tb_writer = SummaryWriter(osp.join(args.tb_dir, log_dir.split('/')[-1]))
        except ImportError:
            print("tensorboardX not installed: tensorboard logging disabled")
    if rank == 0:
        output_cfg(cfg, log_dir, tb_writer)

    # 1. dataset
    all_unseen = cfg['train_unseen'] + cfg['val_unseen']
    collate = None
    if args.synthetic:
        n_img, H, W = args.synthetic
        n_class = 21 if cfg['dataset'] == 'pascal' else 33
        mk = lambda split, unseen, n: SyntheticSegmentation(split=split, n_images=n, size=(H, W), n_class=n_class,
                                                             embed_dim=cfg['embed_dim'], unseen=unseen, seed=1337)
        # splits as in the reference (pascal_dataset.py:62-74, context_dataset.py:75-94): 'train' drops every image that
        # contains a val_unseen class, 'train_seen' additionally drops the train_unseen classes, 'val' keeps everything
        train_dataset = mk('train', cfg['val_unseen'], n_img)
        train_seen_dataset = mk('train_seen', all_unseen, n_img)
        val_dataset = mk('val', [], max(n_img // 4, 1))
    else:
        # real data under <data_dir> (reference layout, train.py:64-80); samples travel raw (uint8 image + label), the BGR /
        # mean transform and the target-embedding gather run on the GPU; images differ in size: batches > 1 are padded to the
        # largest image of the batch with ignored (-1) pixels (datasets.pad_collate; the reference trains at batch size 1)
        from .datasets import PascalContext, PascalVOC, pad_collate
        if args.batch_size > 1:
            collate = pad_collate
        cls = PascalVOC if cfg['dataset'] == 'pascal' else PascalContext
        mkr = lambda split: cls(split=split, embed_dim=cfg['embed_dim'], one_hot_embed=cfg['one_hot_embed'],
                                data_dir=args.data_dir, train_unseen=cfg['train_unseen'], val_unseen=cfg['val_unseen'],
                                native=True)
        train_dataset, train_seen_dataset, val_dataset = mkr('train'), mkr('train_seen'), mkr('val')
    kwargs = {'num_workers': args.workers, 'pin_memory': True}

    def loader(ds, bs, shuffle):
        kw = dict(kwargs, collate_fn=collate) if (collate is not None and bs > 1) else kwargs
        if world > 1 and shuffle:       # every rank holds the same dataset; the sampler deals disjoint shards per epoch
            sampler = torch.utils.data.distributed.DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=True, seed=1337)
            return torch.utils.data.DataLoader(ds, batch_size=bs, sampler=sampler, **kw)
        if world > 1:                   # validation: rank r evaluates images r, r + world, ... and loads ONLY those
            ds = torch.utils.data.Subset(ds, list(range(rank, len(ds), world)))
        return torch.utils.data.DataLoader(ds, batch_size=bs, shuffle=shuffle, **kw)
    train_loader = loader(train_dataset, args.batch_size, True)
    train_seen_loader = loader(train_seen_dataset, args.batch_size, True)
    val_loader = loader(val_dataset, 1, False)            # sharded by rank above; Trainer.validate all-reduces the sums
    val_loader.szn_sharded = world > 1                    # tells the trainers not to skip batches by index on top of it
    label_names = train_dataset.class_names
    if rank == 0 and not osp.exists(osp.join(log_dir, 'counts.csv')):
        with open(osp.join(log_dir, 'counts.csv'), 'w') as f:
            f.write('train_seen,train_unseen,val\n%d,%d,%d\n' % (len(train_seen_loader), len(train_loader) - len(train_seen_loader),
                                                                 len(val_dataset)))

    # 2. model
    model = {'fcn32s': models.FCN32s, 'fcn8s': models.FCN8s}[args.arch](n_class=cfg['embed_dim'] if cfg['embed_dim'] else 21)
    start_epoch, start_iteration, checkpoint = 0, 0, None
    if cfg['load_fcn_path']:
        checkpoint = torch.load(osp.join(args.data_dir, 'logs', cfg['load_fcn_path'], 'best'), map_location='cpu', weights_only=False)
        model.load_state_dict(checkpoint['model_state_dict'], strict=False)
        start_epoch, start_iteration = checkpoint['epoch'], checkpoint['iteration']
    else:
        try:
            model.copy_params_from_vgg16(models.VGG16(pretrained=True, data_dir=args.data_dir))
        except IOError as e:
            print("%s -> deterministic synthetic initialisation" % e)
            model.load_synthetic(1337)
    model = model.to(device)
    precision = {'bf16': torch.bfloat16, 'fp16': torch.float16, 'fp32': torch.float32}[args.precision]
    model.set_precision(precision)
    model._engine.dropout_seed = 1337 + 7919 * rank       # data-parallel ranks draw different Dropout2d masks

    # 3. fcn optimizer and trainer
    optim = make_fcn_optimizer(model, cfg)
    if cfg['load_fcn_path'] and checkpoint.get('optim_state_dict'):
        optim.load_state_dict(checkpoint['optim_state_dict'])
    fcn_trainer = trainer_fcn.Trainer(
        cuda=True, model=model, optimizer=optim, train_loader=train_seen_loader, val_loader=val_loader, log_dir=log_dir,
        dataset=cfg['dataset'], max_epoch=cfg['fcn_epochs'], pixel_embeddings=cfg['embed_dim'], loss_func=cfg['fcn_loss'],
        tb_writer=tb_writer, unseen=all_unseen, val_unseen=cfg['val_unseen'], label_names=label_names,
        forced_unseen=cfg['forced_unseen'], precision=precision, rank=rank)
    fcn_trainer.epoch, fcn_trainer.iteration = start_epoch, start_iteration

    if cfg['mode'] == 'train':
        if cfg['fcn_epochs'] > 0:
            fcn_trainer.train()
        # 4. seen-mask phase
        if cfg['seenmask_epochs'] > 0:
            freeze_for_seenmask(model)
            sm_optim = FusedAdam([{'params': list(get_parameters(model, seenmask=True))}], lr=cfg['seenmask_lr'])
            if world > 1:
                dist.barrier()               # rank 0 has finished writing <log_dir>/best before anyone reads it
            if not checkpoint:
                # the reference reloads <log_dir>/best (train.py:177-179); fall back to the last checkpoint when no
                # validation improved on best_mean_iu = 0 (tiny synthetic runs)
                for fname in ('best', 'checkpoint'):
                    if osp.exists(osp.join(log_dir, fname)):
                        checkpoint = torch.load(osp.join(log_dir, fname), map_location='cpu', weights_only=False)
                        break
                else:
                    checkpoint = {}
            seenmask_trainer = trainer_seenmask.Trainer(
                cuda=True, model=model, optimizer=sm_optim, train_loader=train_loader, val_loader=val_loader,
                log_dir=log_dir, dataset=cfg['dataset'], max_epoch=cfg['seenmask_epochs'], tb_writer=tb_writer,
                checkpoint=checkpoint, unseen=cfg['train_unseen'], rank=rank)
            seenmask_trainer.train()
    elif cfg['mode'] == 'test_fcn':
        fcn_trainer.validate(both_fcn_and_seenmask=False)
    elif cfg['mode'] == 'test_all':
        fcn_trainer.validate(both_fcn_and_seenmask=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
