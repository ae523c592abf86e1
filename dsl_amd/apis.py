"""Config-driven assembly of the training loop: `train_detector` with the reference's signature
(mmdet/apis/train.py:41-212) - optimizer, (distributed) model wrappers, runner, the training hooks named by the
config (`lr_config`, `optimizer_config`, `ema_config`, `checkpoint_config`, `log_config`, `custom_hooks`,
`data.unlabel_pred`), resume / load - so that configs/fcos_semi/*.py drive the HIP step unmodified.

What differs: `dataset` is a list of data LOADERS (objects with __iter__/__len__ yielding the batch dicts of
dsl_amd/data.py; the reference builds its CPU DataLoaders from dataset objects at :45-81 - the decode/augment
pipeline is SURVEY.md §8 f3, out of the hot path), and the wrappers are HipDistributedDataParallel / none instead of
MMDistributedDataParallel / MMDataParallel (:84-104).
"""
import logging
import random
import warnings

import numpy as np
import torch

from .optim import build_optimizer
from .registry import HOOKS, RUNNERS, build_from_cfg
from . import runner as _runner_mod  # noqa: F401  (registers the runner and the hooks)


def set_random_seed(seed, deterministic=False):
    """apis/train.py:23-39."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def get_root_logger(log_level='INFO'):
    logger = logging.getLogger('dsl_amd')
    if not logger.handlers:
        h = logging.StreamHandler()
        h.setFormatter(logging.Formatter('%(asctime)s - %(name)s - %(levelname)s - %(message)s'))
        logger.addHandler(h)
    logger.setLevel(getattr(logging, log_level) if isinstance(log_level, str) else log_level)
    return logger


def build_runner(cfg, default_args=None):
    return build_from_cfg(dict(cfg), RUNNERS, default_args)


def train_detector(model, dataset, cfg, distributed=False, validate=False, timestamp=None, meta=None, ema_model=None):
    logger = get_root_logger(cfg.get('log_level', 'INFO'))
    data_loaders = list(dataset) if isinstance(dataset, (list, tuple)) else [dataset]
    for dl in data_loaders:
        if not (hasattr(dl, '__iter__') and hasattr(dl, '__len__')):
            raise TypeError('train_detector: pass data loaders (see dsl_amd/data.py); dataset pipelines are not built here')

    # put model on gpus (:84-104)
    if distributed:
        from .parallel import HipDistributedDataParallel
        model = HipDistributedDataParallel(model.cuda())
        if ema_model is not None:
            ema_model = ema_model.cuda()          # the teacher is never communicated: every rank's EMA is identical
    else:
        model = model.cuda()
        if ema_model is not None:
            ema_model = ema_model.cuda()

    # build runner (:111-153)
    optimizer_config = cfg.get('optimizer_config') or {}
    optimizer = build_optimizer(model, cfg.optimizer, grad_clip=optimizer_config.get('grad_clip'))
    if 'runner' not in cfg:
        runner_cfg = dict(type='EpochBasedRunner', max_epochs=cfg.total_epochs)
        warnings.warn('config is now expected to have a `runner` section, please set `runner` in your config.', UserWarning)
    else:
        runner_cfg = cfg.runner
        if 'total_epochs' in cfg:
            assert cfg.total_epochs == cfg.runner.max_epochs
    args = dict(model=model, optimizer=optimizer, work_dir=cfg.get('work_dir'), logger=logger, meta=meta)
    if ema_model is not None:
        args.update(ema_model=ema_model, scale_invariant=cfg.get('scale_invariant', False))
    runner = build_runner(runner_cfg, default_args=args)
    runner.ema_flag = False
    runner.timestamp = timestamp
    ema_config = cfg.get('ema_config', None)
    if cfg.get('fp16', None) is not None:
        raise NotImplementedError('fp16 loss scaling (Fp16OptimizerHook) is not part of the HIP path: it computes in bf16')

    # register hooks (:155-176)
    runner.register_training_hooks(cfg.lr_config, optimizer_config, ema_config if ema_model is not None else None,
                                   cfg.get('checkpoint_config'), cfg.get('log_config'), cfg.get('momentum_config', None))
    if distributed:
        runner.register_hook(HOOKS.get('DistSamplerSeedHook_semi')())
    if validate:
        from .evaluation import EvalHook
        val_loader = cfg.get('val_dataloader')           # a loader object the caller put into the config
        if val_loader is None:
            raise ValueError('validate=True needs cfg.val_dataloader (a loader of test batches): datasets are not built here')
        eval_cfg = dict(cfg.get('evaluation', {}))
        runner.register_hook(EvalHook(val_loader, **eval_cfg))
    # for unlabel pred (:191-197)
    unlabel_pred_cfg = (cfg.get('data') or {}).get('unlabel_pred', None)
    if unlabel_pred_cfg is not None:
        ec = unlabel_pred_cfg.get('eval_checkpoint_config') or {}
        hook = HOOKS.get('UnlabelPredHook')(unlabel_pred_cfg, cfg, 'Det', interval_mode=ec.get('mode', 'epoch'),
                                            interval=ec.get('interval', 1),
                                            bank=getattr(data_loaders[0], 'bank', None))
        runner.register_hook(hook)
        for dl in data_loaders:                # loaders read the unlabeled samples' annotations from the hook's bank
            if getattr(dl, 'bank', None) is None and hasattr(dl, 'bank'):
                dl.bank = hook.bank
    # user-defined hooks (:199-211)
    if cfg.get('custom_hooks', None):
        assert isinstance(cfg.custom_hooks, list), f'custom_hooks expect list type, but got {type(cfg.custom_hooks)}'
        for hook_cfg in cfg.custom_hooks:
            assert isinstance(hook_cfg, dict), f'Each item in custom_hooks expects dict type, but got {type(hook_cfg)}'
            hook_cfg = dict(hook_cfg)
            priority = hook_cfg.pop('priority', 'NORMAL')
            runner.register_hook(build_from_cfg(hook_cfg, HOOKS), priority=priority)

    if cfg.get('resume_from'):
        runner.resume(cfg.resume_from)
    elif cfg.get('load_from'):
        runner.load_checkpoint(cfg.load_from)
    runner.run(data_loaders, cfg.get('workflow', [('train', 1)]))
    return runner
