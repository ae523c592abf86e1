"""Host-side logic on CPU: registry / config surface, parameter store layout, runner helpers."""
import os

import numpy as np
import pytest
import torch

from util import fcos_model_cfg

REF_CFG = '/root/reference/configs/fcos_semi/r50_caffe_mslonger_tricks_0.Xdata.py'


def build():
    from dsl_amd import detectors  # noqa: F401
    from dsl_amd.registry import build_detector
    return build_detector(fcos_model_cfg())


def test_state_dict_matches_reference_layout():
    from oracle import fcos_oracle as O
    m = build()
    sd = O.synth_state_dict(0)
    m.load_state_dict(sd)
    out = m.state_dict()
    assert len(out) == 377 and set(out) == set(sd)
    for k, v in sd.items():
        assert tuple(out[k].shape) == tuple(v.shape) and out[k].dtype == v.dtype, k
        assert torch.equal(out[k], v), k
    train = [k for k, p in m.named_parameters() if p.requires_grad]
    assert sorted(train) == sorted(O.trainable_keys(sd))
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == 32021850
    # views alias the flat buffer: an in-place edit through state_dict is seen by the store
    out['bbox_head.conv_cls.bias'].fill_(-1.25)
    assert float(m.store.tview('head.cls_b')[:80].mean()) == -1.25
    lo_hi = m.store.grad_buckets()
    assert lo_hi[-1][0] == 0 and lo_hi[0][1] == m.store.n_train
    assert all(a[0] == b[1] for a, b in zip(lo_hi[:-1], lo_hi[1:]))


def test_hot_path_refuses_to_run_without_gpu():
    m = build()
    with pytest.raises(RuntimeError):
        m.forward_train(torch.zeros(1, 3, 64, 64), [dict()], [torch.zeros(0, 4)], [torch.zeros(0, dtype=torch.long)])


@pytest.mark.skipif(not os.path.exists(REF_CFG), reason='reference tree not present')
def test_reference_config_builds_unmodified():
    from dsl_amd import detectors, runner  # noqa: F401
    from dsl_amd.optim import build_optimizer
    from dsl_amd.registry import Config, build_detector
    cfg = Config.fromfile(REF_CFG)
    m = build_detector(cfg.model)
    opt = build_optimizer(m, cfg.optimizer, grad_clip=cfg.optimizer_config.get('grad_clip'))
    assert type(m).__name__ == 'FCOS' and opt.bias_lr_mult == 2.0 and opt.bias_decay_mult == 0.0
    cfg.merge_from_dict(Config.parse_cfg_options(['model.bbox_head.loss_weight=3.0', 'data.samples_per_gpu=4']))
    assert cfg.model.bbox_head.loss_weight == 3.0 and cfg.data.samples_per_gpu == 4
    # the DSL config: RLA_ResNet backbone, semi-supervised head, and every section train_detector reads
    dsl = Config.fromfile(os.path.join(os.path.dirname(REF_CFG), [f for f in os.listdir(os.path.dirname(REF_CFG)) if f.startswith('RLA')][0]))
    m2 = build_detector(dsl.model)
    assert type(m2.backbone).__name__ == 'RLA_ResNet' and m2.store.backbone == 'rla'
    assert m2.bbox_head.loss_weight == 3.0 and m2.bbox_head.soft_warm_up == 5000
    assert m2.backbone.pretrained_checkpoint.endswith('resnet50_rla_2283.pth.tar')
    from oracle import rla_oracle as RO
    sd, shapes = m2.state_dict(), RO.rla_param_shapes()
    assert set(sd) == set(shapes) and all(tuple(sd[k].shape) == tuple(v) for k, v in shapes.items())
    assert sorted(k for k, p in m2.named_parameters() if p.requires_grad) == sorted(RO.trainable_keys(RO.synth_state_dict(0)))
    opt2 = build_optimizer(m2, dsl.optimizer, grad_clip=dsl.optimizer_config.get('grad_clip'))
    assert opt2.max_norm == 35.0
    from dsl_amd.runner import SemiEpochBasedRunner, UnlabelPredHook
    from dsl_amd.apis import build_runner
    r = build_runner(dsl.runner, default_args=dict(model=m2, optimizer=opt2, work_dir=None, logger=None, meta=None, ema_model=m,
                                                   scale_invariant=dsl.get('scale_invariant', False)))
    assert isinstance(r, SemiEpochBasedRunner) and r.scale_invariant and r.max_epochs == 28
    r.register_training_hooks(dsl.lr_config, dsl.optimizer_config, dsl.ema_config, dsl.checkpoint_config, dsl.log_config)
    names = [type(h).__name__ for h in r._hooks]
    assert names == ['StepLrUpdaterHook', 'OptimizerHook', 'EMAOWNHook', 'CheckpointHook', 'TextLoggerHook'], names
    up = dict(dsl.data.unlabel_pred)
    up.pop('category_info_path')          # a file of the training machine
    hook = UnlabelPredHook(up, dsl, 'Det', interval_mode=up['eval_checkpoint_config']['mode'], interval=up['eval_checkpoint_config']['interval'])
    assert (hook.interval_mode, hook.interval, hook.start_point, hook.iou, hook.use_ema, hook.adathres_compute) == ('iteration', 1, 8, 0.6, True, True)


def test_pretrained_backbone_checkpoint_is_loaded_or_loudly_missing(tmp_path, monkeypatch):
    """init_cfg=dict(type='Pretrained', checkpoint=...) of the configs: the backbone-only checkpoint (un-prefixed keys) goes
    into the frozen / trainable buffers; a checkpoint that is not on the machine warns instead of silently training on
    random frozen features."""
    import warnings
    from dsl_amd import detectors
    from dsl_amd.registry import build_detector
    cfg = fcos_model_cfg()
    cfg['backbone']['init_cfg'] = dict(type='Pretrained', checkpoint='open-mmlab://detectron2/resnet50_caffe')
    m = build_detector(cfg)
    monkeypatch.setenv('DSL_PRETRAINED_DIR', str(tmp_path))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        m.init_weights()
    assert any('not found' in str(x.message) for x in w)
    ref = {k[len('backbone.'):]: torch.randn_like(v) for k, v in m.state_dict().items() if k.startswith('backbone.') and v.is_floating_point()}
    torch.save(dict(state_dict=ref), tmp_path / 'resnet50_msra-5891d200.pth')
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        m.init_weights()
    assert not any('not found' in str(x.message) for x in w)
    sd = m.state_dict()
    assert all(torch.equal(sd['backbone.' + k], v) for k, v in ref.items())
    assert detectors.resolve_checkpoint(str(tmp_path / 'resnet50_msra-5891d200.pth')) is not None


def test_scale_invariant_batch_matches_oracle():
    from dsl_amd.runner import append_half_scale
    from oracle import fcos_oracle as O
    g = torch.Generator().manual_seed(0)
    img = torch.randn(2, 3, 64, 96, generator=g)
    gb = [torch.rand(3, 4) * 50, torch.rand(2, 4) * 50]
    gl = [torch.tensor([1, 2, 3]), torch.tensor([4, 5])]
    ig = [torch.zeros(0, 4), torch.rand(1, 4) * 50]
    metas = [dict(img_shape=(64, 90, 3), pad_shape=(64, 96, 3), scale_factor=1.0)] * 2
    a = append_half_scale(img, gb, gl, ig, metas)
    b = O.append_half_scale(img, gb, gl, ig)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1][2], b[1][2]) and torch.equal(a[3][2], b[3][2])
    assert a[4][2]['img_shape'] == (32, 45, 3) and a[4][2]['pad_shape'] == (32, 48, 3)


def test_soft_warmup_counter_and_lr_schedule():
    from dsl_amd.runner import StepLrUpdaterHook
    m = build()
    h = m.bbox_head
    h.soft_weight, h.soft_warm_up = 1.0, 2
    assert [h.effective_soft_weight(3) for _ in range(5)] == [0.001, 0.001, 0.001, 1.0, 1.0]     # fcos_head.py:323-326
    assert h.effective_soft_weight(2) == 0.0

    class R:
        epoch, iter = 0, 0
        optimizer = type('O', (), {'param_groups': [dict(lr=0.01, initial_lr=0.01), dict(lr=0.02, initial_lr=0.02)]})()
    hook = StepLrUpdaterHook(step=[20, 26], warmup='linear', warmup_iters=500, warmup_ratio=1.0 / 3)
    r = R()
    hook.before_train_iter(r)
    assert r.optimizer.param_groups[0]['lr'] == pytest.approx(0.01 / 3)
    r.iter = 250
    hook.before_train_iter(r)
    assert r.optimizer.param_groups[0]['lr'] == pytest.approx(0.01 * (1 - 0.5 * (2 / 3)))
    r.iter, r.epoch = 10000, 21
    hook.before_train_iter(r)
    assert r.optimizer.param_groups[1]['lr'] == pytest.approx(0.002)


def test_adaptive_thresholds_and_split():
    from dsl_amd.runner import adaptive_thresholds, split_pseudo_labels
    by_c = {0: [0.9, 0.8, 0.35, 0.2], 1: [0.31], 2: [0.1]}
    thr, w = adaptive_thresholds(by_c)
    assert set(thr) == {0, 1} and all(0.3 <= v <= 0.35 for v in thr.values())
    avg = 4 / 2
    assert thr[0] == pytest.approx(max(min((2.05 / avg) ** 0.05 * 0.3, 0.35), 0.3))
    assert w[1] == pytest.approx((avg / 0.31) ** 0.6)
    gt, gl, ig = split_pseudo_labels(np.array([[0, 0, 10, 10], [5, 5, 30, 30], [1, 1, 1.5, 9]]), [0, 0, 1], [0.5, 0.2, 0.9],
                                     {0: 0.32})
    assert gt.shape == (1, 4) and gl.tolist() == [0] and ig.shape == (1, 4)


@pytest.mark.parametrize('backbone', ['resnet', 'rla'])
def test_weight_tiles_never_read_past_the_buffers(backbone):
    """The conv kernels fetch cout_pad rows of a weight; a 32-channel conv stores 32.  Every such tile must end inside the
    flat buffer it lives in (the frozen buffer of RLA_ResNet once ended 61 KB behind recurrent_convs.0: a GPU memory fault
    whenever the following page was unmapped)."""
    from dsl_amd.params import ParamStore
    st = ParamStore(80, 'cpu', backbone=backbone)
    for s in st.convs.values():
        regions, total = (st.train_regions, st.n_train) if s.trainable else (st.frozen_regions, st.n_frozen)
        off = regions[s.name + '.weight'][0]
        assert off + s.cout_pad * s.k * s.k * s.cin_store <= total, s.name
    for name, rows in (('head.cls_w', 128), ('head.regctr_w', 64)):
        assert st.train_regions[name][0] + rows * 9 * 256 <= st.n_train


def test_optimizer_state_is_remapped_by_region_not_by_prefix():
    """Round-3 advisor: a restored momentum tensor whose flat layout differs from this build's (other tile padding between regions)
    used to be prefix-copied - silently attached to the wrong parameters.  The state now carries the region table and is remapped
    region by region; a state without a table whose size is off by more than trailing padding is refused."""
    import pytest
    import torch
    from dsl_amd import detectors  # noqa: F401
    from dsl_amd.optim import FlatSGD
    from dsl_amd.registry import build_detector
    from util import fcos_model_cfg
    m = build_detector(fcos_model_cfg())
    opt = FlatSGD(m, lr=0.01, momentum=0.9)
    st = m.store
    mom = torch.arange(st.train.numel(), dtype=torch.float32) % 1013
    # what another build could have saved: every region shifted by a growing amount of padding
    regs, chunks, off = {}, [], 0
    for i, (k, (o, n, _)) in enumerate(sorted(st.train_regions.items(), key=lambda kv: kv[1][0])):
        pad = 8 * (i % 3)
        chunks += [mom[o:o + n], torch.full((pad,), -7.0)]
        regs[k] = (off, n)
        off += n + pad
    other = torch.cat(chunks)
    assert other.numel() != mom.numel()
    opt.load_state_dict(dict(momentum=other, steps=5, param_groups=opt.param_groups, regions=regs))
    opt._adopt_loaded_state()
    covered = torch.zeros(st.train.numel(), dtype=torch.bool)
    for k, (o, n, _) in st.train_regions.items():
        assert torch.equal(opt.momentum_buf[o:o + n], mom[o:o + n]), k
        covered[o:o + n] = True
    assert float(opt.momentum_buf[~covered].abs().sum()) == 0 and opt.steps == 5 and -7.0 not in opt.momentum_buf
    # round trip of this build's own state: identical
    sd = opt.state_dict()
    assert set(sd) == {'momentum', 'steps', 'param_groups', 'regions'}
    opt2 = FlatSGD(m, lr=0.01, momentum=0.9)
    opt2.load_state_dict(sd)
    opt2._adopt_loaded_state()
    assert torch.equal(opt2.momentum_buf, opt.momentum_buf)
    # no table and a size that is not "this layout + trailing padding": refused
    opt3 = FlatSGD(m, lr=0.01, momentum=0.9)
    opt3.load_state_dict(dict(momentum=other[:-100000], steps=5, param_groups=opt.param_groups))
    with pytest.raises(RuntimeError, match='region table'):
        opt3._adopt_loaded_state()
    # no table, trailing padding only: accepted (checkpoints of the previous rounds)
    opt4 = FlatSGD(m, lr=0.01, momentum=0.9)
    opt4.load_state_dict(dict(momentum=mom[:-8], steps=2, param_groups=opt.param_groups))
    opt4._adopt_loaded_state()
    assert torch.equal(opt4.momentum_buf[:-8], mom[:-8])
