"""Shared helpers for the tests (no reference files are read here)."""
import numpy as np
import torch


def fcos_model_cfg(**head):
    """The `model` dict of the supervised FCOS R50-caffe config (same keys/values the reference's
    configs/fcos_semi/r50_caffe_mslonger_tricks_0.Xdata.py:2-62 uses)."""
    bbox_head = dict(type='FCOSHead', num_classes=80, in_channels=256, stacked_convs=4, feat_channels=256,
                     strides=[8, 16, 32, 64, 128], norm_on_bbox=True, centerness_on_reg=True, dcn_on_last_conv=False,
                     center_sampling=True, conv_bias=True,
                     loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
                     loss_bbox=dict(type='GIoULoss', loss_weight=1.0),
                     loss_centerness=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0))
    bbox_head.update(head)
    return dict(
        type='FCOS',
        backbone=dict(type='ResNet', depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                      norm_cfg=dict(type='BN', requires_grad=False), norm_eval=True, style='caffe'),
        neck=dict(type='FPN', in_channels=[256, 512, 1024, 2048], out_channels=256, start_level=1,
                  add_extra_convs='on_output', num_outs=5, relu_before_extra_convs=True),
        bbox_head=bbox_head,
        train_cfg=dict(assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0,
                                     ignore_iof_thr=-1), allowed_border=-1, pos_weight=-1, debug=False),
        test_cfg=dict(nms_pre=1000, min_bbox_size=0, score_thr=0.05, nms=dict(type='nms', iou_threshold=0.5),
                      max_per_img=100))


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def levels_to_flat(ts):
    return torch.cat([t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]) for t in ts])
