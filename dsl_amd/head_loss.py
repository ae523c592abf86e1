"""Host side of the fused FCOS target assignment + loss (dsl_fcos_assign / dsl_fcos_loss).

Mirrors `FCOSHead.loss` of the reference (mmdet/models/dense_heads/fcos_head.py:170-338): same
inputs (per-image gt boxes / labels / ignore boxes), same outputs (loss_cls, loss_bbox,
loss_centerness[, loss_sisoft]); the gradients w.r.t. the head outputs are produced in the same pass.
All buffers are level-major [level][image][y][x] = the reference's flattened order."""
import ctypes as C

import torch

from . import _lib as L
from . import ops

STRIDES = (8, 16, 32, 64, 128)
REGRESS_RANGES = ((-1, 64), (64, 128), (128, 256), (256, 512), (512, 1e8))   # fcos_head.py:61-62


class FcosLossPlan:
    LD_CLS, LD_RC, LD_GCLS, LD_GRC = 80, 8, 128, 64

    def __init__(self, n, sizes, device, strides=STRIDES, ranges=REGRESS_RANGES, num_classes=80,
                 radius=1.5, max_gt=1024):
        assert num_classes == 80, 'kernel strides are laid out for 80 classes'
        self.n, self.sizes, self.strides, self.device = n, [tuple(s) for s in sizes], strides, device
        self.M = n * sum(h * w for h, w in self.sizes)
        M, dev = self.M, device
        self.labels = torch.empty(M, dtype=torch.int64, device=dev)
        self.bbox_targets = torch.empty(M, 4, dtype=torch.float32, device=dev)
        self.assign_idx = torch.empty(M, dtype=torch.int32, device=dev)
        self.cls_weight = torch.empty(M, dtype=torch.float32, device=dev)
        self.pos_weight = torch.empty(M, dtype=torch.float32, device=dev)
        self.stats = torch.zeros(8, dtype=torch.float32, device=dev)
        self.losses = torch.zeros(4, dtype=torch.float32, device=dev)
        self.logvec = torch.zeros(5, dtype=torch.float32, device=dev)      # the loss terms in log order + their sum (dsl_fcos_desc.logvec)
        self.g_scales = torch.zeros(L.MAX_SEG, dtype=torch.float32, device=dev)
        # gradient buffers: padding columns stay zero forever (the kernels only write real columns)
        self.g_cls = torch.zeros(M, self.LD_GCLS, dtype=torch.bfloat16, device=dev)
        self.g_rc = torch.zeros(M, self.LD_GRC, dtype=torch.bfloat16, device=dev)
        self.max_gt = max_gt
        self.gt_boxes = torch.zeros(max_gt, 4, dtype=torch.float32, device=dev)
        self.gt_labels = torch.zeros(max_gt, dtype=torch.int64, device=dev)
        self.gt_off = torch.zeros(n + 1, dtype=torch.int32, device=dev)
        self.ig_boxes = torch.zeros(max_gt, 4, dtype=torch.float32, device=dev)
        self.ig_off = torch.zeros(n + 1, dtype=torch.int32, device=dev)
        # pinned staging: pageable H2D copies would block the host until the stream drains (one stall per step).
        # A ring of slots, each guarded by an event recorded behind its async copies: the host may run several steps
        # ahead of the GPU (lazy_log, no per-step sync) and must not rewrite a slot whose copy has not executed yet.
        pin = torch.cuda.is_available()
        self._slots = [dict(boxes=torch.zeros(2, max_gt, 4, dtype=torch.float32, pin_memory=pin),
                            labels=torch.zeros(max_gt, dtype=torch.int64, pin_memory=pin),
                            off=torch.zeros(2, n + 1, dtype=torch.int32, pin_memory=pin), ev=None) for _ in range(4)]
        self._slot_i = 0
        self.desc = ops.fcos_desc(n=n, sizes=self.sizes, strides=strides, ranges=ranges, radius=radius,
                                  num_classes=num_classes)
        ops.set_ptrs(self.desc, gt_boxes=self.gt_boxes, gt_labels=self.gt_labels, gt_off=self.gt_off,
                     labels=self.labels, bbox_targets=self.bbox_targets, assign_idx=self.assign_idx,
                     cls_weight=self.cls_weight, pos_weight=self.pos_weight, stats=self.stats,
                     norm=self.stats, g_cls=self.g_cls, g_rc=self.g_rc, g_scales=self.g_scales,
                     losses=self.losses, logvec=self.logvec)
        self.desc.ld_cls, self.desc.ld_rc = self.LD_CLS, self.LD_RC
        self.desc.ld_gcls, self.desc.ld_grc = self.LD_GCLS, self.LD_GRC
        need = L.lib.dsl_fcos_workspace_bytes(C.byref(self.desc))
        self.ws = torch.empty(need, dtype=torch.uint8, device=dev)
        self.desc.workspace, self.desc.workspace_bytes = L.ptr(self.ws), need

    # -- ground truth upload (host lists -> one pinned staging copy) --------------------------------
    def set_targets(self, gt_bboxes, gt_labels, gt_bboxes_ignore=None):
        assert len(gt_bboxes) == self.n == len(gt_labels)
        slot = self._slots[self._slot_i]
        self._slot_i = (self._slot_i + 1) % len(self._slots)
        if slot['ev'] is not None:
            slot['ev'].synchronize()          # the copies issued from this slot len(_slots) steps ago have executed

        def stage(boxes, which, dev_boxes, labels=None):
            """Offsets go through the pinned slot (shapes are host data); box/label payloads that already live on the
            GPU are concatenated there (no D2H round trip), host payloads go through the slot."""
            off, counts = slot['off'], [int(b.shape[0]) for b in boxes]
            tot = sum(counts)
            assert tot <= self.max_gt, f'more than {self.max_gt} boxes in one batch'
            off[which, 0] = 0
            for i, k in enumerate(counts):
                off[which, i + 1] = off[which, i] + k
            on_dev = tot > 0 and all(b.is_cuda for b, k in zip(boxes, counts) if k)
            if tot and on_dev:
                dev_boxes[:tot].copy_(torch.cat([b.reshape(-1, 4) for b, k in zip(boxes, counts) if k]).float())
                if labels is not None:
                    self.gt_labels[:tot].copy_(torch.cat([l.reshape(-1) for l, k in zip(labels, counts) if k]).long())
            elif tot:
                pos = 0
                for i, (b, k) in enumerate(zip(boxes, counts)):
                    if k:
                        slot['boxes'][which, pos:pos + k].copy_(b.reshape(-1, 4))
                        if labels is not None:
                            slot['labels'][pos:pos + k].copy_(labels[i].reshape(-1))
                    pos += k
                dev_boxes[:tot].copy_(slot['boxes'][which, :tot], non_blocking=True)
                if labels is not None:
                    self.gt_labels[:tot].copy_(slot['labels'][:tot], non_blocking=True)
            return tot

        stage(gt_bboxes, 0, self.gt_boxes, gt_labels)
        self.gt_off.copy_(slot['off'][0], non_blocking=True)
        if gt_bboxes_ignore is not None:
            assert len(gt_bboxes_ignore) == self.n
            stage(gt_bboxes_ignore, 1, self.ig_boxes)
            self.ig_off.copy_(slot['off'][1], non_blocking=True)
            ops.set_ptrs(self.desc, ig_boxes=self.ig_boxes, ig_off=self.ig_off)
        else:
            ops.set_ptrs(self.desc, ig_boxes=None, ig_off=None)
        if self.gt_off.is_cuda:
            slot['ev'] = torch.cuda.Event()
            slot['ev'].record()

    def configure(self, loss_weight=1.0, soft_weight=0.0, grad_scale=1.0, inv_world=1.0):
        d = self.desc
        d.loss_weight, d.soft_weight, d.grad_scale, d.inv_world = loss_weight, soft_weight, grad_scale, inv_world

    def bind_outputs(self, cls_logits, regctr, scales):
        ops.set_ptrs(self.desc, cls_logits=cls_logits, regctr=regctr, scales=scales)

    def assign(self):
        L.check(L.lib.dsl_fcos_assign(C.byref(self.desc), L.stream_ptr()), 'dsl_fcos_assign')

    def loss(self):
        L.check(L.lib.dsl_fcos_loss(C.byref(self.desc), L.stream_ptr()), 'dsl_fcos_loss')
