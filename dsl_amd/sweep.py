"""Teacher / test-time sweep: network forward + on-GPU post-processing (dsl_fcos_detect).

Mirrors SingleStageDetector.simple_test (mmdet/models/detectors/single_stage.py:86-107) ->
FCOSHead.get_bboxes (dense_heads/fcos_head.py:340-548) -> multiclass_nms (core/post_processing/
bbox_nms.py:7-94) -> bbox2result (core/bbox/transforms.py:99-116) of the reference; the detections stay
on the GPU (`detect_device`) for the pseudo-label refresh, and `simple_test` converts them to the
reference's per-class numpy lists."""
import ctypes as C

import numpy as np
import torch

from . import _lib as L


class DetectPlan:
    def __init__(self, n, sizes, strides, device, num_classes=80, nms_pre=1000, max_per_img=100, score_thr=0.05,
                 iou_thr=0.5, ld_cls=80, ld_rc=8):
        d = L.DetDesc()
        d.nlvl, d.n = len(sizes), n
        d.h, d.w = L.seg5([s[0] for s in sizes]), L.seg5([s[1] for s in sizes])
        d.stride = L.seg5(strides)
        d.num_classes, d.nms_pre, d.max_per_img = num_classes, nms_pre, max_per_img
        d.score_thr, d.iou_thr = score_thr, iou_thr
        d.ld_cls, d.ld_rc = ld_cls, ld_rc
        self.dets = torch.zeros(n, max_per_img, 5, device=device)
        self.labels = torch.zeros(n, max_per_img, dtype=torch.int64, device=device)
        self.count = torch.zeros(n, dtype=torch.int32, device=device)
        self.img_shapes = torch.zeros(n, 2, device=device)
        self.scale_factors = torch.ones(n, 4, device=device)
        need = L.lib.dsl_detect_workspace_bytes(C.byref(d))
        self.ws = torch.empty(need, dtype=torch.uint8, device=device)
        d.dets, d.det_labels, d.det_count = L.ptr(self.dets), L.ptr(self.labels), L.ptr(self.count)
        d.img_shapes, d.scale_factors = L.ptr(self.img_shapes), L.ptr(self.scale_factors)
        d.workspace, d.workspace_bytes = L.ptr(self.ws), need
        self.desc, self.n = d, n
        self._keep = []

    def bind(self, cls_logits, regctr, scales):
        self.desc.cls_logits, self.desc.regctr, self.desc.scales = L.ptr(cls_logits), L.ptr(regctr), L.ptr(scales)
        self._keep = [cls_logits, regctr, scales]

    def set_meta(self, img_shapes, scale_factors, rescale):
        self.img_shapes.copy_(torch.tensor([[float(s[0]), float(s[1])] for s in img_shapes]), non_blocking=True)
        if rescale:
            sf = []
            for s in scale_factors:
                a = np.asarray(s, dtype=np.float32).reshape(-1)
                sf.append(np.repeat(a, 4) if a.size == 1 else a[:4])
            self.scale_factors.copy_(torch.from_numpy(np.stack(sf)), non_blocking=True)
            self.desc.scale_factors = L.ptr(self.scale_factors)
        else:
            self.desc.scale_factors = L.ptr(None)

    def run(self):
        L.check(L.lib.dsl_fcos_detect(C.byref(self.desc), L.stream_ptr()), 'dsl_fcos_detect')


def detect_device(det, img, img_metas, rescale=False, store=None, single_stream=False):
    """Forward + post-processing; returns (dets [N,100,5], labels [N,100], count [N]) on the GPU."""
    eng = det._get_engine()
    store = store or det.store
    N, _, H, W = img.shape
    plan = eng.plan(store, N, H, W, training=False, single_stream=single_stream)
    plan.bind_image(img)
    plan.fwd.run()
    dp = getattr(plan, 'detplan', None)
    if dp is None:
        cfg = det.test_cfg or {}
        nms = cfg.get('nms', {})
        dp = DetectPlan(N, plan.level_sizes, det.bbox_head.strides, store.device, nms_pre=cfg.get('nms_pre', 1000),
                        max_per_img=cfg.get('max_per_img', 100), score_thr=cfg.get('score_thr', 0.05),
                        iou_thr=nms.get('iou_threshold', nms.get('iou_thr', 0.5)))
        dp.bind(plan.bufs['cls_logits'], plan.bufs['regctr'], store.t32_ptr('head.scales'))
        plan.detplan = dp
    dp.set_meta([m['img_shape'] for m in img_metas], [m.get('scale_factor', 1.0) for m in img_metas], rescale)
    dp.run()
    return dp.dets, dp.labels, dp.count


def bbox2result(bboxes, labels, num_classes):
    """core/bbox/transforms.py:99-116."""
    if bboxes.shape[0] == 0:
        return [np.zeros((0, 5), dtype=np.float32) for _ in range(num_classes)]
    return [bboxes[labels == i, :] for i in range(num_classes)]


def simple_test(det, img, img_metas, rescale=False, store=None):
    dets, labels, count = detect_device(det, img, img_metas, rescale, store)
    dets, labels, count = dets.cpu().numpy(), labels.cpu().numpy(), count.cpu().numpy()
    return [bbox2result(dets[i, :count[i]], labels[i, :count[i]], det.bbox_head.num_classes) for i in range(len(count))]
