"""Generate tests/golden/*.npz by running the REFERENCE's own python files on CPU.

Container-only (needs /root/reference).  Run:  python tests/golden/make_golden.py
The fixtures are data (inputs + the reference's outputs); tests read only the .npz files.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import _ref_import as R  # noqa: E402
from oracle import fcos_oracle as O  # noqa: E402

SUP_CFG = '/root/reference/configs/fcos_semi/r50_caffe_mslonger_tricks_0.Xdata.py'
torch.set_num_threads(8)


def npify(d):
    out = {}
    for k, v in d.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    return out


def save(name, **d):
    np.savez_compressed(os.path.join(HERE, name), **npify(d))
    print('wrote', name, {k: getattr(v, 'shape', None) for k, v in d.items()})


def gts_for(rng, H, W, n, lo=8.0, hi=None):
    b = O.synth_boxes(rng, n, H=H, W=W, lo=lo, hi=hi or max(H, W))
    return torch.from_numpy(b), torch.from_numpy(rng.randint(0, 80, len(b)).astype('int64'))


def head_of(model):
    return model.bbox_head


# ------------------------------------------------------------------------------------------------
def gen_assign(model):
    head = head_of(model)
    # (1) crafted small case: 5 levels of a 128x160 canvas; ties in area, point exactly on a box
    # edge / centre-box edge, nested boxes, degenerate box, box outside all regress ranges
    sizes = [(16, 20), (8, 10), (4, 5), (2, 3), (1, 2)]
    pts = head.get_points(sizes, torch.float32, 'cpu')
    gtb = [torch.tensor([[0., 0., 64., 64.], [32., 32., 96., 96.], [0., 0., 64., 64.],   # tie 0/2
                         [4., 4., 20., 20.], [4., 4., 36., 36.], [60., 20., 60., 90.],
                         [10., 12., 150., 120.], [100., 4., 108., 12.]]),
           torch.zeros(0, 4),
           torch.tensor([[12., 12., 28., 28.], [12., 12., 28., 28.]])]
    gtl = [torch.tensor([1, 2, 3, 4, 5, 6, 7, 8]), torch.zeros(0, dtype=torch.long),
           torch.tensor([9, 10])]
    labels, tg = head.get_targets(pts, gtb, gtl)
    save('assign_small.npz', sizes=np.array(sizes), n_img=3,
         gt0=gtb[0], gt1=gtb[1], gt2=gtb[2], gl0=gtl[0], gl1=gtl[1], gl2=gtl[2],
         labels=torch.cat(labels).to(torch.int16), bbox_targets=torch.cat(tg),
         points=torch.cat(pts))
    # (2) full size canvas 800x1344, G up to 40
    sizes = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
    pts = head.get_points(sizes, torch.float32, 'cpu')
    rng = np.random.RandomState(2024)
    gtb, gtl = [], []
    for n in (7, 40):
        b, l = gts_for(rng, 800, 1333, n, lo=16.0, hi=600.0)
        gtb.append(b)
        gtl.append(l)
    labels, tg = head.get_targets(pts, gtb, gtl)
    save('assign_full.npz', sizes=np.array(sizes), n_img=2, gt0=gtb[0], gt1=gtb[1], gl0=gtl[0],
         gl1=gtl[1], labels=torch.cat(labels).to(torch.int16), bbox_targets=torch.cat(tg))


def rand_head_outputs(g, B, sizes, cls_bias=-2.0):
    cls = [(torch.randn(B, 80, h, w, generator=g) * 1.5 + cls_bias).requires_grad_() for h, w in sizes]
    reg = [(torch.rand(B, 4, h, w, generator=g) * 6.0 * (torch.rand(B, 4, h, w, generator=g) > 0.1)
            ).requires_grad_() for h, w in sizes]
    ctr = [torch.randn(B, 1, h, w, generator=g).requires_grad_() for h, w in sizes]
    return cls, reg, ctr


def gen_loss(model):
    head = head_of(model)
    sizes = [(16, 24), (8, 12), (4, 6), (2, 3), (1, 2)]       # 128 x 192 canvas
    rng = np.random.RandomState(7)
    for name, B, lw, sw, warm, with_ig in (('loss_sup', 2, 1.0, 0.0, 0, False),
                                           ('loss_sup_ig', 2, 1.0, 0.0, 0, True),
                                           ('loss_dsl', 3, 3.0, 1.0, 0, True),
                                           ('loss_dsl_warm', 3, 3.0, 1.0, 5, True),
                                           ('loss_dsl_even', 2, 3.0, 1.0, 0, True),
                                           ('loss_nopos', 2, 1.0, 0.0, 0, False)):
        g = torch.Generator().manual_seed(hash(name) & 0xFFFF if False else sum(map(ord, name)))
        cls, reg, ctr = rand_head_outputs(g, B, sizes)
        gtb, gtl, igb = [], [], []
        for i in range(B):
            n = 0 if name == 'loss_nopos' else int(rng.randint(1, 6))
            b, l = gts_for(rng, 128, 192, n, lo=8.0, hi=160.0)
            gtb.append(b)
            gtl.append(l)
            ib, _ = gts_for(rng, 128, 192, int(rng.randint(0, 4)), lo=8.0, hi=100.0)
            igb.append(ib)
        if name.startswith('loss_dsl') and B == 3:
            gtb[2], gtl[2], igb[2] = gtb[1] / 2, gtl[1], igb[1] / 2
        head.loss_weight, head.soft_weight, head.soft_warm_up, head.cur_iter = lw, sw, warm, 0
        metas = [dict(img_shape=(128, 192, 3))] * B
        losses = head.loss(cls, reg, ctr, gtb, gtl, metas, gt_bboxes_ignore=igb if with_ig else None)
        total = sum(v for v in losses.values())
        total.backward()
        d = dict(sizes=np.array(sizes), B=B, loss_weight=lw, soft_weight=sw, soft_warm_up=warm,
                 with_ig=int(with_ig))
        for i in range(B):
            d[f'gt{i}'], d[f'gl{i}'], d[f'ig{i}'] = gtb[i], gtl[i], igb[i]
        for i in range(5):
            d[f'cls{i}'], d[f'reg{i}'], d[f'ctr{i}'] = cls[i], reg[i], ctr[i]
            d[f'gcls{i}'], d[f'greg{i}'], d[f'gctr{i}'] = cls[i].grad, reg[i].grad, ctr[i].grad
        for k, v in losses.items():
            d[k] = np.float64(float(v))
        save(name + '.npz', **d)
    head.loss_weight, head.soft_weight, head.soft_warm_up, head.cur_iter = 1.0, 0.0, 0, 0


def gen_net(model):
    """Whole detector, tiny canvas, synthetic key-addressed weights (oracle.synth_state_dict)."""
    sd = O.synth_state_dict(0)
    missing = model.load_state_dict(sd, strict=True)
    print('load_state_dict', missing)
    model.train()
    tk = O.trainable_keys(sd)
    ref_tk = [k for k, p in model.named_parameters() if p.requires_grad]
    assert sorted(tk) == sorted(ref_tk), (set(tk) ^ set(ref_tk))
    rng = np.random.RandomState(11)
    for name, B, H, W, dsl in (('net_tiny', 2, 64, 96, False), ('net_small_dsl', 2, 96, 128, True)):
        g = torch.Generator().manual_seed(5 + B + H)
        img = torch.randn(B, 3, H, W, generator=g) * 40.0
        gtb, gtl, igb = [], [], []
        for i in range(B):
            b, l = gts_for(rng, H, W, int(rng.randint(1, 4)), lo=8.0, hi=80.0)
            gtb.append(b)
            gtl.append(l)
            ib, _ = gts_for(rng, H, W, int(rng.randint(1, 3)), lo=8.0, hi=60.0)
            igb.append(ib)
        head = head_of(model)
        if dsl:
            head.loss_weight, head.soft_weight, head.soft_warm_up, head.cur_iter = 3.0, 1.0, 0, 1
            img, gtb, gtl, igb = O.append_half_scale(img, gtb, gtl, igb)
        model.zero_grad()
        metas = [dict(img_shape=(H, W, 3), pad_shape=(H, W, 3), scale_factor=1.0)] * len(gtb)
        feats = model.extract_feat(img)
        outs = model.bbox_head(feats)
        losses = model.bbox_head.loss(*outs, gtb, gtl, metas, gt_bboxes_ignore=igb if dsl else None)
        sum(losses.values()).backward()
        d = dict(img=img, B=len(gtb), dsl=int(dsl))
        for i in range(len(gtb)):
            d[f'gt{i}'], d[f'gl{i}'], d[f'ig{i}'] = gtb[i], gtl[i], igb[i]
        for i in range(5):
            d[f'feat{i}'] = feats[i]
            d[f'cls{i}'], d[f'reg{i}'], d[f'ctr{i}'] = outs[0][i], outs[1][i], outs[2][i]
        for k, v in losses.items():
            d[k] = np.float64(float(v))
        named = dict(model.named_parameters())
        d['grad_keys'] = np.array(tk)
        d['grad_norms'] = np.array([float(named[k].grad.norm()) for k in tk], dtype=np.float64)
        for k in ('bbox_head.conv_reg.weight', 'bbox_head.conv_cls.bias', 'bbox_head.scales.0.scale',
                  'bbox_head.scales.3.scale', 'bbox_head.cls_convs.0.gn.weight',
                  'bbox_head.reg_convs.3.gn.bias', 'neck.lateral_convs.2.conv.bias',
                  'backbone.layer2.0.conv1.weight'):
            d['grad/' + k] = named[k].grad
        save(name + '.npz', **d)
        head.loss_weight, head.soft_weight, head.soft_warm_up, head.cur_iter = 1.0, 0.0, 0, 0
    # teacher sweep on the tiny net (eval mode: bbox_pred * stride); raise the focal prior so that
    # some scores pass score_thr (recorded in the fixture as cls_bias)
    sd2 = dict(sd)
    sd2['bbox_head.conv_cls.bias'] = torch.full((80,), -1.5)
    model.load_state_dict(sd2, strict=True)
    model.eval()
    g = torch.Generator().manual_seed(99)
    img = torch.randn(2, 3, 96, 128, generator=g) * 40.0
    sf = np.array([1.25, 1.25, 1.25, 1.25], dtype=np.float32)
    metas = [dict(img_shape=(90, 120, 3), pad_shape=(96, 128, 3), scale_factor=sf)] * 2
    with torch.no_grad():
        feats = model.extract_feat(img)
        outs = model.bbox_head(feats)
        dets = model.bbox_head.get_bboxes(*outs, metas, rescale=True)
    d = dict(img=img, scale_factor=sf, img_shape=np.array([90, 120, 3]), cls_bias=-1.5)
    for i in range(5):
        d[f'cls{i}'], d[f'reg{i}'], d[f'ctr{i}'] = outs[0][i], outs[1][i], outs[2][i]
    for i, (b, l) in enumerate(dets):
        d[f'det{i}'], d[f'lab{i}'] = b, l
        print('dets', i, b.shape)
    save('sweep_tiny.npz', **d)


def gen_bboxes(model):
    """get_bboxes on synthetic head outputs with enough confident scores to exercise top-k + NMS."""
    head = head_of(model)
    model.eval()
    sizes = [(40, 56), (20, 28), (10, 14), (5, 7), (3, 4)]     # 320 x 448 canvas, P3 has 2240 > 1000
    g = torch.Generator().manual_seed(3)
    B = 2
    cls = [torch.randn(B, 80, h, w, generator=g) * 2.0 - 3.0 for h, w in sizes]
    reg = [torch.rand(B, 4, h, w, generator=g) * 8.0 * s for (h, w), s in zip(sizes, O.STRIDES)]
    ctr = [torch.randn(B, 1, h, w, generator=g) for h, w in sizes]
    sf = np.array([0.5, 0.5, 0.5, 0.5], dtype=np.float32)
    metas = [dict(img_shape=(310, 440, 3), scale_factor=sf)] * B
    dets = head.get_bboxes(cls, reg, ctr, metas, rescale=True)
    d = dict(sizes=np.array(sizes), scale_factor=sf, img_shape=np.array([310, 440, 3]))
    for i in range(5):
        d[f'cls{i}'], d[f'reg{i}'], d[f'ctr{i}'] = cls[i], reg[i], ctr[i]
    for i, (b, l) in enumerate(dets):
        d[f'det{i}'], d[f'lab{i}'] = b, l
        print('dets', i, b.shape)
    save('bboxes_synth.npz', **d)


def gen_adathres():
    """Adaptive per-class thresholds / class weights of the pseudo-label refresh (runner/hooks/unlabel_pred_hook.py
    `adathres`, :295-367).  The hook module itself needs pycocotools / mmcv.parallel / the dataset pipelines at import;
    the function is plain Python over per-image JSON files, so it is run here from its own source text (read from the
    reference tree at generation time - nothing of it is stored) on synthetic pseudo-label files, twice: without and
    with the history file of the first call.  The fixture holds the inputs and the two outputs."""
    import ast
    import json
    import tempfile
    path = os.path.join(R.REF, 'mmdet/runner/hooks/unlabel_pred_hook.py')
    src = open(path).read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == 'adathres'][0]
    ns = dict(os=os, json=json)
    exec(compile(ast.Module(body=[fn], type_ignores=[]), path, 'exec'), ns)
    adathres = ns['adathres']
    rng = np.random.RandomState(11)
    names = [f'cls{i}' for i in range(12)]                 # category names <-> ids 1..12 (COCO ids are 1-based)
    cat2id = {n: i + 1 for i, n in enumerate(names)}
    id2cat = {str(i + 1): n for i, n in enumerate(names)}
    tmp = tempfile.mkdtemp()
    rounds = []
    files = [f'unl/img{i}.jpg' for i in range(40)]
    for rnd in range(2):
        per_img = []
        for f in files:
            k = int(rng.poisson(3))
            tags = [names[j] for j in rng.randint(0, 10 if rnd == 0 else 12, k)]     # round 2 meets two unseen classes
            scores = [round(float(v), 6) for v in rng.uniform(0.1, 0.99, k)]
            with open(os.path.join(tmp, os.path.basename(f) + '.json'), 'w') as fh:
                json.dump(dict(imageName=f, targetNum=k, rects=[[0, 0, 5, 5]] * k, tags=tags, masks=[[]] * k, scores=scores), fh)
            per_img.append((tags, scores))
        hist = os.path.join(tmp, 'thres.json')
        adathres(0, True, hist, id2cat, cat2id, files, tmp, {})
        out = json.load(open(hist))
        rounds.append(dict(per_img=per_img, thres=out['thres'], weights=out['cat'], id_weights=out['id']))
        print('adathres round', rnd, {k: round(v, 4) for k, v in list(out['thres'].items())[:4]})
    with open(os.path.join(HERE, 'adathres.json'), 'w') as fh:
        json.dump(dict(names=names, rounds=rounds), fh)
    print('wrote adathres.json')


def gen_pseudo_split():
    """The (gt, ignore) split of stored pseudo labels: SemiCOCODataset._parse_ann_info (datasets/semicoco.py:184-291),
    run from its own source text on synthetic per-image JSON files with (a) no threshold file yet (default band
    [0.1, 0.4)), (b) a per-class threshold file that misses some classes."""
    import ast
    import json
    import tempfile
    import types
    path = os.path.join(R.REF, 'mmdet/datasets/semicoco.py')
    tree = ast.parse(open(path).read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == 'SemiCOCODataset'][0]
    fn = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == '_parse_ann_info'][0]
    ns = dict(os=os, json=json, np=np)
    exec(compile(ast.Module(body=[fn], type_ignores=[]), path, 'exec'), ns)
    parse = ns['_parse_ann_info']
    # the class's own default band: the literal assigned to self.default_thres in SemiCOCODataset.__init__ (semicoco.py:56)
    init = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == '__init__'][0]
    default_thres = [ast.literal_eval(n.value) for n in ast.walk(init) if isinstance(n, ast.Assign)
                     and isinstance(n.targets[0], ast.Attribute) and n.targets[0].attr == 'default_thres'][0]
    rng = np.random.RandomState(13)
    names = [f'cls{i}' for i in range(8)]
    cat2id = {n: i for i, n in enumerate(names)}
    tmp = tempfile.mkdtemp()
    W, H = 200, 120
    imgs = []
    for k in range(12):
        n = int(rng.poisson(4))
        x1, y1 = rng.randint(-10, W, n), rng.randint(-10, H, n)
        rects = np.stack([x1, y1, x1 + rng.randint(0, 80, n), y1 + rng.randint(0, 60, n)], 1).tolist()
        tags = [names[j] for j in rng.randint(0, 8, n)]
        scores = [round(float(v), 6) for v in rng.uniform(0.05, 0.99, n)]
        with open(os.path.join(tmp, f'im{k}.jpg.json'), 'w') as fh:
            json.dump(dict(imageName=f'im{k}.jpg', targetNum=n, rects=rects, tags=tags, masks=[[]] * n, scores=scores), fh)
        imgs.append(dict(rects=rects, tags=tags, scores=scores))
    thres_file = os.path.join(tmp, 'thres.json')
    thres = {names[i]: float(v) for i, v in zip((0, 1, 2, 4, 5), (0.3, 0.32, 0.35, 0.31, 0.33))}     # classes 3, 6, 7 unseen
    modes = []
    for mode in ('no_file', 'file'):
        if mode == 'file':
            json.dump(dict(thres=thres), open(thres_file, 'w'))
        self = types.SimpleNamespace(ann_path=tmp, thres=thres_file, default_thres=default_thres, labelmapper=dict(cat2id=cat2id))
        outs = []
        for k in range(12):
            a = parse(self, dict(filename=f'im{k}.jpg', width=W, height=H), None)
            outs.append(dict(bboxes=a['bboxes'].tolist(), labels=a['labels'].tolist(), ignore=a['bboxes_ignore'].tolist()))
        modes.append(dict(mode=mode, thres=thres if mode == 'file' else None, outs=outs))
    json.dump(dict(names=names, wh=[W, H], imgs=imgs, modes=modes, default_thres=default_thres),
              open(os.path.join(HERE, 'pseudo_split.json'), 'w'))
    print('wrote pseudo_split.json', sum(len(o['bboxes']) for o in modes[0]['outs']), sum(len(o['ignore']) for o in modes[0]['outs']))


def gen_parse_dets():
    """parse_det_results + the score sort of gen_save_json_dict (runner/hooks/unlabel_pred_hook.py:20-57), run from their
    own source text on random per-class detection arrays (bbox2result layout)."""
    import ast
    import json
    path = os.path.join(R.REF, 'mmdet/runner/hooks/unlabel_pred_hook.py')
    tree = ast.parse(open(path).read())
    fns = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ('parse_det_results', 'gen_save_json_dict')]
    ns = {}
    exec(compile(ast.Module(body=fns, type_ignores=[]), path, 'exec'), ns)
    rng = np.random.RandomState(17)
    cases = []
    for _ in range(4):
        results = []
        for c in range(6):
            k = int(rng.poisson(2))
            b = np.concatenate([rng.uniform(-3, 300, (k, 4)), rng.uniform(0.02, 0.99, (k, 1))], 1).astype(np.float32)
            results.append(b)
        out = ns['gen_save_json_dict'](dict(task_type='Det', infer_score_thre=0.1, result=results), None)['infer_results']
        cases.append(dict(results=[r.tolist() for r in results],
                          out=[dict(c=o['category_index'], score=o['score'], bbox=o['bbox']) for o in out]))
    json.dump(dict(score_thr=0.1, cases=cases), open(os.path.join(HERE, 'parse_dets.json'), 'w'))
    print('wrote parse_dets.json', [len(c['out']) for c in cases])


def _fuse_namespace():
    """save_results2file and its helpers compiled from the reference's own source text (runner/hooks/unlabel_pred_hook.py:20-171);
    mmcv.ops.nms (not in the tree) is the documented mmcv 1.3.10 behaviour restated here: keep scores > score_threshold, sort
    descending, greedy, suppress IoU > iou_threshold."""
    import ast
    import json
    path = os.path.join(R.REF, 'mmdet/runner/hooks/unlabel_pred_hook.py')
    tree = ast.parse(open(path).read())
    want = ('parse_det_results', 'gen_save_json_dict', 'create_dir', 'save_results2file')
    fns = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want]

    def nms(boxes, scores, iou_threshold, offset=0, score_threshold=0, max_num=-1):
        valid = np.nonzero(scores > np.float32(score_threshold))[0]
        b, sc = boxes[valid], scores[valid]
        order = np.argsort(-sc, kind='stable')
        keep = []
        for i in order:
            ok = True
            for j in keep:
                w = max(np.float32(0), min(b[i, 2], b[j, 2]) - max(b[i, 0], b[j, 0]))
                h = max(np.float32(0), min(b[i, 3], b[j, 3]) - max(b[i, 1], b[j, 1]))
                inter = np.float32(w * h)
                union = np.float32((b[i, 2] - b[i, 0]) * (b[i, 3] - b[i, 1]) + (b[j, 2] - b[j, 0]) * (b[j, 3] - b[j, 1]) - inter)
                with np.errstate(divide='ignore', invalid='ignore'):
                    if np.float32(inter / union) > np.float32(iou_threshold):
                        ok = False
                        break
            if ok:
                keep.append(i)
        keep = np.array(keep, dtype=np.int64)
        return np.concatenate([b[keep], sc[keep, None]], 1).reshape(-1, 5), valid[keep]

    ns = dict(os=os, json=json, np=np, nms=nms)
    exec(compile(ast.Module(body=fns, type_ignores=[]), path, 'exec'), ns)
    return ns


def _synth_dets(rng, k, C_):
    """Clusters of overlapping boxes so that the second NMS has work to do; fractional coordinates, scores around both
    thresholds."""
    ctr = rng.uniform(20, 300, (max(1, k // 4), 2))
    which = rng.randint(0, len(ctr), k)
    wh = rng.uniform(8, 90, (k, 2))
    c = ctr[which] + rng.normal(0, 6, (k, 2))
    boxes = np.concatenate([c - wh / 2, c + wh / 2], 1)
    scores = np.sort(rng.uniform(0.03, 0.99, k))[::-1]
    scores[rng.randint(0, k)] = 0.1           # exactly at infer_score_thre: kept by parse (>=), dropped by nms (>)
    scores = np.sort(scores)[::-1]
    dets = np.concatenate([boxes, scores[:, None]], 1).astype(np.float32)
    labels = rng.randint(0, C_, k)
    return dets, labels


def gen_fuse():
    """The label-file step of the refresh, save_results2file (runner/hooks/unlabel_pred_hook.py:84-171, fuse=False), run
    from its own source text on synthetic detector outputs."""
    import json
    import tempfile
    ns = _fuse_namespace()
    C_ = 6
    names = [f'cls{i}' for i in range(C_)] + ['background']
    id2cat = {str(i): n for i, n in enumerate(names)}
    cat2id = {n: i for i, n in enumerate(names)}
    rng = np.random.RandomState(23)
    cases = []
    for case in range(5):
        tmp = tempfile.mkdtemp()
        root = os.path.join(tmp, 'images')
        os.makedirs(root)
        k = int(rng.randint(5, 60))
        dets, labels = _synth_dets(rng, k, C_)
        result = [dets[labels == i] for i in range(C_)]          # bbox2result (core/bbox/transforms.py:99-116)
        json.dump(dict(imageName='a.jpg', targetNum=0, rects=[], tags=[], masks=[], scores=[]),
                  open(os.path.join(tmp, 'a.jpg.json'), 'w'))
        iou = [0.6, 0.5, 0.3, 0.6, 0.45][case]
        ns['save_results2file'](result, os.path.join(root, 'a.jpg'), 400, 400, 'json', 'iteration_1.pth', 0.1, id2cat, cat2id,
                                root, tmp, 'Det', anno_root_path=tmp, iou=iou, fuse=False, first_ignore=True)
        out = json.load(open(os.path.join(tmp, 'a.jpg.json')))
        cases.append(dict(dets=dets.tolist(), labels=labels.tolist(), iou=iou, rects=out['rects'],
                          tags=[cat2id[t] for t in out['tags']], scores=out['scores'], targetNum=out['targetNum']))
    json.dump(dict(infer_score_thre=0.1, nms_score_thr=0.1, id2cat=id2cat, cases=cases), open(os.path.join(HERE, 'fuse.json'), 'w'))
    print('wrote fuse.json', [(len(c['dets']), c['targetNum']) for c in cases])


def gen_rla():
    """RLA_ResNet of the reference (mmdet/models/backbones/resnet_rla.py, imported unmodified; `.flops = True` selects its
    CPU path for the initial h, :297-300) on a seeded 2 x 3 x 64 x 96 input with the synthetic weights of
    oracle/rla_oracle.py: the four stage outputs and the gradients of sum_i <out_i, r_i> w.r.t. every parameter that
    _freeze_stages leaves trainable."""
    import importlib
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import rla_oracle as RO
    R.install()
    mod = importlib.import_module('mmdet.models.backbones.resnet_rla')
    net = mod.RLA_ResNet(layers=[3, 4, 6, 3], frozen_stages=1, norm_eval=True, style='pytorch')
    net.flops = True
    sd = RO.synth_state_dict(0)
    bsd = {k[len('backbone.'):]: v for k, v in sd.items() if k.startswith('backbone.')}
    missing, unexpected = net.load_state_dict(bsd, strict=True), None
    net.train()
    g = torch.Generator().manual_seed(21)
    x = (torch.randn(2, 3, 64, 96, generator=g) * 40).bfloat16().float()
    outs = net(x)
    rs = [torch.randn(o.shape, generator=g) for o in outs]
    sum((o * r).sum() for o, r in zip(outs, rs)).backward()
    named = dict(net.named_parameters())
    tk = [k for k, p in named.items() if p.requires_grad]
    assert sorted('backbone.' + k for k in tk) == sorted(k for k in RO.trainable_keys(sd) if k.startswith('backbone.')), 'trainable sets differ'
    out = dict(x=x.numpy(), n_train=len(tk))
    for i, (o, r) in enumerate(zip(outs, rs)):
        out[f'out{i}'] = o.detach().numpy()
        out[f'r{i}'] = r.numpy()
    keys = sorted(tk)
    out['grad_keys'] = np.array(['backbone.' + k for k in keys])
    out['grad_norms'] = np.array([float(named[k].grad.norm()) for k in keys])
    # full gradients of a few representative parameters (first / middle / last stage, every kind of layer)
    for k in ('stages.1.0.conv1.weight', 'stages.1.0.bn1.weight', 'stages.1.0.bn3.bias', 'stages.1.0.downsample.0.weight',
              'stages.1.0.downsample.1.weight', 'stages.2.3.conv2.weight', 'stages.3.2.conv3.weight', 'stages.3.1.bn2.bias',
              'conv_outs.1.weight', 'recurrent_convs.2.weight', 'stage_bns.1.2.weight', 'stage_bns.3.0.bias', 'conv_outs.3.weight'):
        out['grad:backbone.' + k] = named[k].grad.numpy()
    np.savez_compressed(os.path.join(HERE, 'rla_tiny.npz'), **out)
    print('wrote rla_tiny.npz', [tuple(o.shape) for o in outs], len(tk), 'trainable tensors')


def gen_patch_shuffle():
    """PatchShuffle (mmdet/datasets/pipelines/transforms.py:2143-2248) run from the reference file itself: the class and its
    helper are compiled out of the module's source (the module imports cv2 / torchvision / imgaug, none installed) with numpy
    stand-ins for the three image calls it makes (mmcv.imcrop with inclusive corners, cv2.hconcat / vconcat)."""
    import ast
    import json
    import random
    import types
    path = '/root/reference/mmdet/datasets/pipelines/transforms.py'
    src = open(path).read()
    tree = ast.parse(src)
    keep = [n for n in tree.body if (isinstance(n, ast.FunctionDef) and n.name == 'get_bbox_fields') or
            (isinstance(n, ast.ClassDef) and n.name == 'PatchShuffle')]
    for n in keep:
        n.decorator_list = []
    mod = ast.Module(body=keep, type_ignores=[])
    mmcv = types.SimpleNamespace(imcrop=lambda img, b: img[int(b[1]):int(b[3]) + 1, int(b[0]):int(b[2]) + 1])
    cv2 = types.SimpleNamespace(hconcat=lambda xs: np.concatenate(xs, 1), vconcat=lambda xs: np.concatenate(xs, 0))
    g = dict(np=np, random=random, mmcv=mmcv, cv2=cv2)
    exec(compile(mod, path, 'exec'), g)
    PS = g['PatchShuffle']
    cases = []
    rng = np.random.RandomState(77)
    for ci in range(24):
        np.random.seed(1000 + ci)
        random.seed(2000 + ci)
        h, w = int(rng.randint(12, 40)), int(rng.randint(12, 40))
        img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        n, k = int(rng.randint(0, 6)), int(rng.randint(0, 3))
        def boxes(m):
            x1, y1 = rng.uniform(0, w - 2, m), rng.uniform(0, h - 2, m)
            x2, y2 = x1 + rng.uniform(1, w / 2, m), y1 + rng.uniform(1, h / 2, m)
            return np.stack([x1, y1, np.minimum(x2, w), np.minimum(y2, h)], 1).astype(np.float32).reshape(-1, 4)
        res = dict(img=img.copy(), gt_bboxes=boxes(n), gt_labels=rng.randint(0, 80, n).astype(np.int64),
                   gt_bboxes_ignore=boxes(k), bbox_fields=['gt_bboxes_ignore'])
        inp = {kk: (v.tolist() if isinstance(v, np.ndarray) else v) for kk, v in res.items() if kk not in ('bbox_fields', 'img')}
        inp['img_shape'], inp['img_hex'] = [h, w, 3], img.tobytes().hex()
        t = PS(ratio=0.9 if ci % 6 else 1.0, ranges=[0.0, 1.0] if ci % 3 else [0.2, 0.8], mode=['flip', 'flop'])
        out = t(res)
        cases.append(dict(seed=(1000 + ci, 2000 + ci), ratio=t.ratio, ranges=t.ranges, input=inp,
                          output=dict(PS=bool(out['PS']), PS_place=None if out['PS_place'] is None else float(out['PS_place']),
                                      PS_mode=out['PS_mode'], img_hex=np.ascontiguousarray(out['img']).tobytes().hex(), gt_bboxes=np.asarray(out['gt_bboxes']).reshape(-1, 4).tolist(),
                                      gt_labels=np.asarray(out['gt_labels']).tolist(),
                                      gt_bboxes_ignore=np.asarray(out['gt_bboxes_ignore']).reshape(-1, 4).tolist())))
    json.dump(cases, open(os.path.join(HERE, 'patch_shuffle.json'), 'w'))
    print('wrote patch_shuffle.json', len(cases), 'cases,', sum(c['output']['PS'] for c in cases), 'shuffled')


def gen_ubaug():
    """UBAug's image arithmetic (mmdet/datasets/pipelines/transforms.py:2098-2140).  torchvision is not installable here; on PIL
    images its ColorJitter / RandomGrayscale / the class's GaussianBlur are thin wrappers of Pillow calls (torchvision
    transforms/functional_pil.py: adjust_brightness = ImageEnhance.Brightness(img).enhance(f), adjust_contrast / adjust_saturation
    likewise with Contrast / Color, adjust_hue = convert('HSV'), H += uint8(f * 255), convert back; to_grayscale = convert('L')
    merged three times; transforms.py:1934-1937: img.filter(ImageFilter.GaussianBlur(radius=sigma))) - Pillow itself IS in the build
    container, so the fixture holds Pillow's own outputs for the parameter ranges UBAug draws from."""
    from PIL import Image, ImageEnhance, ImageFilter
    rng = np.random.RandomState(11)
    rgb = rng.randint(0, 256, (45, 61, 3)).astype(np.uint8)
    rgb[:6] = rgb[:6, :, :1]                   # a band of grey pixels (saturation 0)
    rgb[6:9] = (rgb[6:9] // 32) * 32           # and a band with few levels
    im = Image.fromarray(rgb)
    out = dict(rgb=rgb)
    fac = dict(brightness=[0.6, 0.83, 1.0, 1.4], contrast=[0.6, 1.17, 1.4], saturation=[0.6, 0.99, 1.4], hue=[-0.1, -0.031, 0.0, 0.07, 0.1],
               blur=[0.1, 0.35, 0.8, 1.0, 1.234, 1.5, 2.0])
    for i, f in enumerate(fac['brightness']):
        out[f'brightness_{i}'] = np.asarray(ImageEnhance.Brightness(im).enhance(f))
    for i, f in enumerate(fac['contrast']):
        out[f'contrast_{i}'] = np.asarray(ImageEnhance.Contrast(im).enhance(f))
    for i, f in enumerate(fac['saturation']):
        out[f'saturation_{i}'] = np.asarray(ImageEnhance.Color(im).enhance(f))
    for i, f in enumerate(fac['hue']):
        h, s_, v = im.convert('HSV').split()
        nh = np.array(h, dtype=np.uint8)
        with np.errstate(over='ignore'):
            nh += np.array(f * 255).astype(np.uint8) if f >= 0 else np.uint8(int(f * 255) & 255)
        out[f'hue_{i}'] = np.asarray(Image.merge('HSV', (Image.fromarray(nh, 'L'), s_, v)).convert('RGB'))
    g = np.asarray(im.convert('L'))
    out['gray'] = np.dstack([g, g, g])
    for i, f in enumerate(fac['blur']):
        out[f'blur_{i}'] = np.asarray(im.filter(ImageFilter.GaussianBlur(radius=f)))
    for k, v in fac.items():
        out['f_' + k] = np.asarray(v, np.float64)
    save('ubaug_pil.npz', **out)


def gen_randaug():
    """RandAug's histogram / filter ops the reference's colour branch draws for an image without boxes (mmdet/datasets/pipelines/
    autoaug_fast.py:219-224 AutoContrast / Equalize, :244-250 Posterize, :371-372 Solarize, :407 Sharpness; semi_aug.py:464-477,
    494-497).  autoaug_fast.py's TransformT wrappers are one-line calls of PIL.ImageOps / ImageEnhance on ToPILImage(img) with the
    level -> parameter maps restated in oracle.datapath_oracle.randaug_level; Pillow itself is in the build container, so the fixture
    holds Pillow's own outputs for every level 1 .. 9 the policy draws (np.random.randint(1, magnitude = 10))."""
    from PIL import Image, ImageEnhance, ImageOps
    rng = np.random.RandomState(23)
    y, x = np.mgrid[0:57, 0:74]
    imgs = [rng.randint(0, 256, (45, 61, 3)).astype(np.uint8),                                                   # full-range noise
            np.clip(np.stack([60 + x * 1.7 + y * 0.3, 90 + 50 * np.sin(x / 6.0), 40 + y * 2.1], -1)
                    + rng.randint(-9, 9, (57, 74, 3)), 0, 255).astype(np.uint8),                                 # narrow, smooth histograms
            np.stack([np.full((31, 40), 17), rng.randint(100, 102, (31, 40)), (rng.randint(0, 256, (31, 40)) // 64) * 64], -1).astype(np.uint8),
            rng.randint(0, 256, (3, 3, 3)).astype(np.uint8)]                                                     # one interior pixel
    out = dict(n_img=np.int64(len(imgs)))
    for k, a in enumerate(imgs):
        im = Image.fromarray(a)
        out[f'img{k}'] = a
        out[f'autocontrast{k}'] = np.asarray(ImageOps.autocontrast(im))
        out[f'equalize{k}'] = np.asarray(ImageOps.equalize(im))
        for level in range(1, 10):
            out[f'solarize{k}_{level}'] = np.asarray(ImageOps.solarize(im, 256 - int(level * 256 / 10)))
            out[f'posterize{k}_{level}'] = np.asarray(ImageOps.posterize(im, 4 - int(level * 4 / 10)))
            out[f'sharpness{k}_{level}'] = np.asarray(ImageEnhance.Sharpness(im).enhance(float(level) * 1.8 / 10 + .1))
    save('randaug_pil.npz', **out)


def gen_fuse_history():
    """save_results2file with fuse=True (:131-141): the previous contents of the image's label file join the new detections
    before the per-class NMS.  Three rounds per case on one file, as the hook runs them (:470-510): round 1 with
    first_ignore (first_fuse=False: the initial labels are dropped), rounds 2 and 3 fuse with what the round before wrote."""
    import json
    import tempfile
    ns = _fuse_namespace()
    C_ = 6
    names = [f'cls{i}' for i in range(C_)] + ['background']
    id2cat = {str(i): n for i, n in enumerate(names)}
    cat2id = {n: i for i, n in enumerate(names)}
    rng = np.random.RandomState(41)
    cases = []
    for case in range(4):
        tmp = tempfile.mkdtemp()
        root = os.path.join(tmp, 'images')
        os.makedirs(root)
        # initial labels of the file (the first training stage's predictions): fractional boxes, some scores under 0.1
        k0 = int(rng.randint(3, 20))
        d0, l0 = _synth_dets(rng, k0, C_)
        init = dict(imageName='a.jpg', targetNum=k0, rects=d0[:, :4].astype(np.float64).round(3).tolist(),
                    tags=[names[i] for i in l0], masks=[[] for _ in range(k0)], scores=d0[:, 4].astype(np.float64).round(6).tolist())
        json.dump(init, open(os.path.join(tmp, 'a.jpg.json'), 'w'))
        iou = [0.6, 0.5, 0.3, 0.45][case]
        first_ignore = case % 2 == 0
        rounds = []
        for rd in range(3):
            k = int(rng.randint(5, 50))
            dets, labels = _synth_dets(rng, k, C_)
            result = [dets[labels == i] for i in range(C_)]
            old = json.load(open(os.path.join(tmp, 'a.jpg.json')))
            ns['save_results2file'](result, os.path.join(root, 'a.jpg'), 400, 400, 'json', 'iteration_1.pth', 0.1, id2cat, cat2id,
                                    root, tmp, 'Det', anno_root_path=tmp, iou=iou, fuse=True, first_ignore=(first_ignore and rd == 0))
            out = json.load(open(os.path.join(tmp, 'a.jpg.json')))
            rounds.append(dict(dets=dets.tolist(), labels=labels.tolist(), first_ignore=bool(first_ignore and rd == 0),
                               old_rects=old['rects'], old_tags=[cat2id[t] for t in old['tags']], old_scores=old['scores'],
                               rects=out['rects'], tags=[cat2id[t] for t in out['tags']], scores=out['scores'],
                               targetNum=out['targetNum']))
        cases.append(dict(iou=iou, rounds=rounds))
    json.dump(dict(infer_score_thre=0.1, nms_score_thr=0.1, id2cat=id2cat, cases=cases),
              open(os.path.join(HERE, 'fuse_hist.json'), 'w'))
    print('wrote fuse_hist.json', [[(len(r['old_scores']), len(r['dets']), r['targetNum']) for r in c['rounds']] for c in cases])


if __name__ == '__main__':
    if sys.argv[1:] == ['ubaug']:
        gen_ubaug()
        sys.exit(0)
    if sys.argv[1:] == ['randaug']:
        gen_randaug()
        sys.exit(0)
    if sys.argv[1:] == ['ps']:
        gen_patch_shuffle()
        sys.exit(0)
    if sys.argv[1:] == ['parse_dets']:
        gen_parse_dets()
        sys.exit(0)
    if sys.argv[1:] == ['rla']:
        gen_rla()
        sys.exit(0)
    if sys.argv[1:] == ['fuse']:
        gen_fuse()
        sys.exit(0)
    if sys.argv[1:] == ['fuse_hist']:
        gen_fuse_history()
        sys.exit(0)
    if sys.argv[1:] == ['adathres']:
        gen_adathres()
        sys.exit(0)
    if sys.argv[1:] == ['pseudo_split']:
        gen_pseudo_split()
        sys.exit(0)
    model = R.build_fcos(SUP_CFG)
    model.train()
    which = sys.argv[1:] or ['assign', 'loss', 'bboxes', 'net']
    if 'assign' in which:
        gen_assign(model)
    if 'loss' in which:
        gen_loss(model)
    if 'bboxes' in which:
        gen_bboxes(model)
    if 'net' in which:
        gen_net(model)
