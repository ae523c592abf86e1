"""Loader-side contract of the semi-supervised loop and a synthetic implementation of it.

The reference couples the pseudo-label refresh to the DataLoader through files and prefetch depth: UnlabelPredHook
keeps its own iterator over the unlabeled sampler's index order (`runner.ITER`, sampler_seed.py:7-14), advanced
`preload` = 2*workers+2 steps ahead of the training loop, so that an image's annotation file is rewritten just before
a loader worker reads it (unlabel_pred_hook.py:455-469,512-562; datasets/builder.py:159-352).  Here the same hand-off
is an explicit, deterministic interface between the hook and the loader:

  loader.unlabeled            -> an UnlabeledSource
    .names                       all unlabeled image names                      (hook's image_list)
    .upcoming(k)                 names this rank consumes k iterations from now  (replaces runner.ITER + preload)
    .test_view(name)             (img [1,3,H,W], img_meta) for the teacher       (config data.unlabel_pred.pipeline)
  loader.bank                 -> the PseudoLabelBank the loader reads an unlabeled sample's annotations from when it
                                 assembles a batch (SemiCOCODataset._parse_ann_info reading the JSON, semicoco.py:232-264)

Real datasets implement the same three methods over their decode / augment pipeline; the metric of this repository
uses synthetic inputs (BASELINE.json), so the implementation here is `SyntheticSemiLoader`: COCO-shaped random boxes
(SURVEY.md §8d), one labeled + one unlabeled image per batch (batch_config ratio [[1, 1]], configs/fcos_semi/RLA_*.py).
"""
import numpy as np
import torch


def mark_ready(img, stream=None, event=None):
    """Attaches the producer's event to a batch image tensor: recorded now on `stream` (default: the current stream).  A loader
    that renders batches on its own stream calls this after its last kernel; FCOS.forward_train then starts the frozen prefix of
    the forward pass (image layout, stem, layer1) behind THAT event instead of behind everything queued on the training
    stream - i.e. under the tail of the previous step's backward pass (FCOS.pipeline_prefix).
    The mark is ONE-SHOT and belongs to this tensor object: forward_train consumes it (a buffer that is refilled in place needs a
    new mark_ready per batch - without one the prefix is ordered behind the caller's stream, which is always correct), and a
    view or copy of the tensor (.contiguous(), slicing, append_half_scale's cat) does not carry it."""
    if img.is_cuda:
        if event is None:
            event = torch.cuda.Event()
            event.record(stream if stream is not None else torch.cuda.current_stream(img.device))
        img._dsl_ready = event          # `event`: one the producer recorded itself (a resident batch: once, when it was written)
    return img


def synth_boxes(rng, n, H=800, W=1333, lo=16.0, hi=600.0):
    cx, cy = rng.uniform(0, W, n), rng.uniform(0, H, n)
    w = np.exp(rng.uniform(np.log(lo), np.log(min(hi, W)), n))
    h = np.exp(rng.uniform(np.log(lo), np.log(min(hi, H)), n))
    b = np.stack([np.clip(cx - w / 2, 0, W), np.clip(cy - h / 2, 0, H), np.clip(cx + w / 2, 0, W),
                  np.clip(cy + h / 2, 0, H)], 1).astype('float32')
    return b[((b[:, 2] - b[:, 0]) >= 1) & ((b[:, 3] - b[:, 1]) >= 1)]


class SyntheticUnlabeled:
    """UnlabeledSource over a fixed set of synthetic images (deterministic per name)."""

    def __init__(self, loader):
        self._l = loader
        self.names = [f'unlabeled_{i:05d}.jpg' for i in range(loader.n_unlabeled)]
        self.prefetch_depth = 0        # the synthetic loader assembles a batch when it is asked for: labels can be as fresh as the previous iteration

    def upcoming(self, k=0):
        """Unlabeled names of the batch that will be produced k batches after the next one."""
        l = self._l
        pos = l._pos + k
        if pos >= len(l):
            return []
        return [self.names[l._order(l._epoch)[pos % l.n_unlabeled]]]

    def image(self, name):
        """Deterministic per name; generated once and kept on the loader's device (the loop then does no host-side image work)."""
        c = self._l._cache
        if name not in c:
            i = self.names.index(name)
            g = torch.Generator().manual_seed(7919 * (i + 1) + self._l.seed)
            c[name] = (torch.randn(3, self._l.H, self._l.W, generator=g) * self._l.img_std).bfloat16().float().to(self._l.device)
        return c[name]

    def test_view(self, name):
        l = self._l
        meta = dict(filename=name, ori_filename=name, ori_shape=(l.H, l.W_img, 3), img_shape=(l.H, l.W_img, 3),
                    pad_shape=(l.H, l.W, 3), scale_factor=np.ones(4, dtype=np.float32), flip=False)
        return self.image(name)[None], meta


class SyntheticSemiLoader:
    """One labeled + one unlabeled image per batch.  The unlabeled image's annotations come from `bank` at the moment
    the batch is assembled (gt boxes above the class threshold, ignore boxes in the band below it)."""

    def __init__(self, bank, n_labeled=8, n_unlabeled=8, iters_per_epoch=None, H=800, W=1344, W_img=None, seed=0,
                 device='cuda', img_std=1.0, rank=0):
        self.bank, self.n_labeled, self.n_unlabeled = bank, n_labeled, n_unlabeled
        self.H, self.W, self.W_img = H, W, (W_img if W_img is not None else W)
        self.seed, self.device, self.img_std, self.rank = seed + 1000 * rank, device, img_std, rank
        self._len = iters_per_epoch or max(n_labeled, n_unlabeled)
        self._epoch, self._pos = 0, 0
        self._cache = {}
        self.CLASSES = tuple(bank.class_names) if bank.class_names else tuple(f'class_{i}' for i in range(bank.num_classes))
        self.unlabeled = SyntheticUnlabeled(self)
        rng = np.random.RandomState(2024 + self.seed)
        self._gt = []
        for _ in range(n_labeled):
            b = synth_boxes(rng, int(np.clip(rng.poisson(7), 1, 40)), H=H, W=self.W_img, lo=min(16.0, H / 8), hi=min(600.0, H))
            self._gt.append((torch.from_numpy(b), torch.from_numpy(rng.randint(0, bank.num_classes, len(b)).astype('int64'))))

    def __len__(self):
        return self._len

    def _order(self, epoch):
        return np.random.RandomState(self.seed + 31 * epoch).permutation(self.n_unlabeled)

    def set_epoch(self, epoch):
        self._epoch = epoch

    def __iter__(self):
        self._pos = 0
        return self

    def __next__(self):
        if self._pos >= self._len:
            self._epoch += 1
            raise StopIteration
        i = self._pos
        li = (self._epoch * self._len + i) % self.n_labeled
        uname = self.unlabeled.upcoming(0)[0]
        lname = f'labeled_{li:05d}.jpg'
        if lname not in self._cache:
            g = torch.Generator().manual_seed(104729 * (li + 1) + self.seed)
            self._cache[lname] = (torch.randn(3, self.H, self.W, generator=g) * self.img_std).bfloat16().float().to(self.device)
        img = torch.stack([self._cache[lname], self.unlabeled.image(uname)])
        ugt, ugl, uig = self.bank.ann_info(uname, img_wh=(self.W_img, self.H))
        gtb, gtl = self._gt[li]
        metas = [dict(filename=n, ori_filename=n, ori_shape=(self.H, self.W_img, 3), img_shape=(self.H, self.W_img, 3),
                      pad_shape=(self.H, self.W, 3), scale_factor=np.ones(4, dtype=np.float32), flip=False)
                 for n in (lname, uname)]
        self._pos += 1
        return dict(img=img, img_metas=metas, gt_bboxes=[gtb, ugt], gt_labels=[gtl, ugl],
                    gt_bboxes_ignore=[torch.zeros(0, 4), uig])


class SyntheticValLoader:
    """Validation-side loader contract (what EvalHook / single_gpu_test consume): batches dict(img, img_metas) in dataset
    order, rank-strided like DistributedSampler(shuffle=False); the dataset attributes the COCO json export and the built-in
    evaluator need (`img_ids`, `cat_ids`, `annotations`, `num_images`)."""

    def __init__(self, n_images=4, num_classes=80, H=800, W=1344, W_img=None, seed=0, device='cuda', samples_per_gpu=1,
                 rank=0, world_size=1, cat_ids=None):
        self.num_images, self.H, self.W, self.W_img = n_images, H, W, (W_img if W_img is not None else W)
        self.device, self.bs, self.rank, self.world = device, samples_per_gpu, rank, world_size
        self.img_ids = [1000 + i for i in range(n_images)]
        self.cat_ids = list(cat_ids) if cat_ids is not None else list(range(1, num_classes + 1))
        rng = np.random.RandomState(4242 + seed)
        self.annotations, self._imgs, self.seed = [], {}, seed
        for _ in range(n_images):
            b = synth_boxes(rng, int(np.clip(rng.poisson(7), 1, 40)), H=H, W=self.W_img, lo=min(16.0, H / 8), hi=min(600.0, H))
            lab = rng.randint(0, num_classes, len(b))
            self.annotations.append([dict(bbox=[float(x[0]), float(x[1]), float(x[2] - x[0]), float(x[3] - x[1])],
                                          category_id=self.cat_ids[int(l)], iscrowd=0) for x, l in zip(b, lab)])
        self.dataset = self

    def __len__(self):
        mine = len(range(self.rank, self.num_images, self.world))
        return (mine + self.bs - 1) // self.bs

    def _image(self, i):
        if i not in self._imgs:
            g = torch.Generator().manual_seed(15485863 * (i + 1) + self.seed)
            self._imgs[i] = torch.randn(3, self.H, self.W, generator=g).bfloat16().float().to(self.device)
        return self._imgs[i]

    def __iter__(self):
        idx = list(range(self.rank, self.num_images, self.world))
        for k in range(0, len(idx), self.bs):
            chunk = idx[k:k + self.bs]
            metas = [dict(filename=f'val_{i:05d}.jpg', ori_filename=f'val_{i:05d}.jpg', ori_shape=(self.H, self.W_img, 3),
                          img_shape=(self.H, self.W_img, 3), pad_shape=(self.H, self.W, 3),
                          scale_factor=np.ones(4, dtype=np.float32), flip=False) for i in chunk]
            yield dict(img=[torch.stack([self._image(i) for i in chunk])], img_metas=[metas])   # MultiScaleFlipAug's one-element lists
