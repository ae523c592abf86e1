"""Data-parallel path with world_size 2 on CPU (gloo): parameter broadcast, bucketed gradient all-reduce
over the flat buffer in backward order, the 2-float (num_pos, sum ctr) reduce_mean, log-var averaging."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from util import fcos_model_cfg


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _NoOps:
    def run(self):
        pass


def _worker(rank, world, port, q, deferred=False):
    try:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group('gloo', rank=rank, world_size=world)
        from dsl_amd import detectors  # noqa: F401
        from dsl_amd.parallel import HipDistributedDataParallel
        from dsl_amd.registry import build_detector
        torch.manual_seed(rank)
        det = build_detector(fcos_model_cfg())
        with torch.no_grad():
            det.store.train.add_(float(rank))              # make the replicas differ before wrapping
        ddp = HipDistributedDataParallel(det)
        ref0 = det.store.train.clone()
        lst = [torch.zeros_like(ref0) for _ in range(world)]
        dist.all_gather(lst, ref0)
        assert all(torch.equal(t, lst[0]) for t in lst), 'parameters not broadcast from rank 0'
        # bucketed all-reduce in backward order
        det.store.grad.copy_(torch.arange(det.store.n_train, dtype=torch.float32) % 7 + rank)

        # the engine's schedule: segment s queues the weight gradients of bucket s on the side stream and records named
        # event s; the bucket's all-reduce is issued behind that event
        _b = det.store.grad_buckets()

        class Plan:
            bwd_segments = [(_NoOps(), dict(bucket=b, slot=i, main=(i == 3))) for i, b in enumerate(_b)]
        if deferred:
            # the deferred head update's list structure (engine.Plan.defer): the head + FPN list completes no bucket, bucket 0
            # follows the last segment - every rank must issue the collectives in THIS order
            Plan.bwd_segments = ([(_NoOps(), dict(bucket=None, slot=None, main=False))] + Plan.bwd_segments[1:]
                                 + [(_NoOps(), dict(bucket=_b[0], slot=0, main=False, deferred=True))])
        det._run_backward(Plan)
        assert len(det._pending) == 4
        done = [i['bucket'] for i in det._last_bwd_infos if i['bucket'] is not None]
        assert done == ([_b[1], _b[2], _b[3], _b[0]] if deferred else list(_b))
        assert sorted(done) == sorted(_b) and sum(hi - lo for lo, hi in done) == det.store.n_train      # the buckets tile the buffer
        det.wait_grads()
        expect = (torch.arange(det.store.n_train, dtype=torch.float32) % 7) * world + sum(range(world))
        assert torch.equal(det.store.grad, expect)
        # p.grad views see the reduced values
        named = dict(det.named_parameters())
        k = 'bbox_head.conv_cls.bias'
        off = det.store.train_regions['head.cls_b'][0]
        assert torch.equal(named[k].grad, expect[off:off + 80])
        # reduce_mean of (num_pos, sum centerness): the loss kernel applies max(sum * inv_world, floor)
        stats = torch.tensor([3.0 + rank, 1.5 * (rank + 1)])
        dist.all_reduce(stats)
        assert stats[0].item() * (1.0 / world) == pytest.approx(3.0 + (world - 1) / 2) and stats[1].item() / world == pytest.approx(1.5 * (world + 1) / 2)
        # log vars averaged over ranks with one collective
        losses = dict(loss_cls=torch.tensor(1.0 + rank), loss_bbox=torch.tensor(2.0), loss_centerness=torch.tensor(0.5))
        total, log = det._parse_losses(losses)
        assert log['loss_cls'] == pytest.approx(1.0 + (world - 1) / 2) and log['loss'] == pytest.approx(3.5 + (world - 1) / 2)
        assert float(total) == pytest.approx(3.5 + rank)
        dist.destroy_process_group()
        q.put((rank, 'ok'))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))


def _run(world, deferred):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, deferred)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=300) for _ in procs]
    finally:
        for p in procs:
            p.join(60)
            if p.is_alive():
                p.kill()
    assert all(r[1] == 'ok' for r in res), res


def test_ddp_world2_gloo():
    _run(2, False)


def test_ddp_world4_gloo_deferred_bucket_order():
    """Four ranks, the bucket order of the deferred head update (layer4, layer3, layer2, then head + FPN): every rank issues the
    same collectives in the same order and ends with the same sums."""
    _run(4, True)
