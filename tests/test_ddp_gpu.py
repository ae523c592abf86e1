"""The data-parallel training step with the REAL engine: two ranks (two processes sharing cuda:0, gloo collectives on
device tensors) run forward + loss + hand-written backward with the bucketed, event-ordered gradient all-reduce, and
the result must equal ONE process training on the concatenated batch (DDP averaging + reduce_mean normalisers ==
the bigger batch)."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from util import fcos_model_cfg, rel_l2

pytestmark = pytest.mark.gpu
H, W = 128, 192


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def make_batch(rank):
    """Two images per rank; rank r's images are images [2r, 2r+1] of the 4-image global batch."""
    from oracle import fcos_oracle as O
    g = torch.Generator().manual_seed(77)
    img = (torch.randn(4, 3, H, W, generator=g) * 30).bfloat16().float()
    rng = np.random.RandomState(9)
    gtb = [torch.from_numpy(O.synth_boxes(rng, 3, H=H, W=W, lo=8, hi=100)) for _ in range(4)]
    gtl = [torch.from_numpy(rng.randint(0, 80, len(b)).astype('int64')) for b in gtb]
    metas = [dict(img_shape=(H, W, 3), pad_shape=(H, W, 3), scale_factor=1.0)] * 4
    sl = slice(0, 4) if rank is None else slice(2 * rank, 2 * rank + 2)
    return dict(img=img[sl].cuda(), img_metas=metas[sl], gt_bboxes=gtb[sl], gt_labels=gtl[sl])


def build():
    from dsl_amd import detectors  # noqa: F401
    from dsl_amd.registry import build_detector
    from oracle import fcos_oracle as O
    model = build_detector(fcos_model_cfg())
    model.load_state_dict(O.synth_state_dict(0))
    return model.cuda()


def _worker(rank, world, port, q, eager, out_path, backend='gloo', comm='torch'):
    try:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        one_gpu_per_rank = backend in ('nccl', 'gloo_nccl_devices')     # nccl (= RCCL): one GPU per rank
        torch.cuda.set_device(rank if one_gpu_per_rank else 0)
        dist.init_process_group('nccl' if backend == 'nccl' else 'gloo', rank=rank, world_size=world)
        from dsl_amd.parallel import HipDistributedDataParallel
        model = build()
        model.eager_backward = eager                 # must be the same on every rank: it changes the collective order
        ddp = HipDistributedDataParallel(model, comm=comm)
        assert (model.rccl is not None) == (comm == 'rccl')
        out = ddp.train_step(make_batch(rank), None)
        out['loss'].backward()
        model.wait_grads()
        torch.cuda.synchronize()
        g = model.store.grad.detach().cpu()
        lst = [torch.zeros_like(g) for _ in range(world)]
        dist.all_gather(lst, g)
        same = all(torch.equal(t, lst[0]) for t in lst)
        logs = {k: float(v) for k, v in out['log_vars'].items()}
        dist.destroy_process_group()
        if rank == 0:
            torch.save(g, out_path)          # (a tensor in the queue would die with this process)
        q.put((rank, 'ok', same, None, logs))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc(), False, None, None))


def _ddp_vs_big_batch(eager, tmp_path, backend, comm='torch'):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    out_path = str(tmp_path / 'g_ddp.pt')
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, eager, out_path, backend, comm)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=150) for _ in procs]
    finally:
        for p in procs:
            p.join(20)
            if p.is_alive():
                p.kill()
    assert all(r[1] == 'ok' for r in res), [r[1] for r in res]
    assert all(r[2] for r in res), 'ranks hold different gradients after the all-reduce'
    g_ddp = torch.load(out_path)
    logs = [r[4] for r in res]
    assert logs[0] == pytest.approx(logs[1], rel=1e-6)                 # log vars are averaged over the ranks
    # single process, the concatenated 4-image batch
    model = build()
    out = model.train_step(make_batch(None), None)
    out['loss'].backward()
    torch.cuda.synchronize()
    g_big = model.store.grad.detach().cpu()
    assert torch.isfinite(g_ddp).all() and float(g_big.abs().max()) > 0
    # two bf16-storage computations of the same gradient (2+2 images vs 4) decorrelate to the bf16 noise floor of
    # the deep layers; the predictors, one conv away from the fp32 loss, pin the semantics
    assert rel_l2(g_ddp, g_big) < 0.10, rel_l2(g_ddp, g_big)
    reg = model.store.train_regions
    for name in ('head.cls_w', 'head.cls_b', 'head.regctr_w', 'head.regctr_b'):
        o, n = reg[name][:2]
        assert rel_l2(g_ddp[o:o + n], g_big[o:o + n]) < 2e-2, (name, rel_l2(g_ddp[o:o + n], g_big[o:o + n]))
    for k, v in out['log_vars'].items():
        assert logs[0][k] == pytest.approx(float(v), rel=2e-3), (k, logs[0][k], float(v))


@pytest.mark.parametrize('eager', [False, True])
def test_ddp_step_equals_big_batch(eager, tmp_path):
    _ddp_vs_big_batch(eager, tmp_path, 'gloo')


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='RCCL needs one GPU per rank: this box has fewer than 2')
def test_ddp_step_equals_big_batch_rccl(tmp_path):
    """The same check with backend "nccl" (= RCCL over xGMI) and one GPU per rank: the bucketed all-reduces run on the
    communication stream behind the named events, the per-bucket optimizer path is NOT involved (gradients are compared).
    Skipped on 1-GPU boxes; on a multi-GPU node it is RCCL's first contact with this code before the scaling bench."""
    _ddp_vs_big_batch(True, tmp_path, 'nccl')


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='RCCL needs one GPU per rank: this box has fewer than 2')
def test_ddp_step_equals_big_batch_cabi_communicator(tmp_path):
    """The same check with the exchanges carried by the C-ABI's own rcclComm_t (dsl_comm_init_rank, dsl_allreduce_bucket) -
    torch.distributed (gloo) only hands rank 0's unique id to rank 1."""
    _ddp_vs_big_batch(True, tmp_path, 'gloo_nccl_devices', comm='rccl')


def _one_rank_comm(port, q):
    try:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1')
        torch.cuda.set_device(0)
        dist.init_process_group('gloo', rank=0, world_size=1)
        from dsl_amd import _lib as L
        from dsl_amd.parallel import RcclComm, StreamWork
        comm = RcclComm()
        assert L.lib.dsl_comm_size(comm.comm) == 1
        g = torch.Generator(device='cuda').manual_seed(0)
        buf = torch.randn(1 << 20, device='cuda', generator=g)
        want = buf.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        comm.all_reduce(buf[4096:8192 + 4096], side)           # a bucket = a contiguous range of the flat buffer
        comm.all_reduce(buf[:2], side)                         # the (num_pos, sum centerness) pair
        w = StreamWork(side)
        w.wait()
        out = buf * 1.0                                        # on the current stream, ordered behind the collective
        torch.cuda.synchronize()
        same = torch.equal(out, want)                          # one rank: the sum over the ranks is the operand
        # error path: a null communicator is refused with a message, not a crash
        rc = L.lib.dsl_allreduce_bucket(None, L.ptr(buf), 16, L.stream_ptr())
        msg = L.lib.dsl_last_error().decode()
        comm.close()
        dist.destroy_process_group()
        q.put(('ok', same, rc, msg))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((traceback.format_exc(), False, 0, ''))


def test_cabi_communicator_one_rank():
    """dsl_comm_unique_id / dsl_comm_init_rank / dsl_allreduce_bucket on this box's one GPU: librccl is bound at the first
    call, a one-rank communicator comes up, all-reduces queued on a side stream leave the operand unchanged and order a
    StreamWork waiter behind them.  (Two ranks need two GPUs: test_ddp_step_equals_big_batch_cabi_communicator.)"""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_one_rank_comm, args=(_free_port(), q))
    p.start()
    try:
        status, same, rc, msg = q.get(timeout=240)
    finally:
        p.join(20)
        if p.is_alive():
            p.kill()
    assert status == 'ok', status
    assert same
    assert rc != 0 and 'null' in msg


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='RCCL needs one GPU per rank: this box has fewer than 2')
def test_bench_two_rank_rccl():
    """`bench.py --gpus 2` exactly as the driver launches it (nccl, one GPU per rank), with the per-bucket communication
    trace in the JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '5', '--warmup', '2',
           '--no-prof']
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    assert j['n_gpus'] == 2 and j['value'] > 0 and len(j['extra']['comm']['buckets']) == 4


def test_bench_two_rank_dry_run():
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one rank per GPU), here with both ranks on the
    one GPU of the box and gloo collectives (DSL_BENCH_ONE_GPU / DSL_DIST_BACKEND test hooks): the multi-rank path of the
    benchmark - rank-local batches, bucketed all-reduce behind the named events, max-over-ranks timing, ONE JSON line
    from rank 0 with whole-job throughput - runs end to end before an 8-GPU node ever sees it."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DSL_BENCH_ONE_GPU='1', DSL_DIST_BACKEND='gloo', MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
           '--no-prof']
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['steps'] == 3 and j['warmup'] == 1 and j['scaling'] == 'weak'
    assert j['config']['global_batch'] == 4 and j['config']['parallelism'] == 'dp2'
    assert j['value'] == pytest.approx(4 * 3 / (j['ms_per_step'] * 3e-3), rel=1e-3) and j['value'] > 0
    assert j['cpu_baseline'] is None and np.isfinite(j['final_losses']['loss'])
    comm = j['extra']['comm']          # per gradient bucket: [MB, all-reduce start, done] in ms from the start of the step
    assert len(comm['buckets']) == 4 and all(b['done_ms'] >= b['start_ms'] >= 0 for b in comm['buckets'])


def _worker_opts(rank, world, port, q):
    """Two ranks, bf16 gradient buckets + a clipping optimizer: the second step takes the norm from the per-bucket partial sums."""
    try:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        torch.cuda.set_device(0)
        dist.init_process_group('gloo', rank=rank, world_size=world)
        from dsl_amd import _lib as L
        from dsl_amd.optim import FlatSGD
        from dsl_amd.parallel import HipDistributedDataParallel
        model = build()
        ddp = HipDistributedDataParallel(model, grad_dtype='bf16', wgrad_slots=112)
        slots = C.c_int(0)
        L.check(L.lib.dsl_get_option(b'wgrad_slots', C.byref(slots)))
        assert model.grad_bf16 and slots.value == 112
        opt = FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=1e-4, paramwise_cfg=dict(bias_lr_mult=2., bias_decay_mult=0.),
                      grad_clip=dict(max_norm=1.0, norm_type=2))
        norms = []
        for it in range(2):
            out = ddp.train_step(make_batch(rank), opt)
            out['loss'].backward()
            used_partials = bool(model._partials_valid)
            opt.step()
            torch.cuda.synchronize()
            ref = torch.zeros(1, device='cuda')
            ws = torch.zeros(1024, device='cuda')
            L.check(L.lib.dsl_sumsq_det(L.ptr(model.store.grad), model.store.n_train, L.ptr(ref), L.ptr(ws), L.stream_ptr()))
            torch.cuda.synchronize()
            norms.append((used_partials, float(opt.gnorm_sq), float(ref)))
        w = model.store.train.detach().cpu()
        g = model.store.grad.detach().cpu()
        lst = [torch.zeros_like(w) for _ in range(world)]
        dist.all_gather(lst, w)
        same = all(torch.equal(t, lst[0]) for t in lst)
        bf16_valued = bool(torch.equal(g, g.bfloat16().float()))       # the reduced gradient came back through bf16
        dist.destroy_process_group()
        q.put((rank, 'ok', same, norms, bf16_valued))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc(), False, None, None))


def test_ddp_bf16_buckets_and_clip_norm_in_pieces():
    """HipDistributedDataParallel(grad_dtype='bf16') + FlatSGD(grad_clip=...) on two ranks (gloo, one GPU): the gradient buckets cross
    the wire as bf16 copies (what comes back is bf16-valued, identical on both ranks), the first step's clipping norm is the
    whole-buffer sum, the second step's is folded from the per-bucket partial sums taken on the communication stream - both equal
    the fixed-order sum over the reduced gradient buffer to fp32 summation-order noise - and the ranks end with identical weights."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_opts, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=200) for _ in procs]
    finally:
        for p in procs:
            p.join(20)
            if p.is_alive():
                p.kill()
    assert all(r[1] == 'ok' for r in res), [r[1] for r in res]
    assert all(r[2] for r in res), 'ranks hold different weights'
    assert all(r[4] for r in res), 'the reduced gradient is not bf16-valued: the bucket did not go through the bf16 copy'
    for r in res:
        (p0, n0, ref0), (p1, n1, ref1) = r[3]
        assert not p0 and p1, (p0, p1)                  # step 1: whole-buffer norm; step 2: per-bucket partial sums
        assert n0 == pytest.approx(ref0, rel=1e-5) and n1 == pytest.approx(ref1, rel=1e-5), r[3]
    assert res[0][3] == res[1][3]
