"""Data parallelism for the HIP path: one process per GPU, torch.distributed ("nccl" = RCCL over xGMI).

Replaces mmcv's MMDistributedDataParallel / torch DDP as wired at mmdet/apis/train.py:92-102 of the
reference.  The per-step exchanges are:
  * one 2-float all-reduce of (num_pos, sum centerness targets) before the loss is normalised
    (reduce_mean, fcos_head.py:264-274) - issued right after the assignment kernel so that it overlaps
    the whole network forward;
  * the gradient all-reduce, issued bucket by bucket (head+FPN, layer4, layer3, layer2) as soon as
    the kernels of that backward segment are queued, so RCCL traffic overlaps the remaining backward;
    the 1/world averaging is folded into the loss kernel's gradient scale, so the buckets are plain sums;
  * one small all-reduce of the log vars (base.py:201-206 does one per key).
The flat gradient buffer makes every bucket one contiguous range: no flatten/unflatten copies.

Two carriers for the same exchanges: torch.distributed's process group (default; gloo on CPU for the tests, nccl = RCCL
on GPUs), or `comm='rccl'` / DSL_COMM=rccl: an rcclComm_t of the C-ABI (include/dsl_hip.h dsl_comm_*,
dsl_allreduce_bucket) - what a caller that is not PyTorch binds - with torch.distributed used once, to hand rank 0's
unique id to the other ranks.
"""
import ctypes as C
import os

import torch
import torch.distributed as dist
import torch.nn as nn


class StreamWork:
    """What dist.Work.wait() is for a collective queued on `stream`: the caller's current stream waits for it."""

    def __init__(self, stream):
        self.event = torch.cuda.Event()
        self.event.record(stream)

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)


class RcclComm:
    """An RCCL communicator owned through the C-ABI (dsl_comm_init_rank on this process's current device)."""

    def __init__(self, group=None):
        from . import _lib as L
        self.L = L
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        uid = C.create_string_buffer(128)
        if self.rank == 0:
            L.check(L.lib.dsl_comm_unique_id(uid), 'dsl_comm_unique_id')
        box = [bytes(uid.raw)]
        if self.world > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        comm = C.c_void_p()
        L.check(L.lib.dsl_comm_init_rank(C.byref(comm), self.world, C.create_string_buffer(box[0], 128), self.rank),
                'dsl_comm_init_rank')
        self.comm = comm
        assert L.lib.dsl_comm_size(self.comm) == self.world

    def all_reduce(self, t, stream=None):
        """In-place fp32 sum of the contiguous tensor `t` over the ranks, queued on `stream` (default: the current one)."""
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        s = stream if stream is not None else torch.cuda.current_stream()
        self.L.check(self.L.lib.dsl_allreduce_bucket(self.comm, C.c_void_p(t.data_ptr()), t.numel(), C.c_void_p(s.cuda_stream)),
                     'dsl_allreduce_bucket')
        return t

    def close(self):
        if self.comm is not None and self.comm.value:
            self.L.lib.dsl_comm_destroy(self.comm)
        self.comm = None


class HipDistributedDataParallel(nn.Module):
    """comm: 'torch' (default; torch.distributed's process group), 'rccl' (the C-ABI's own communicator) or 'none' (every
    collective skipped: a timing probe that attributes a scaling loss to communication - the gradients are then WRONG);
    env DSL_COMM.  grad_dtype: 'fp32' (default) or 'bf16' - gradient buckets cross xGMI as bf16 copies (64 MB instead of 128 MB per
    step; the master gradient, the clipping norm and the update stay fp32); env DSL_GRAD_DTYPE.
    wgrad_slots: workgroup budget of the persistent weight-gradient grids (library option "wgrad_slots", default 128).  With more than
    one rank RCCL's own kernels need CUs beside a backward pass that otherwise fills the chip; a lower cap (e.g. 112) is the knob
    prepared for that - opt-in and unmeasured until a multi-GPU node runs bench.py: it changes the weight-gradient
    planner's split factors, hence summation order, so a run with it is not bit-comparable with a single-GPU run."""

    def __init__(self, module, process_group=None, broadcast_buffers=False, find_unused_parameters=False,
                 device_ids=None, comm=None, grad_dtype=None, wgrad_slots=None, **kw):
        super().__init__()
        assert dist.is_initialized(), 'init_process_group first (tools/train.py:116-123)'
        self.module = module
        self.group = process_group
        module.dist_group = process_group
        module.world_size = dist.get_world_size(process_group)
        comm = comm if comm is not None else os.environ.get('DSL_COMM', 'torch')
        assert comm in ('torch', 'rccl', 'none'), comm
        module.comm_off = comm == 'none'
        grad_dtype = grad_dtype if grad_dtype is not None else os.environ.get('DSL_GRAD_DTYPE', 'fp32')
        assert grad_dtype in ('fp32', 'bf16'), grad_dtype
        module.grad_bf16 = grad_dtype == 'bf16'
        if wgrad_slots is not None:
            from .tuning import set_tune
            set_tune('lib.wgrad_slots', int(wgrad_slots))      # (an explicit library setter: read when a launch is planned)
        module.rccl = RcclComm(process_group) if (comm == 'rccl' and module.store.train.is_cuda) else None
        # initial parameter broadcast from rank 0 (what DDP's constructor does)
        st = module.store
        for buf in (st.train, st.frozen):
            dist.broadcast(buf, src=0, group=process_group)
        st.dirty = True

    def forward(self, *a, **k):
        return self.module(*a, **k)

    def train_step(self, data, optimizer):
        return self.module.train_step(data, optimizer)

    def val_step(self, data, optimizer=None):
        return self.module.val_step(data, optimizer)

    def state_dict(self, *a, **k):
        return self.module.state_dict(*a, **k)

    def load_state_dict(self, sd, strict=True):
        return self.module.load_state_dict(sd, strict)
