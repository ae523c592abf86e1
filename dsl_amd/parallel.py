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
"""
import torch
import torch.distributed as dist
import torch.nn as nn


class HipDistributedDataParallel(nn.Module):
    def __init__(self, module, process_group=None, broadcast_buffers=False, find_unused_parameters=False,
                 device_ids=None, **kw):
        super().__init__()
        assert dist.is_initialized(), 'init_process_group first (tools/train.py:116-123)'
        self.module = module
        self.group = process_group
        module.dist_group = process_group
        module.world_size = dist.get_world_size(process_group)
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
