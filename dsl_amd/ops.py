"""Descriptor builders for the libdsl_hip.so entry points, over torch CUDA tensors used purely as
device memory.  Every function returns the ctypes descriptor (keep it alive while it is referenced
from an op list) and, unless `launch=False`, enqueues the kernel on the current stream."""
import ctypes as C

import torch

from . import _lib as L

lib = L.lib


def _segs(shapes):
    return L.seg5([s[0] for s in shapes]), L.seg5([s[1] for s in shapes])


def conv_desc(src, wgt, dst, *, n, grid, src_hw, dst_hw, cs, cd, cd_pad, ldd, kh, kw, stride=1, pad=0,
              mode=0, os=1, flags=0, scale=None, bias=None, addend=None, lda=0, add_hw=None, mask=None,
              ldm=0, workspace=None, cs_real=0, lds=0):
    """grid/src_hw/dst_hw/add_hw: list of (h, w) per level segment."""
    d = L.ConvDesc()
    d.nseg, d.n = len(grid), n
    d.gh, d.gw = _segs(grid)
    d.sh, d.sw = _segs(src_hw)
    d.dh, d.dw = _segs(dst_hw)
    if add_hw is not None:
        d.ah, d.aw = _segs(add_hw)
    d.cs, d.cd, d.cd_pad, d.ldd, d.lda, d.ldm = cs, cd, cd_pad, ldd, lda, ldm
    d.kh, d.kw, d.stride, d.pad, d.mode, d.os, d.flags = kh, kw, stride, pad, mode, os, flags
    d.cs_real, d.lds = cs_real, lds
    d.src, d.wgt, d.dst = L.ptr(src), L.ptr(wgt), L.ptr(dst)
    d.scale, d.bias, d.addend, d.mask = L.ptr(scale), L.ptr(bias), L.ptr(addend), L.ptr(mask)
    if workspace is not None:
        d.workspace, d.workspace_bytes = L.ptr(workspace), workspace.numel() * workspace.element_size()
    d._keep = (src, wgt, dst, scale, bias, addend, mask, workspace)
    return d


# ---- data gradient of a 3x3 / stride-2 / pad-1 convolution as four stride-1 convolutions ------------------------------------------
# dX[y][x] = sum over taps (r, s) with (y + 1 - r), (x + 1 - s) even of dY[(y + 1 - r) / 2][(x + 1 - s) / 2] . W[r][s]: an output
# pixel of parity class (py, px) = (y % 2, x % 2) only ever meets the taps r = py +- 1, s = px +- 1 - one tap for (0, 0), two for
# (0, 1) / (1, 0), four for (1, 1).  Each class is a stride-1 transposed convolution over the dY grid with a 1x1 (class (0, 0)) or
# 2x2 kernel (weight tap t -> source row i + pad - t, the library's mode-1 rule), written with output stride 2 at offset (py, px).
# 13 of the 36 tap-pixel products of the strided gather per 2x2 output block remain (the 2x2 classes (0, 1), (1, 0) carry two zero
# taps: the descriptor has one `pad` for both axes), and the launches go to the pipelined kernel instead of the general-gather one
# (conv_glds: 292 / 167 / 75 us for RLA_ResNet's three stride-2 convolutions at N = 3, profiles/r03_rla_sequence.txt).
def s2_class(py, px):
    """(k, pad, source taps of the class's k*k weight taps in (t_r, t_s) order; -1 = a zero tap)."""
    if py == 0 and px == 0:
        return 1, 0, [4]
    taps = []
    for tr in (0, 1):
        r = (py - 1) if tr == 0 else (py + 1)
        for ts in (0, 1):
            s_ = (px - 1) if ts == 0 else (px + 1)
            taps.append(r * 3 + s_ if (0 <= r <= 2 and 0 <= s_ <= 2) else -1)
    return 2, 1, taps


def dgrad_s2_descs(dy, packs, dst, *, n, dy_hw, dst_hw, cs, cd, cd_pad=None, ldd=None, mask=None, ldm=0, addend=None, lda=0, flags=0,
                   workspace=None, cs_real=0):
    """Four conv descriptors (one per parity class) for the data gradient of a 3x3 / 2 / pad 1 convolution.  dy: [n][oh][ow][cs] bf16
    (tensor or pointer), packs: {(py, px): pointer of the class's [cd_in = Cin][k][k][cs] bf16 pack}, dst / mask / addend: tensors or
    raw pointers of [n][h][w][.] rows.  One segment."""
    (oh, ow), (h, w) = dy_hw, dst_hw
    ldd = ldd or cd
    raw = lambda t: 0 if t is None else (t if isinstance(t, int) else t.data_ptr())
    out = []
    for py in (0, 1):
        for px in (0, 1):
            gh, gw = (h - py + 1) // 2, (w - px + 1) // 2
            if gh <= 0 or gw <= 0:
                continue
            k, pad, _ = s2_class(py, px)
            off = py * w + px
            d = conv_desc(dy, packs[(py, px)], raw(dst) + off * ldd * 2, n=n, grid=[(gh, gw)], src_hw=[(oh, ow)], dst_hw=[(h, w)], cs=cs, cd=cd,
                          cd_pad=cd_pad or cd, ldd=ldd, kh=k, kw=k, stride=1, pad=pad, mode=1, os=2, flags=flags,
                          addend=(raw(addend) + off * (lda or cd) * 2) if addend is not None else None, lda=lda or cd,
                          mask=(raw(mask) + off * (ldm or cd) * 2) if mask is not None else None, ldm=ldm or cd, workspace=workspace,
                          cs_real=cs_real)
            d._keep = d._keep + (dy, dst, mask, addend)
            out.append(d)
    return out


def bneck_desc(x, w1, w2, w3, idt, bn1, bn2, bn3, a1, a2, out, *, n, hin, win, h, w, planes, cin, ldx=None, stride=1, ldi=None, ldo=None):
    """dsl_bottleneck_fwd: a trained bottleneck's forward pass as one launch (csrc/bneck.hip).  Tensors or raw pointers; bn_k = (scale, bias)
    pointers of the folded BatchNorms."""
    d = L.BneckDesc()
    raw = lambda t: 0 if t is None else (t if isinstance(t, int) else t.data_ptr())
    d.x, d.w1, d.w2, d.w3, d.idt = raw(x), raw(w1), raw(w2), raw(w3), raw(idt)
    (d.s1, d.b1), (d.s2, d.b2), (d.s3, d.b3) = [(raw(s_), raw(b_)) for s_, b_ in (bn1, bn2, bn3)]
    d.a1, d.a2, d.out = raw(a1), raw(a2), raw(out)
    d.n, d.hin, d.win, d.h, d.w = n, hin, win, h, w
    d.planes, d.cin, d.ldx, d.stride = planes, cin, ldx or cin, stride
    d.ldi, d.ldo = ldi or 4 * planes, ldo or 4 * planes
    d._keep = (x, w1, w2, w3, idt, bn1, bn2, bn3, a1, a2, out)
    return d


def bottleneck_fwd(*a, **k):
    d = bneck_desc(*a, **k)
    L.check(lib.dsl_bottleneck_fwd(C.byref(d), L.stream_ptr()), 'dsl_bottleneck_fwd')
    return d


def conv_workspace_bytes(d):
    return lib.dsl_conv2d_workspace_bytes(C.byref(d))


def conv2d(*a, **k):
    d = conv_desc(*a, **k)
    L.check(lib.dsl_conv2d(C.byref(d), L.stream_ptr()), 'dsl_conv2d')
    return d


def wgrad_desc(dy, x, dw, *, n, grid, src_hw, cs, cy, cd, kh, kw, stride=1, pad=0, scale=None, db=None,
               workspace=None, force_cfg=None, ldx=0, shared=0, slots=0):
    d = L.WgradDesc()
    d.nseg, d.n = len(grid), n
    d.gh, d.gw = _segs(grid)
    d.sh, d.sw = _segs(src_hw)
    d.cs, d.cy, d.cd, d.kh, d.kw, d.stride, d.pad = cs, cy, cd, kh, kw, stride, pad
    d.ldx, d.shared, d.slots = ldx, shared, slots
    d.splits = 0 if force_cfg is None else -(force_cfg + 1)     # negative: test hook forcing a tile config
    if force_cfg is None:
        d.splits = lib.dsl_wgrad_splits(C.byref(d))
    d.dy, d.x, d.scale, d.dw, d.db = L.ptr(dy), L.ptr(x), L.ptr(scale), L.ptr(dw), L.ptr(db)
    need = lib.dsl_wgrad_workspace_bytes(C.byref(d))
    if workspace is None:
        workspace = torch.empty(need, dtype=torch.uint8, device=dy.device)
    assert workspace.numel() * workspace.element_size() >= need, (workspace.numel(), need)
    d.workspace, d.workspace_bytes = L.ptr(workspace), workspace.numel() * workspace.element_size()
    tab = pixtab(d, dy.device if hasattr(dy, 'device') else 'cuda')
    d._keep = (dy, x, dw, scale, db, workspace, tab)
    return d


_pixtabs = None


def pixtab(d, device='cuda'):
    """The pixel descriptor table of d's geometry (dsl_wgrad_desc.pixtab: caller-owned since round 6, the library allocates nothing):
    one tensor per geometry and device, shared by every descriptor that has it, filled by dsl_wgrad_pixtab_fill and freed with the last
    descriptor that holds it (weak cache).  Sets d.pixtab / d.pixtab_bytes; returns the tensor (None: this geometry needs no table)."""
    global _pixtabs
    import weakref
    if _pixtabs is None:
        _pixtabs = weakref.WeakValueDictionary()
    need = lib.dsl_wgrad_pixtab_bytes(C.byref(d))
    if need == 0:
        d.pixtab, d.pixtab_bytes = None, 0
        return None
    dev = torch.device(device)
    if dev.index is None:
        dev = torch.device('cuda', torch.cuda.current_device())
    key = (dev.index, d.nseg, d.n, d.kh, d.kw, d.stride, d.pad) + tuple(d.gh) + tuple(d.gw) + tuple(d.sh) + tuple(d.sw)
    t = _pixtabs.get(key)
    if t is None:
        t = torch.empty(need, dtype=torch.uint8, device=dev)
        L.check(lib.dsl_wgrad_pixtab_fill(C.byref(d), L.ptr(t), need, L.stream_ptr()), 'dsl_wgrad_pixtab_fill')
        torch.cuda.current_stream().synchronize()       # once per geometry, at plan-build time: its readers run on other streams
        _pixtabs[key] = t
    d.pixtab, d.pixtab_bytes = L.ptr(t), t.numel()
    return t


def wgrad_workspace_bytes(*, n, grid, src_hw, cs, cy, cd, kh, kw, stride=1, pad=0):
    d = L.WgradDesc()
    d.nseg, d.n = len(grid), n
    d.gh, d.gw = _segs(grid)
    d.sh, d.sw = _segs(src_hw)
    d.cs, d.cy, d.cd, d.kh, d.kw, d.stride, d.pad = cs, cy, cd, kh, kw, stride, pad
    d.splits = 0
    return lib.dsl_wgrad_workspace_bytes(C.byref(d))


def wgrad_group(descs, workspace=None):
    """Packs same-geometry wgrad descriptors into one contiguous array for dsl_conv2d_wgrad_group; element 0 carries
    the group's workspace."""
    assert 1 <= len(descs) <= L.MAX_GROUP
    arr = (L.WgradDesc * len(descs))()
    for i, d in enumerate(descs):
        C.memmove(C.addressof(arr[i]), C.addressof(d), C.sizeof(L.WgradDesc))
        arr[i].splits = 0
    need = lib.dsl_wgrad_group_workspace_bytes(arr, len(descs))
    if workspace is None:
        workspace = torch.empty(need, dtype=torch.uint8, device='cuda')
    nbytes = workspace.numel() * workspace.element_size()
    assert nbytes >= need, (nbytes, need)
    arr[0].workspace, arr[0].workspace_bytes = L.ptr(workspace), nbytes
    arr._keep = (list(descs), workspace)
    return arr


class WgradMulti:
    """Launch plan of dsl_conv2d_wgrad_multi: `subs` = list of same-geometry descriptor lists, all of one tile
    configuration (lib.dsl_wgrad_multi_config).  Owns the host table, its device copy and references to the descriptors."""

    def __init__(self, subs, workspace=None, device='cuda'):
        assert 1 <= len(subs) <= L.MAX_MULTI and all(1 <= len(g) <= L.MAX_GROUP for g in subs)
        flat = [d for g in subs for d in g]
        self.descs = (L.WgradDesc * len(flat))()
        for i, d in enumerate(flat):
            C.memmove(C.addressof(self.descs[i]), C.addressof(d), C.sizeof(L.WgradDesc))
            self.descs[i].splits = 0
        self.counts = (C.c_int * len(subs))(*[len(g) for g in subs])
        self.nsub = len(subs)
        need = lib.dsl_wgrad_multi_workspace_bytes(self.descs, self.counts, self.nsub)
        if need == 0:
            L.check(-1, 'dsl_wgrad_multi_workspace_bytes')
        self.workspace_bytes = need
        if callable(workspace):                     # workspace(need) -> tensor: the caller's shared scratch buffer
            workspace = workspace(need)
        if workspace is None:
            workspace = torch.empty(need, dtype=torch.uint8, device=device)
        assert workspace.numel() * workspace.element_size() >= need
        nb = lib.dsl_wgrad_multi_table_bytes()
        self.host = torch.zeros(nb, dtype=torch.uint8)
        L.check(lib.dsl_wgrad_multi_build(self.descs, self.counts, self.nsub, L.ptr(workspace), workspace.numel() * workspace.element_size(),
                                          C.c_void_p(self.host.data_ptr()), nb), 'dsl_wgrad_multi_build')
        self.dev = self.host.to(device)
        self._keep = (flat, workspace)

    def run(self, stream=None):
        L.check(lib.dsl_conv2d_wgrad_multi(C.c_void_p(self.host.data_ptr()), L.ptr(self.dev), stream or L.stream_ptr()), 'dsl_conv2d_wgrad_multi')


def conv2d_wgrad_group(descs, workspace=None):
    arr = wgrad_group(descs, workspace)
    L.check(lib.dsl_conv2d_wgrad_group(arr, len(descs), L.stream_ptr()), 'dsl_conv2d_wgrad_group')
    return arr


def conv2d_wgrad(*a, **k):
    d = wgrad_desc(*a, **k)
    L.check(lib.dsl_conv2d_wgrad(C.byref(d), L.stream_ptr()), 'dsl_conv2d_wgrad')
    return d


def gn_desc(x, y, gamma, beta, stats, workspace=None, *, n, hw, c=256, groups=32, eps=1e-5, dy=None, dx=None,
            dgamma=None, dbeta=None, dbias=None, y8=None, y8_scale=None, y8_amax=None):
    """workspace: uint8/float tensor of >= dsl_groupnorm_workspace_bytes (allocated here when None); calls that may
    run concurrently (different streams) need different workspaces."""
    d = L.GnDesc()
    d.nseg, d.n, d.c, d.groups = len(hw), n, c, groups
    d.h, d.w = _segs(hw)
    d.eps = eps
    d.x, d.y, d.gamma, d.beta, d.stats = (L.ptr(t) for t in (x, y, gamma, beta, stats))
    d.dy, d.dx, d.dgamma, d.dbeta, d.dbias = (L.ptr(t) for t in (dy, dx, dgamma, dbeta, dbias))
    need = lib.dsl_groupnorm_workspace_bytes(C.byref(d))
    if workspace is None:
        workspace = torch.empty(need, dtype=torch.uint8, device='cuda')
    nbytes = workspace.numel() * workspace.element_size()
    assert nbytes >= need, (nbytes, need)
    d.workspace, d.workspace_bytes = L.ptr(workspace), nbytes
    d.y8, d.y8_scale, d.y8_amax = (L.ptr(t) for t in (y8, y8_scale, y8_amax))
    d._keep = (x, y, gamma, beta, stats, workspace, dy, dx, dgamma, dbeta, dbias, y8, y8_scale, y8_amax)
    return d


def fcos_desc(*, n, sizes, strides, ranges, radius=1.5, num_classes=80):
    d = L.FcosDesc()
    d.nlvl, d.n = len(sizes), n
    d.h, d.w = _segs(sizes)
    d.stride = L.seg5(strides)
    for i, (lo, hi) in enumerate(ranges):
        d.range_lo[i], d.range_hi[i] = float(lo), float(hi)
    d.radius, d.num_classes = radius, num_classes
    d.loss_weight, d.soft_weight, d.grad_scale, d.inv_world = 1.0, 0.0, 1.0, 1.0
    d._keep = {}
    return d


def set_ptrs(d, **tensors):
    for k, t in tensors.items():
        setattr(d, k, L.ptr(t))
        d._keep[k] = t
    return d
