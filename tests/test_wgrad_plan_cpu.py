"""The weight-gradient launch planner (csrc/conv.hip wgrad_plan, round 4) on the geometries of the benchmark step (N = 2, 800 x 1344):
pure host logic, callable without a device through dsl_wgrad_plan_probe.  What the launches compute does not depend on the plan
(tests/test_kernels_gpu.py::test_wgrad_multi runs them on the GPU); here: the plan is never worse than round 3's rule, its schedule
table holds every work item exactly once (the probe checks that itself and fails otherwise), and the three cases the planner was
written for come out as intended."""
import ctypes as C

import pytest

from dsl_amd import _lib as L


def stages(px):
    return (px + 31) // 32


def probe(subs, cfg, cap=128):
    n = len(subs)
    st = (C.c_int * n)(*[s[0] for s in subs])
    ti = (C.c_int * n)(*[s[1] for s in subs])
    te = (C.c_longlong * n)(*[s[2] for s in subs])
    sp = (C.c_int * n)()
    info = (C.c_int * 5)()
    L.check(L.lib.dsl_wgrad_plan_probe(st, ti, te, n, cfg, cap, sp, info), 'dsl_wgrad_plan_probe')
    return list(sp), dict(grid=info[0], makespan=info[1], items=info[2], old_makespan=info[3], old_items=info[4])


PX = dict(p3=33600, p4=8400, p5=2100, p6=546, p7=154, all=44800)
CASES = {
    # (K stages, output tiles over all members, fp32 elements of one split's tile set)
    'predictors': (3, [(stages(PX['all']), 9, 128 * 2304)] * 2),
    'fpn': (1, [(stages(PX[k]), 9, 256 * 2304) for k in ('p7', 'p6', 'p5', 'p4', 'p3')]
            + [(stages(PX['p5']), 8, 256 * 2048), (stages(PX['p4']), 4, 256 * 1024), (stages(PX['p3']), 2, 256 * 512)]),
    'layer4': (1, [(stages(PX['p5']), 48, 3 * 2048 * 512), (stages(PX['p5']), 108, 3 * 512 * 4608), (stages(PX['p5']), 32, 2 * 512 * 2048),
                   (stages(PX['p5']), 8, 512 * 1024), (stages(PX['p5']), 32, 2048 * 1024)]),
    'layer3': (1, [(stages(PX['p4']), 24, 6 * 1024 * 256), (stages(PX['p4']), 54, 6 * 256 * 2304), (stages(PX['p4']), 20, 5 * 256 * 1024),
                   (stages(PX['p4']), 2, 256 * 512), (stages(PX['p4']), 8, 1024 * 512)]),
}


@pytest.mark.parametrize('name', sorted(CASES))
def test_plan_not_worse_than_the_stride_walk(name):
    cfg, subs = CASES[name]
    sp, info = probe(subs, cfg)
    assert info['grid'] % 8 == 0 and 8 <= info['grid'] <= 128
    assert info['makespan'] <= info['old_makespan'], (name, sp, info)
    ideal = sum(s[0] * s[1] for s in subs) / 128
    assert info['makespan'] <= 1.35 * ideal + 16, (name, sp, info, ideal)
    for (st, _, _), s in zip(subs, sp):
        assert 1 <= s <= max(1, st // 8)
        tps = -(-st // s)
        assert (s - 1) * tps < st, 'an empty split would leave a partial tile unwritten'


def test_the_three_cases_the_planner_was_written_for():
    sp, info = probe(*reversed(CASES['predictors']))
    assert info['items'] <= 128 and info['makespan'] < 0.65 * info['old_makespan']       # one round instead of two for 16 workgroups
    sp, info = probe(*reversed(CASES['layer3']))
    assert sp == [1] * 5 and info['grid'] < 128                                              # whole tiles: no partials, no reduce pass
    sp, info = probe(*reversed(CASES['fpn']))
    assert info['makespan'] < 0.8 * info['old_makespan']


def test_small_caps_and_single_sub_launch():
    for cap in (8, 64, 72, 256):
        sp, info = probe([(stages(44800), 72, 8 * 256 * 2304)], 1, cap=cap if cap % 8 == 0 else cap + (8 - cap % 8))
        assert info['makespan'] > 0 and info['grid'] <= max(8, cap)
