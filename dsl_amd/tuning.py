"""The supported tuning knobs of the package, in ONE place.

Every knob has a measured default (DESIGN.md section 3 names the measurement); `DSL_TUNE="key=value,key=value"` overrides them
for A/B runs (tools/exp_env.sh) - it is the only environment variable the compute path reads (deployment settings: DSL_HIP_LIB,
DSL_PRETRAINED_DIR, DSL_COMM, DSL_GRAD_DTYPE).  Keys starting with `lib.` are handed to the C library's own option table
(dsl_set_option).  An unknown key is an error: rejected experiments do not linger as silent switches.
"""
import os

DEFAULTS = {
    # ---- schedule of the step's op lists (engine.py)
    'side': '1',               # library side streams (weight gradients, second tower, image-split chains); 0 = everything on the caller's stream
    'pipe_prefix': '1',        # the next step's frozen prefix (stem + layer1) on its own stream beside the backward tail
    'img_split': '234',        # forward stages (layer numbers) that run as two half-batch chains
    'bneck_fwd': '2',          # forward stages whose bottlenecks run as one launch each (dsl_bottleneck_fwd; not image-split then): layer2 + 1.5 %;
                               # '23' adds layer3, where the fused kernel only ties the two half-batch chains (- 0.7 %, profiles/r05_bneck_ab_v4.txt)
    'img_split_bwd': '23',     # backward stages whose data-gradient chains do
    'tower_slots': '72',       # workgroup budget of the towers' x8 weight-gradient group
    'tail_slots': '192',       # ... of the last segment's weight gradients (layer2; the RLA backbone's stage 1)
    # ---- RLA_ResNet engine
    'rla_split': '123',        # forward stages of the RLA backbone that run as two chains
    'rla_tail': '1',           # 1: a block's recurrent path (conv_out -> BN + tanh -> 3x3) as one launch (dsl_rla_tail_fwd); 0: three
    'rla_split_bwd': '',       # backward stages whose data-gradient chains do (measured slower on the saturated N = 3 iteration: 12.0 -> 12.15 ms)
    # ---- checks / measurement
    'check_backward_grad': '0',    # 1: verify the gradient handed to loss.backward() on every step (default: the first steps only)
    'skip': '',                # timing-only ablation: '+'-separated items - region tags (fwd.l2 ... bwd.l2), 'sgd', 'prefix'
    # ---- C library options (dsl_set_option)
    'lib.wgrad_slots': '128',
    'lib.stream_probe': '1',   # 0: the library takes its streams as the runtime deals them (no hardware-queue probe)
    'lib.comm_queue': '1',     # hardware queue the communication stream is placed on (1 weight gradients, 2 second chain, 3 prefix, 4 caller; 0 = as dealt)
    'lib.rla_tail_form': '0',  # dsl_rla_tail_fwd's tile form: 0 by shape, 1 = 14 x 14 tiles, 2 = 6 x 6 tiles with conv_out's K over four waves
    'lib.debug_sync': '0',
    'lib.skip_kinds': '0',
}

_values = dict(DEFAULTS)
_parsed = False


def _parse():
    global _parsed
    if _parsed:
        return
    _parsed = True
    for item in os.environ.get('DSL_TUNE', '').split(','):
        item = item.strip()
        if not item:
            continue
        k, _, v = item.partition('=')
        if k not in DEFAULTS:
            raise KeyError(f'DSL_TUNE: unknown key {k!r} (supported: {sorted(DEFAULTS)})')
        _values[k] = v


def tune(key):
    """The value of knob `key` as a string."""
    _parse()
    return _values[key]


def tune_int(key):
    return int(tune(key), 0)


def set_tune(key, value):
    """Programmatic override (tests, the data-parallel wrapper); `lib.` keys go to the library at once."""
    _parse()
    if key not in DEFAULTS:
        raise KeyError(key)
    _values[key] = str(value)
    if key.startswith('lib.'):
        from . import _lib as L
        L.check(L.lib.dsl_set_option(key[4:].encode(), int(str(value), 0)), 'dsl_set_option')


def skip_items():
    return frozenset(t for t in tune('skip').split('+') if t)


def push_lib_options(lib):
    """Called by _lib once the library is loaded."""
    _parse()
    for k, v in _values.items():
        if k.startswith('lib.') and v != DEFAULTS[k]:
            if lib.dsl_set_option(k[4:].encode(), int(v, 0)) != 0:
                raise RuntimeError(f'dsl_set_option({k[4:]}) failed')
