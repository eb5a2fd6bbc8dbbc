"""Data layout shared by the host shim, the HIP library and the tests.

Everything here is a restatement of shapes fixed by the reference:
  * observation keys / unit counts      : /root/reference/policy.py:45-49
  * action heads and their widths       : /root/reference/policy.py:46
  * parameter names / shapes (wire fmt) : /root/reference/policy.py:54-75
  * sub-reward count                    : /root/reference/policy.py:20

Flattened per-env-step record (what the kernels read, one row per step):
  obs  : float32[483] = env(3) | 40 units x 12 features, unit order
         allied_heroes(1) enemy_heroes(5) allied_nonheroes(16) enemy_nonheroes(16)
         allied_towers(1) enemy_towers(1)        (concat order of policy.py:130-131)
  act  : uint8[65]  one-hot-or-zero, heads in order enum(4) x(9) y(9) target_unit(40) ability(3)
  mask : uint8[65]  same layout
  rew  : float32[10] sub-rewards (REWARD_KEYS order)
"""
from collections import OrderedDict

INPUT_KEYS = ['env', 'allied_heroes', 'enemy_heroes', 'allied_nonheroes', 'enemy_nonheroes',
              'allied_towers', 'enemy_towers']
UNIT_KEYS = INPUT_KEYS[1:]
UNIT_COUNTS = OrderedDict([('allied_heroes', 1), ('enemy_heroes', 5), ('allied_nonheroes', 16),
                           ('enemy_nonheroes', 16), ('allied_towers', 1), ('enemy_towers', 1)])
# parameter-name suffix per unit type (policy.py:58-63)
UNIT_SUFFIX = OrderedDict([('allied_heroes', 'ah'), ('enemy_heroes', 'eh'), ('allied_nonheroes', 'anh'),
                           ('enemy_nonheroes', 'enh'), ('allied_towers', 'ath'), ('enemy_towers', 'eth')])
N_TYPES = 6
MAX_UNITS = sum(UNIT_COUNTS.values())          # 40
UNIT_FEATS = 12
ENV_FEATS = 3
OBS_DIM = ENV_FEATS + MAX_UNITS * UNIT_FEATS   # 483
EMB = 128                                      # embedding width (policy.py:54-63)
XCAT = 7 * EMB                                 # 896 (policy.py:135-136)
PRE = 256                                      # affine_pre_rnn out / rnn input (policy.py:65-66)

HEAD_COUNTS = OrderedDict([('enum', 4), ('x', 9), ('y', 9), ('target_unit', MAX_UNITS), ('ability', 3)])
OUTPUT_KEYS = list(HEAD_COUNTS.keys())
N_HEADS = 5
ACT_DIM = sum(HEAD_COUNTS.values())            # 65
HEAD_OFFSETS = OrderedDict()
_o = 0
for _k, _c in HEAD_COUNTS.items():
    HEAD_OFFSETS[_k] = _o
    _o += _c
N_REWARDS = 10

# Column layout of the fused head projection ("headout", leading dimension HEADOUT_LD):
#   [0,128)   affine_unit_attention (query)
#   [128,132) enum  [132,141) x  [141,150) y  [150,153) ability  [153] value
HEADOUT_Q = 0
HEADOUT_ENUM = 128
HEADOUT_X = 132
HEADOUT_Y = 141
HEADOUT_ABILITY = 150
HEADOUT_VALUE = 153
HEADOUT_N = 154
HEADOUT_LD = 160

GATES = {'gru': 3, 'lstm': 4}


def param_shapes(cell='gru', hidden=256, layers=1):
    """name -> shape, in the reference's named_parameters() order (policy.py:54-75).

    For cell='gru', hidden=256, layers=1 this is exactly the reference state_dict (34 tensors,
    765 210 floats).  Other cells/sizes are the parametrised restatement of SURVEY.md section 8(c).
    """
    g = GATES[cell]
    h = hidden
    d = OrderedDict()
    d['affine_env.weight'] = (EMB, ENV_FEATS)
    d['affine_env.bias'] = (EMB,)
    d['affine_unit_basic_stats.weight'] = (EMB, UNIT_FEATS)
    d['affine_unit_basic_stats.bias'] = (EMB,)
    for suf in UNIT_SUFFIX.values():
        d['affine_unit_%s.weight' % suf] = (EMB, EMB)
        d['affine_unit_%s.bias' % suf] = (EMB,)
    d['affine_pre_rnn.weight'] = (PRE, XCAT)
    d['affine_pre_rnn.bias'] = (PRE,)
    for l in range(layers):
        inp = PRE if l == 0 else h
        d['rnn.weight_ih_l%d' % l] = (g * h, inp)
        d['rnn.weight_hh_l%d' % l] = (g * h, h)
        d['rnn.bias_ih_l%d' % l] = (g * h,)
        d['rnn.bias_hh_l%d' % l] = (g * h,)
    d['affine_head_enum.weight'] = (4, h)
    d['affine_head_enum.bias'] = (4,)
    d['affine_move_x.weight'] = (9, h)
    d['affine_move_x.bias'] = (9,)
    d['affine_move_y.weight'] = (9, h)
    d['affine_move_y.bias'] = (9,)
    d['affine_unit_attention.weight'] = (EMB, h)
    d['affine_unit_attention.bias'] = (EMB,)
    d['affine_head_ability.weight'] = (3, h)
    d['affine_head_ability.bias'] = (3,)
    d['affine_value.weight'] = (1, h)
    d['affine_value.bias'] = (1,)
    return d


def flat_order(cell='gru', hidden=256, layers=1):
    """Order of the tensors inside the flat fp32 parameter buffer used by the HIP library.

    The order differs from named_parameters(): the six head projections are stored back to back
    so that they form one [154, H] weight and one [154] bias (HEADOUT_* column layout above) and
    the six unit-type weights form one [6,128,128] block.  Only names/shapes are wire format
    (optimizer.py:706-716, agent.py:186,315), the in-memory order is ours.
    """
    names = ['affine_env.weight', 'affine_env.bias',
             'affine_unit_basic_stats.weight', 'affine_unit_basic_stats.bias']
    names += ['affine_unit_%s.weight' % s for s in UNIT_SUFFIX.values()]
    names += ['affine_unit_%s.bias' % s for s in UNIT_SUFFIX.values()]
    names += ['affine_pre_rnn.weight', 'affine_pre_rnn.bias']
    for l in range(layers):
        names += ['rnn.weight_ih_l%d' % l, 'rnn.weight_hh_l%d' % l,
                  'rnn.bias_ih_l%d' % l, 'rnn.bias_hh_l%d' % l]
    hw = ['affine_unit_attention', 'affine_head_enum', 'affine_move_x', 'affine_move_y',
          'affine_head_ability', 'affine_value']
    names += [n + '.weight' for n in hw]
    names += [n + '.bias' for n in hw]
    return names


def flat_layout(cell='gru', hidden=256, layers=1):
    """name -> (offset, numel, shape) in the flat buffer; offsets in floats.

    Every tensor starts on a 4-float (16 B) boundary so kernels may use 16-byte loads on weight
    rows; the pad floats are zero, receive zero gradients and are never exposed.
    Returns (OrderedDict, total_floats).
    """
    shapes = param_shapes(cell, hidden, layers)
    out = OrderedDict()
    off = 0
    # members of the two dense head blocks (weights [154,H], biases [154]) that must directly
    # follow their predecessor without alignment padding
    followers = set()
    for n in ['affine_head_enum', 'affine_move_x', 'affine_move_y', 'affine_head_ability', 'affine_value']:
        followers.add(n + '.weight')
        followers.add(n + '.bias')
    for n in flat_order(cell, hidden, layers):
        shp = shapes[n]
        numel = 1
        for s in shp:
            numel *= s
        if n not in followers:
            off = (off + 3) // 4 * 4
        out[n] = (off, numel, shp)
        off += numel
    total = (off + 3) // 4 * 4
    return out, total
