"""ORACLE (test infrastructure only - never imported by the product path).

CPU restatement, in plain PyTorch fp32, of the reference actor-critic network
(/root/reference/policy.py:51-178), parametrised over the recurrent cell as SURVEY.md section 8(c)
prescribes for the configurations that have no reference implementation:

    cell='gru',  hidden=256, layers=1  -> the reference network itself (policy.py:66)
    cell='lstm', hidden=H,   layers=n  -> nn.GRU swapped for nn.LSTM(256, H, n), head in-features H

Parity status: PINNED for gru/256/1 - tests/test_oracle.py loads the same weights into this class and
checks it against tests/golden/*.npz, which were produced by importing the real reference
(tests/golden/make_golden.py).  The lstm / other-H variants have no reference to pin against
("parity unpinned" for those): they are only ever compared with this restatement.

Reference quirks kept on purpose:
  * the pooled enemy-tower slot is the max over the *enemy non-hero* embedding (policy.py:127);
  * masked_softmax has no max-subtraction and yields +inf on all-False rows (policy.py:169-178).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

# (observation key, units, parameter suffix) in concat order, policy.py:100-131
_UNIT_TYPES = [('allied_heroes', 1, 'ah'), ('enemy_heroes', 5, 'eh'), ('allied_nonheroes', 16, 'anh'),
               ('enemy_nonheroes', 16, 'enh'), ('allied_towers', 1, 'ath'), ('enemy_towers', 1, 'eth')]
HEAD_WIDTHS = {'enum': 4, 'x': 9, 'y': 9, 'target_unit': 40, 'ability': 3}      # policy.py:46


class RefPolicy(nn.Module):
    def __init__(self, cell='gru', hidden=256, layers=1):
        super().__init__()
        self.cell, self.hidden_size, self.layers = cell, hidden, layers
        self.affine_env = nn.Linear(3, 128)                           # policy.py:54
        self.affine_unit_basic_stats = nn.Linear(12, 128)             # policy.py:56 (shared)
        for _, _, suf in _UNIT_TYPES:                                 # policy.py:58-63
            setattr(self, 'affine_unit_' + suf, nn.Linear(128, 128))
        self.affine_pre_rnn = nn.Linear(896, 256)                     # policy.py:65
        rnn_cls = {'gru': nn.GRU, 'lstm': nn.LSTM}[cell]
        self.rnn = rnn_cls(input_size=256, hidden_size=hidden, num_layers=layers, batch_first=True)  # :66
        self.affine_head_enum = nn.Linear(hidden, 4)                  # policy.py:69-75
        self.affine_move_x = nn.Linear(hidden, 9)
        self.affine_move_y = nn.Linear(hidden, 9)
        self.affine_unit_attention = nn.Linear(hidden, 128)
        self.affine_head_ability = nn.Linear(hidden, 3)
        self.affine_value = nn.Linear(hidden, 1)

    def init_hidden(self, batch=1):                                   # policy.py:77-78
        z = torch.zeros(self.layers, batch, self.hidden_size)
        return z if self.cell == 'gru' else (z, z.clone())

    def forward(self, obs, hidden):
        """obs: dict key -> (B,S,...) float32; hidden as init_hidden.  Returns (logits dict, value, hidden)."""
        pooled = {'env': F.relu(self.affine_env(obs['env']))}          # policy.py:97
        per_unit = []
        for key, _, suf in _UNIT_TYPES:                               # policy.py:100-127
            basic = F.relu(self.affine_unit_basic_stats(obs[key]))
            emb = getattr(self, 'affine_unit_' + suf)(basic)          # (B,S,units,128)
            per_unit.append(emb)
            pooled[suf] = emb.max(dim=2).values
        pooled['eth'] = pooled['enh']                                 # policy.py:127 (reference bug, kept)
        units = torch.cat(per_unit, dim=2)                            # (B,S,40,128)  policy.py:130
        x = torch.cat([pooled[k] for k in ('env', 'ah', 'eh', 'anh', 'enh', 'ath', 'eth')], dim=2)  # :135
        x = F.relu(self.affine_pre_rnn(x))                            # policy.py:138
        x, hidden = self.rnn(x, hidden)                               # policy.py:141
        query = self.affine_unit_attention(x)                         # policy.py:144
        logits = {
            'enum': self.affine_head_enum(x),
            'x': self.affine_move_x(x),
            'y': self.affine_move_y(x),
            'target_unit': torch.einsum('bsk,bsuk->bsu', query, units),   # policy.py:152
            'ability': self.affine_head_ability(x),
        }
        return logits, self.affine_value(x), hidden                   # policy.py:155-167


def masked_log_softmax(logits, mask):
    """policy.py:169-178: logits - log(sum over mask of exp(logits)), no max-subtraction."""
    e = torch.exp(logits)
    e = torch.where(mask, e, torch.zeros_like(e))
    return logits - torch.log(e.sum(dim=-1, keepdim=True))
