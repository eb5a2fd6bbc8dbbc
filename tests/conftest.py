import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU: skip (with the reason) instead of failing inside the first HIP call.  On a GPU box
    # nothing is skipped, and the product path itself has no CPU fallback (dotaclient_amd/_lib.py, engine.Engine).
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='needs a real MI355X (torch.cuda.is_available() is False)')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


# Element-wise relative errors of every parity compare() of the session (tests/test_gpu_parity.py): kept as evidence next to the
# scaled 1e-4 figures, written to gpurun_out/elementwise_parity.json when the session ends.
_ELEMENTWISE = []


def record_elementwise(rec):
    test = os.environ.get('PYTEST_CURRENT_TEST', '?').split(' ')[0]
    _ELEMENTWISE.append({'test': test, 'quantities': rec})


def pytest_sessionfinish(session, exitstatus):
    if not _ELEMENTWISE:
        return
    import json
    out = os.path.join(REPO, 'gpurun_out')
    try:
        os.makedirs(out, exist_ok=True)
        worst = {}
        for e in _ELEMENTWISE:
            for k, v in e['quantities'].items():
                w = worst.setdefault(k, {'scaled': 0.0, 'elementwise': 0.0})
                w['scaled'], w['elementwise'] = max(w['scaled'], v['scaled']), max(w['elementwise'], v['elementwise'])
        with open(os.path.join(out, 'elementwise_parity.json'), 'w') as f:
            json.dump({'floor': '|ref| > 1e-3 * max|ref|', 'worst_per_quantity': worst, 'compares': _ELEMENTWISE}, f, indent=1)
    except OSError:
        pass
