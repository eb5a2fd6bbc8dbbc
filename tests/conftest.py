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
