import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "ransac-flow_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (authoring container only)")


def pytest_collection_modifyitems(config, items):
    import ref_loader
    if not ref_loader.available():
        skip = pytest.mark.skip(reason="/root/reference not present on this machine")
        for it in items:
            if "reference" in it.keywords:
                it.add_marker(skip)


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("a test marked gpu ran without a GPU: run it with -m gpu on the GPU box")
    return torch.device("cuda:0")
