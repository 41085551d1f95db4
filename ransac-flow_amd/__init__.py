"""ransac-flow_amd: MI355X-native RANSAC-Flow coarse-to-fine alignment hot path.

The directory name is not a Python identifier; load it with ``importlib.import_module("ransac-flow_amd")``
or put this directory on ``sys.path`` and ``import rfx`` (host package) / ``import outil, model,
coarseAlignFeatMatch`` from ``dropin/`` (the reference's bare-module surface).
"""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)

import rfx  # noqa: E402,F401

DROPIN_DIR = os.path.join(_HERE, "dropin")
