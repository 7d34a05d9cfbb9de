#!/usr/bin/env python3
"""Which part of the one-lookup-per-bag proof disturbs the whole-step HIP graph at D = 16 (tests/test_gpu_model.py::
test_graphed_step_equals_eager_step)?  Variants of ops.offsets_are_iota: V0 no device work, V1 stream synchronise only, V2 kernel only,
V3 the real thing."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pytest  # noqa: E402

from dlrm_amd import ops  # noqa: E402

real = ops.offsets_are_iota


def v0(o):
    return True


def v1(o):
    torch.cuda.current_stream().synchronize()
    return True


def v2(o):
    t = o if isinstance(o, torch.Tensor) else o[0]
    (t + 0).sum()                      # some kernel on the current stream, no synchronisation
    return True


for name, fn in (("V0 none", v0), ("V1 sync only", v1), ("V2 kernel only", v2), ("V3 real", real)):
    ops.offsets_are_iota = fn
    ops._iota_cache.clear()
    rc = pytest.main(["-q", "-x", "-m", "gpu", os.path.join(ROOT, "tests/test_gpu_model.py"), "-k", "graphed_step_equals_eager_step and deterministic",
                      "-p", "no:cacheprovider"])
    print("RESULT", name, "rc =", int(rc), flush=True)
