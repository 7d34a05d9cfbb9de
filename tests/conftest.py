import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    d = {k: z[k] for k in z.files}
    meta = json.loads(bytes(d.pop("meta")).decode())
    return d, meta


def golden_batches(d, meta, n_tables=None):
    T = n_tables if n_tables is not None else len(meta["ln_emb"])
    out = []
    for s in range(meta["steps"]):
        out.append((d[f"s{s}.X"], [d[f"s{s}.off{k}"] for k in range(T)], [d[f"s{s}.idx{k}"] for k in range(T)],
                    d[f"s{s}.T"]))
    return out


def params_with_prefix(d, prefix):
    p = prefix + "."
    return {k[len(p):]: v for k, v in d.items() if k.startswith(p)}


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
