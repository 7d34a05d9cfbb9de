#!/usr/bin/env python3
"""`make -C oracle ref` — build oracle/_ref/: the REAL reference, compiled where it lies.  TEST / BASELINE INFRASTRUCTURE.

The reference (facebookresearch/dlrm, $DLRM_REFERENCE, default /root/reference) is pure Python, so its "binary" is CPython
bytecode: every module the hot path's callers import is compiled with `py_compile` FROM ITS SOURCE FILE IN THE REFERENCE
CHECKOUT into a sourceless `.pyc` under oracle/_ref/ (git-ignored, not gpurun-ignored: it travels to the GPU box like our
own built .so files).  No reference source text is copied into this repository — only compiler output leaves the checkout,
exactly as `gcc` output would for a C reference.  CPython imports a `<module>.pyc` that sits where `<module>.py` would be
(importlib's SourcelessFileLoader), so `sys.path.insert(0, "oracle/_ref"); import dlrm_s_pytorch` gives the unmodified
reference on a box that has no checkout.  The bytecode is tied to the interpreter's magic number; the build container and the
GPU box run the same image (python 3.10), and MANIFEST.json records the magic number + the SHA-256 of every source so
`ref_dir()` refuses a stale or foreign build instead of importing it.

Users of oracle/_ref (all test / baseline infrastructure, never the product path):
  * tests/test_gpu_model.py::test_launcher_trains_under_the_unmodified_reference_run — SURVEY 8 a-11 on the GPU;
  * bench.py's baseline leg: `cpu_baseline.kind = "reference"` (the reference's own DLRM_Net on the host cores) and
    `stock_gpu_baseline` (the same unmodified module with --use-gpu semantics: stock ATen kernels on the same MI355X).
"""
from __future__ import annotations

import hashlib
import importlib.util
import json
import os
import py_compile
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
# what dlrm_s_pytorch.py imports at module level (dlrm_s_pytorch.py:71-107) + the config-5 input class
MODULES = [
    "dlrm_s_pytorch.py", "dlrm_data_pytorch.py", "extend_distributed.py", "mlperf_logger.py", "data_utils.py",
    "data_loader_terabyte.py", "optim/rwsadagrad.py", "tricks/md_embedding_bag.py", "tricks/qr_embedding_bag.py",
    "torchrec_dlrm/multi_hot.py",
]


def _sha(path: str) -> str:
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def build(reference: str | None = None, quiet: bool = False) -> str | None:
    ref = reference or os.environ.get("DLRM_REFERENCE", "/root/reference")
    if not os.path.isfile(os.path.join(ref, "dlrm_s_pytorch.py")):
        if not quiet:
            print(f"oracle/_ref: no reference checkout at {ref}; nothing built (prebuilt files, if any, are kept)")
        return None
    manifest = {"python_magic": importlib.util.MAGIC_NUMBER.hex(), "python": sys.version.split()[0],
                "reference": "facebookresearch/dlrm", "modules": {}}
    for rel in MODULES:
        src = os.path.join(ref, rel)
        dst = os.path.join(OUT, rel[:-3] + ".pyc")
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        # dfile: the name tracebacks show; the source itself stays in the checkout
        py_compile.compile(src, cfile=dst, dfile="<reference>/" + rel, doraise=True, optimize=0)
        manifest["modules"][rel] = {"sha256": _sha(src), "pyc": os.path.relpath(dst, OUT)}
    with open(os.path.join(OUT, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    if not quiet:
        print(f"oracle/_ref: {len(MODULES)} modules compiled from {ref} (python magic {manifest['python_magic']})")
    return OUT


def ref_dir() -> str | None:
    """oracle/_ref if it holds a build THIS interpreter can import, else None."""
    mf = os.path.join(OUT, "MANIFEST.json")
    if not os.path.isfile(mf):
        return None
    m = json.load(open(mf))
    if m.get("python_magic") != importlib.util.MAGIC_NUMBER.hex():
        return None
    if not all(os.path.isfile(os.path.join(OUT, v["pyc"])) for v in m["modules"].values()):
        return None
    return OUT


if __name__ == "__main__":
    sys.exit(0 if (build() or ref_dir()) else 0)
