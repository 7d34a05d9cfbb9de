"""Run the UNMODIFIED reference training script with the MI355X hot path swapped in.

    python -m dlrm_amd.launch --reference /path/to/facebookresearch/dlrm -- \
        --arch-sparse-feature-size=128 --arch-mlp-bot=13-512-256-128 --arch-mlp-top=1024-1024-512-256-1 \
        --arch-embedding-size=... --mini-batch-size=65536 --use-gpu ...
    torchrun --nproc-per-node 8 -m dlrm_amd.launch --reference ... -- ... --dist-backend=nccl --use-gpu

`run()` in the reference resolves `DLRM_Net` and `ext_dist` by module-global name at call time
(dlrm_s_pytorch.py:1076,1285,1329-1336), so the swap is two assignments: the reference keeps its CLI, data
generation, training loop, timing and printing; every device operation under `dlrm(...)`, `loss_fn`, `backward()`
and `optimizer.step()` runs in libdlrm_hip.so.  (INTEGRATION.md has the details and the raw ctypes binding.)
"""
from __future__ import annotations

import argparse
import importlib
import os
import sys
import types


def _stub_tensorboard_if_missing() -> None:
    """dlrm_s_pytorch.py:101 imports SummaryWriter unconditionally; provide a no-op when tensorboard is absent."""
    try:
        importlib.import_module("torch.utils.tensorboard")
        return
    except Exception:
        pass
    tb = types.ModuleType("torch.utils.tensorboard")

    class SummaryWriter:  # noqa: D401 - no-op stand-in
        def __init__(self, *a, **k): pass
        def add_scalar(self, *a, **k): pass
        def close(self): pass
    tb.SummaryWriter = SummaryWriter
    sys.modules["torch.utils.tensorboard"] = tb


def load_reference(reference_dir: str, device_tables: bool = False):
    """Import the reference's dlrm_s_pytorch with DLRM_Net / ext_dist replaced; returns the module."""
    import dlrm_amd
    from dlrm_amd import ext_dist

    if not any(os.path.isfile(os.path.join(reference_dir, "dlrm_s_pytorch" + ext)) for ext in (".py", ".pyc")):
        # (.pyc: a checkout compiled into a sourceless tree with py_compile)
        sys.exit("ERROR: %s does not contain dlrm_s_pytorch.py" % reference_dir)
    _stub_tensorboard_if_missing()
    if reference_dir not in sys.path:
        sys.path.insert(0, reference_dir)
    # the reference's own `import extend_distributed as ext_dist` must bind OUR implementation, so that the rank /
    # size globals DLRM_Net reads (dlrm_s_pytorch.py:252,353,518) and the ones run() sets are the same objects
    sys.modules["extend_distributed"] = ext_dist
    ref = importlib.import_module("dlrm_s_pytorch")
    ref.DLRM_Net = dlrm_amd.DLRM_Net
    ref.ext_dist = ext_dist
    if device_tables:
        import torch
        dev = torch.device("cuda", max(int(os.environ.get("LOCAL_RANK", "0")), 0))
        dlrm_amd.set_embedding_init(dev)
    return ref


def main(argv=None) -> None:
    argv = list(sys.argv[1:] if argv is None else argv)
    ref_args = []
    if "--" in argv:
        i = argv.index("--")
        argv, ref_args = argv[:i], argv[i + 1:]
    ap = argparse.ArgumentParser(prog="python -m dlrm_amd.launch")
    ap.add_argument("--reference", default=os.environ.get("DLRM_REFERENCE", "."),
                    help="checkout of facebookresearch/dlrm (directory holding dlrm_s_pytorch.py)")
    ap.add_argument("--device-tables", action="store_true",
                    help="allocate + initialise embedding tables directly in HBM (needed for Criteo-Terabyte sizes: the "
                         "reference's numpy float64 temporary does not fit host RAM); same distribution, torch RNG")
    a = ap.parse_args(argv)
    ref = load_reference(os.path.abspath(a.reference), a.device_tables)
    sys.argv = [os.path.join(a.reference, "dlrm_s_pytorch.py")] + ref_args
    ref.run()


if __name__ == "__main__":
    main()
