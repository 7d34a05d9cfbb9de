"""Inference pass with the metrics computed on the device.

Restates `inference()` of the reference (dlrm_s_pytorch.py:759-899): forward every test batch, gather the
batch-split predictions of all ranks (`ext_dist.all_gather`, :806), then
  * plain mode      : accuracy = #{round(score) == target} / #samples                      (:819-823, :858)
  * mlperf mode     : recall, precision, f1, average precision, ROC-AUC, accuracy          (:828-847)
The reference moves every prediction to the host and calls numpy / scikit-learn; here predictions stay in HBM and one
`dlrm_binary_metrics` call (rocPRIM sort + scans + fp64 reductions) produces all numbers — a single small D2H copy per
evaluation instead of one per batch.
"""
from __future__ import annotations

from typing import Dict, Iterable, Optional

import torch

from . import ext_dist, ops


@torch.no_grad()
def inference(model, test_batches: Iterable, device: Optional[torch.device] = None, nbatches: int = -1) -> Dict[str, float]:
    """`test_batches` yields (X, lS_o, lS_i, T) like the reference's unpack_batch (:733-757).  Returns the
    validation_results dictionary of the reference plus `round_accuracy` (its non-mlperf accuracy) and raw counts."""
    scores, targets = [], []
    for i, (X, lS_o, lS_i, T) in enumerate(test_batches):
        if 0 < nbatches <= i:
            break
        if ext_dist.is_distributed() and X.size(0) % ext_dist.my_size != 0:
            print("Warning: Skiping the batch %d with size %d" % (i, X.size(0)))       # reference :784-787
            continue
        if device is not None:
            X = X.to(device)
            lS_o = [o.to(device) for o in lS_o] if isinstance(lS_o, (list, tuple)) else lS_o.to(device)
            lS_i = [x.to(device) for x in lS_i] if isinstance(lS_i, (list, tuple)) else lS_i.to(device)
            T = T.to(device)
        Z = model(X, lS_o, lS_i)
        if ext_dist.is_distributed():
            _, batch_split_lengths = ext_dist.get_split_lengths(X.size(0))
            Z = ext_dist.all_gather(Z, batch_split_lengths)
        scores.append(Z.detach().reshape(-1))
        targets.append(T.reshape(-1).to(Z.device, torch.float32))
    if not scores:
        raise RuntimeError("dlrm_amd.evaluate.inference: no test batch was evaluated")
    S = scores[0] if len(scores) == 1 else torch.cat(scores)
    Y = targets[0] if len(targets) == 1 else torch.cat(targets)
    return ops.binary_metrics(S.contiguous(), Y.contiguous())
