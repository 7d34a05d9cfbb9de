#!/bin/bash
# r2b: new kernels (row-wise Adagrad, metrics, multi-tensor SGD, loss scale) + model goldens
TAG=${1:-r2b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
echo "== pytest new"; timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 -p no:cacheprovider -k "adagrad or dense_optimizer or binary_metrics or training_matches or smoke" > $OUT/pytest.log 2>&1; echo "rc=$?"; tail -25 $OUT/pytest.log
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3
