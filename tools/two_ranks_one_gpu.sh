#!/bin/bash
# Exercises bench.py's multi-rank code path (self-launch of the ranks, rendezvous, barriers, max-over-ranks timing, the
# training leg's gradient all-reduce, the rank-0 JSON line) on a ONE-GPU box: `python3 bench.py --gpus 2` starts two ranks
# itself, both land on cuda:0 (LOCAL_RANK modulo the device count), collectives over gloo.  The throughput it prints is
# meaningless (two ranks share one GPU); the point is that the N > 1 path runs by the plain command the driver uses and
# prints one well-formed line.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
SO3X_DIST_BACKEND=gloo python3 bench.py --gpus 2 --steps 200 --warmup 100 --batch-log2 18
echo "rc=$?"
