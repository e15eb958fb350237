#!/bin/bash
# Exercises bench.py's multi-rank code path (barriers, max-over-ranks timing, rank-0 JSON line) on a ONE-GPU box: two
# processes, RANK 0/1, both on cuda:0 (LOCAL_RANK=0), collectives over gloo.  The throughput it prints is meaningless
# (two ranks share one GPU); the point is that the N > 1 path runs and prints one well-formed line.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29581 WORLD_SIZE=2 LOCAL_RANK=0 SO3X_DIST_BACKEND=gloo
RANK=1 python3 bench.py --gpus 2 --steps 200 --warmup 100 --batch-log2 18 > /tmp/rank1.out 2>&1 &
p1=$!
RANK=0 python3 bench.py --gpus 2 --steps 200 --warmup 100 --batch-log2 18
rc=$?
wait $p1; rc1=$?
echo "rank0 rc=$rc rank1 rc=$rc1"; tail -2 /tmp/rank1.out
