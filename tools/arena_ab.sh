#!/bin/bash
# one lease, fresh processes back to back (in both orders): tools/arena_ab.py
cd ${GRAFT_REPO_ROOT:-.}
for v in tensors malloc set column auto auto column set malloc tensors; do
    timeout -k 10 300 python tools/arena_ab.py $v ${1:-20} 2>&1 | grep median
done
