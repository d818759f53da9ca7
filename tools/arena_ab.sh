#!/bin/bash
# one lease, four fresh processes back to back (twice, in both orders): tools/arena_ab.py
cd ${GRAFT_REPO_ROOT:-.}
for v in tensors malloc set column column set malloc tensors; do
    timeout -k 10 300 python tools/arena_ab.py $v ${1:-20} 2>&1 | grep median
done
