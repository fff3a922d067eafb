#!/bin/bash
# Dev tool (GPU box): stamps of several diagnostic builds in one call.  usage: r04_x.sh outdir TAG...
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; shift; cd $R
for t in "$@"; do K1TAG=$t timeout -k 10 120 python3 tools/k1_stamps.py > $O/stamps_$t.txt 2>&1 || { echo "$t failed"; exit 1; }; echo "== $t"; tail -7 $O/stamps_$t.txt; done
