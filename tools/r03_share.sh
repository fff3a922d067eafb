#!/bin/bash
# Dev tool (GPU box): split of an env's sixteen ray rounds between a step wave and its copy wave (tools/build_diag.py SHARE_*).
R=$GRAFT_REPO_ROOT; cd $R
for t in "" SHARE_8_14 SHARE_8_16 SHARE_7_14 SHARE_6_13 SHARE_16_16 ""; do
  echo "== ${t:-product (8, 12)}"; ABLTAG=$t timeout -k 10 120 python3 tools/fused_probe.py 4096 2>&1 | grep "one launch\|different"
done
