#!/bin/bash
# Dev tool (GPU box): rocprofv3 kernel durations of the scan kernel's ablation builds (tools/build_diag.py NOCOPY / NORAYS / NOCOPYRAYS).
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for tag in ${K2ABL_TAGS:-"" NOCOPY NORAYS NOCOPYRAYS}; do
  O=$R/gpurun_out/k2abl_${tag:-product}; mkdir -p $O
  ABLTAG=$tag rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/pmc_run.py 4096 200 > $O/log.txt 2>&1
  f=$(find $O -name "*kernel_stats.csv" | head -1)
  echo "== ${tag:-product}"; python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.reader(open(sys.argv[1])))[1:3]:
    print("  ", r[0][:60].replace("(anonymous namespace)::",""), r[1], r[3], r[5], r[6])
PY
  rm -rf $O
done
