#!/bin/bash
# Dev tool (GPU box), round 5: the span-limited window copy (build/abl SPANS) -- its test on the product and on the variant, us per step of
# product / PREV / SPANS, and FETCH_SIZE of both (the ratio is what is wanted; units as rocprofv3 reports them)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r05_spans}; mkdir -p $O
cd $R
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "window_spans" > $O/spans_product.log 2>&1; rc=$?; tail -3 $O/spans_product.log
if [ $rc -ne 0 ]; then echo "product rc=$rc"; exit $rc; fi
timeout -k 10 600 python3 -c "
import sys, pytest
import isaac_rover_orbit_amd._lib as L
L.LIB_PATH = '$R/build/abl/librover_ablSPANS.so'
sys.exit(pytest.main(['tests/test_gpu_parity.py', 'tests/test_gpu_configs.py', '-x', '-q', '-m', 'gpu']))" > $O/spans_variant.log 2>&1; rc=$?; tail -3 $O/spans_variant.log
if [ $rc -ne 0 ]; then echo "variant rc=$rc"; exit $rc; fi
TAGS="PREV SPANS" bash tools/r05_ab.sh $1 || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f_product -- python3 $R/tools/pmc_run.py 4096 20 > $O/f_product.log 2>&1 || exit 1
export ABLTAG=SPANS
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f_spans -- python3 $R/tools/pmc_run.py 4096 20 > $O/f_spans.log 2>&1 || exit 1
unset ABLTAG
python3 - <<PY > $O/fetch.txt
import csv, glob
for tag in ("product", "spans"):
    v = []
    for f in glob.glob("$O/f_%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            if "rover_step_scan_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
                v.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    v.sort(); v = [x for _, x in v][2:]
    print(tag, "FETCH_SIZE per launch (as reported):", sum(v) / len(v), "over", len(v), "launches")
PY
cat $O/fetch.txt; rm -rf $O/f_product $O/f_spans
