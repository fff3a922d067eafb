#!/bin/bash
# Dev tool (GPU box): the parity files on the build variants kept in the source (tools/build_diag.py BARRIERS SPANS): every form gives the same bits
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_variants; mkdir -p $O; cd $R
for t in BARRIERS SPANS; do
timeout -k 10 600 python3 -c "
import sys, pytest
import isaac_rover_orbit_amd._lib as L
L.LIB_PATH = '$R/build/abl/librover_abl$t.so'
sys.exit(pytest.main(['tests/test_gpu_parity.py', 'tests/test_gpu_configs.py', '-x', '-q', '-m', 'gpu']))" > $O/$t.log 2>&1; rc=$?; echo $t; tail -1 $O/$t.log
[ $rc -eq 0 ] || exit $rc
done
