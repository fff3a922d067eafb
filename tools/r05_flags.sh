#!/bin/bash
# Dev tool (GPU box), round 5: scheduler strategies per translation unit -- rover step (4096 envs + the n sweep's other kernels), policy pair, lift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r05_flags2}; mkdir -p $O
cd $R
TAGS="F_MAXILP F_MAXMEM" bash tools/r05_ab.sh $1 || exit 1
for t in "" P_MAXILP; do POLTAG=$t timeout -k 10 120 python3 tools/pair_time.py >> $O/pair.txt 2>&1 || exit 1; done
for t in "" L_MAXILP; do ABLTAG=$t timeout -k 10 120 python3 tools/lift_time.py 2048 >> $O/lift.txt 2>&1 || exit 1; done
grep -E "pair [0-9]" $O/pair.txt; grep "us per step" $O/lift.txt
for t in "" F_MAXILP; do ABLTAG=$t timeout -k 10 300 python3 tools/n_sweep.py > $O/n_sweep_$t.txt 2>&1 || exit 1; done
python3 - <<PY
import json
for t in ("", "F_MAXILP"):
    for l in open("$O/n_sweep_%s.txt" % t):
        if l.startswith("{"):
            d = json.loads(l); print(t or "product", d["num_envs"], d["mapping"], d["kernels"][0][:28], round(d["env_steps_per_s"] / 1e6, 1))
PY
