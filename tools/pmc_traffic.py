#!/usr/bin/env python3
"""Dev tool: profiles/hbm_traffic.json from three rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum
TCC_MISS_sum) of tools/pmc_run.py.  Usage: pmc_traffic.py <dir_fetch> <dir_write> <dir_tcc> <out.json> [note]"""
import csv, glob, json, re, sys, collections


def short_name(name):
    """rocprofv3's kernel name without return type, '(anonymous namespace)::' qualifiers and the parameter list:
    'void (anonymous namespace)::rover_scan_step_kernel<true, true, 1024, 2>((anonymous namespace)::RvParams, ...)' ->
    'rover_scan_step_kernel<true, true, 1024, 2>' -- the same string rover_kernel_names() returns."""
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void\s+", "", name)
    depth, out = 0, []
    for ch in name:
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).strip()


def per_kernel(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        for r in rows:
            key = short_name(r["Kernel_Name"])
            if key.startswith(("rover_scan_step_kernel", "rover_scan_obs_kernel<2", "rover_step_kernel", "rover_step_scan_kernel",
                               "lift_step_kernel")):
                acc[key][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    out = {}
    for k, d2 in acc.items():
        out[k] = {}
        for c, v in d2.items():
            v.sort()
            v = [x for _, x in v][2:]          # skip the first launches (cold caches)
            out[k][c] = sum(v) / max(len(v), 1)
    return out


def stamp():
    """What the counters belong to: the source digest of this checkout's librover_hip.so and the solver / mass settings of the run
    (QB_ITERS / QB_MASS of tools/pmc_run.py, else the cfg defaults) -- bench.py prints a summary only beside the build it describes."""
    import hashlib, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from isaac_rover_orbit_amd.cfg import RoverEnvCfg
    c = RoverEnvCfg()
    from isaac_rover_orbit_amd import build
    return {"_lib_sha256": build.source_digest() + ":librover_hip.so",   # sources + headers: stable across rebuilds (bench.py: library_digest)
            "_solver_iterations": int(os.environ.get("QB_ITERS") or c.solver_iterations),
            "_mass_model": os.environ.get("QB_MASS") or c.mass_model}


for d in sys.argv[1:4]:
    if not glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        sys.exit(f"{d}: no *counter_collection.csv (run rocprofv3 with --output-format csv)")
fetch, write, tcc = per_kernel(sys.argv[1]), per_kernel(sys.argv[2]), per_kernel(sys.argv[3])
res = {"_how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum (three separate passes) -- "
               "python3 tools/pmc_run.py 4096 20; per-launch means over launches 3..20, N=4096, 1x MI355X. FETCH_SIZE/WRITE_SIZE "
               "are KiB at the L2's memory side (Infinity-Cache hits included). MI355X_MICROARCH.md: on gfx950 FETCH_SIZE "
               "reports 1/2 of the bytes of a wide coalesced (16 B/lane) read stream -> doubled for the scan kernels (their "
               "tile staging is 16 B/lane global_load_lds); the step kernels' 4-byte gathers are an uncalibrated width -> raw "
               "value.  Keys = kernel names exactly as rocprofv3 prints them (minus qualifiers / parameter list) = "
               "rover_kernel_names().",
       "round": 5, "_build": sys.argv[5] if len(sys.argv) > 5 else "", **stamp()}
# The one-launch kernel (rover_step_scan_kernel) contains both read streams: the step phase's 4-byte terrain gathers (raw value)
# and the scan phase's 16-byte window staging (x 2).  Its corrected fetch = 2 x raw - the step phase's share, taken from the
# two-launch step kernel measured in the same passes (tools/pmc_run.py with ROVER_FUSED=0 runs behind the product run).
step_raw = fetch.get("rover_step_kernel_group", {}).get("FETCH_SIZE")
for k in sorted(set(fetch) & set(write) & set(tcc)):
    f, w = fetch[k]["FETCH_SIZE"], write[k]["WRITE_SIZE"]
    h, m = tcc[k].get("TCC_HIT_sum", 0.0), tcc[k].get("TCC_MISS_sum", 0.0)
    if k.startswith("rover_step_scan_kernel"):
        corr = "2 x raw - step-phase share" if step_raw is not None else "2 (upper bound: the step phase's gathers doubled too)"
        fetch_kib = 2.0 * f - (step_raw or 0.0)
    else:
        corr = 2.0 if k.startswith("rover_scan") else 1.0
        fetch_kib = f * corr
    res[k] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "fetch_correction": corr,
              "bytes_per_launch": int((fetch_kib + w) * 1024), "l2_hit_rate": h / max(h + m, 1.0)}
json.dump(res, open(sys.argv[4], "w"), indent=1)
print(json.dumps(res, indent=1))
