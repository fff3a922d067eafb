#!/usr/bin/env python3
"""Dev tool (build container): static instruction mix of one kernel of a hipcc -S listing, split at the s_memtime stamps of the
diagnostic build (-DRV_K1_STAMP) -- which phase of the step kernel holds how many VALU / SALU / LDS / VMEM instructions.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only -DRV_K1_STAMP -o /tmp/k1stamp.s rover_kernels.hip
    python tools/isa_segments.py /tmp/k1stamp.s rover_step_scan_kernelILb1E

Loops are counted once (the solver loop body holds two iterations and runs iterations / 2 times; the substep loop body runs
decimation - 1 times): the listing is static.  8-byte encodings (VOP3 / packed / DPP / SDWA) are counted separately -- on
gfx950 with one wave per SIMD they cost ~5.5 cycles of issue against ~4.5 for a 4-byte VOP1 / VOP2 (docs/history.md 3.2)."""
import re, sys, collections

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and ":" in l.split(";")[0])
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith((".section", ".amdhsa_kernel")))
seg, segs, label = collections.Counter(), [], "entry"


def klass(op):
    if op.startswith("v_"):
        wide = op.endswith(("_e64", "_dpp", "_sdwa")) or op.startswith(("v_pk_", "v_fma_", "v_med3", "v_mad_", "v_add3", "v_lshl_add", "v_div_", "v_cndmask_b32_e64", "v_readlane", "v_writelane", "v_bfe", "v_alignbit", "v_and_or", "v_lshl_or", "v_perm"))
        return "valu8" if wide else "valu4"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith(("s_load", "s_buffer_load", "s_memtime", "s_store")):
        return "smem"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_load_lds", "buffer_load")) and "lds" in op:
        return "dma"
    if op.startswith(("global_load", "flat_load", "buffer_load", "scratch_load")):
        return "vmem_rd"
    if op.startswith(("global_store", "flat_store", "buffer_store", "scratch_store", "global_atomic")):
        return "vmem_wr"
    return "other"


for l in lines[start + 1:end]:
    s = l.strip()
    if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
        if s.endswith(":") and not s.startswith(";"):
            seg["labels"] += 1
        continue
    op = s.split()[0]
    if op == "s_memtime":
        segs.append((label, seg))
        seg, label = collections.Counter(), f"stamp@{len(segs)}"
        continue
    seg[klass(op)] += 1
segs.append((label, seg))
cols = ["valu4", "valu8", "salu", "lds", "dma", "vmem_rd", "vmem_wr", "smem", "wait", "barrier", "branch", "nop", "labels"]
print(f"{'segment':12s} " + " ".join(f"{c:>7s}" for c in cols) + "   ~issue cycles (4.5 / 5.5 per VALU)")
tot = collections.Counter()
for name, c in segs:
    tot.update(c)
    est = 4.5 * c["valu4"] + 5.5 * c["valu8"]
    print(f"{name:12s} " + " ".join(f"{c[k]:7d}" for k in cols) + f"   {est:8.0f}")
print(f"{'total':12s} " + " ".join(f"{tot[k]:7d}" for k in cols))
