"""Fold one rocprofv3 PMC pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE) of `bench.py` into the matrix-core busy
fraction per kernel:

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_mfma \
        -- python bench.py --no-cpu-baseline --no-train --steps 3 --warmup 1
    python scripts/pmc_mfma.py gpurun_out/pmc_mfma profiles/r02_pmc_mfma.json

busy = sum over SIMDs of the cycles its matrix pipe was busy / (kernel cycles x 1024 SIMDs).  MI355X_MICROARCH.md:
SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD; rocprofv3 reports GRBM_GUI_ACTIVE as the SUM over the 8 XCDs, so the
kernel's cycles are GRBM_GUI_ACTIVE / 8 (the quotient reads high on dispatches shorter than ~0.3 ms).  A kernel that holds
32 of the 256 CUs (the sampler) can reach 0.125 at most; the GEMMs run beside it on the other 224."""
import csv, glob, json, os, sys
from collections import defaultdict


def main():
    d, out_path = sys.argv[1], sys.argv[2]
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {d}")
    per = defaultdict(lambda: defaultdict(dict))                       # kernel -> dispatch -> counter -> value
    for f in files:
        for r in csv.DictReader(open(f)):
            for key in (r["Kernel_Name"], (r["Kernel_Name"], int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])))):
                c = per[key][r["Dispatch_Id"]]
                c[r["Counter_Name"]] = c.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    out, shapes = {}, {}
    for k, disp in per.items():
        busy = sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for v in disp.values())
        act = sum(v.get("GRBM_GUI_ACTIVE", 0.0) for v in disp.values()) / 8.0
        if act <= 0 or busy <= 0:
            continue
        kn = k[0] if isinstance(k, tuple) else k
        name = kn.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        rec = {"launches": len(disp), "mfma_busy_cycles_per_launch": round(busy / len(disp)),
               "kernel_cycles_per_launch": round(act / len(disp)), "mfma_busy_frac": round(busy / (act * 1024.0), 4)}
        if isinstance(k, tuple):      # per problem shape: the launch's workgroup count tells the conv layers / M = 4000 / 8000 apart
            shapes[f"{name} @ {k[1]} workgroups"] = rec
        else:
            out[name] = rec
    doc = {"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE over bench.py; busy fraction = busy cycles / "
                     "(GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)", "kernels": out,
           "by_shape": dict(sorted(shapes.items(), key=lambda kv: -kv[1]["mfma_busy_cycles_per_launch"] * kv[1]["launches"])[:60])}
    with open(out_path, "w") as fh:
        json.dump(doc, fh, indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["mfma_busy_cycles_per_launch"] * kv[1]["launches"])[:12]:
        print(f"{k[:56]:56s} n={v['launches']:5d} busy {v['mfma_busy_frac']:.3f}  kernel cycles {v['kernel_cycles_per_launch']}")


if __name__ == "__main__":
    main()
