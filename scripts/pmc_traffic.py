"""Fold the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of `bench.py` into per-kernel HBM bytes per launch.

Usage on the GPU box (separate passes: FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2):
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python bench.py ...
    python scripts/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_traffic.json

Corrections (MI355X_MICROARCH.md, HBM): both counters are in KiB; on gfx950 FETCH_SIZE tallies the 128-B requests
of wide coalesced reads at 64 B, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane stores."""
import csv, glob, json, os, sys
from collections import defaultdict


def fold(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {d}")
    acc = defaultdict(lambda: [0.0, set()])
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            # per kernel, and per (kernel, problem shape): Grid_Size is the launch's thread count
            for key in (r["Kernel_Name"], (r["Kernel_Name"], int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])))):
                a = acc[key]
                a[0] += float(r["Counter_Value"])
                a[1].add(r["Dispatch_Id"])
    return {k: (v[0], len(v[1])) for k, v in acc.items()}


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0]


def main():
    fetch, write = fold(sys.argv[1], "FETCH_SIZE"), fold(sys.argv[2], "WRITE_SIZE")
    out, shapes = {}, {}
    for k in sorted(set(fetch) | set(write), key=str):
        fk, fn = fetch.get(k, (0.0, 0))
        wk, wn = write.get(k, (0.0, 0))
        n = max(fn, wn)
        if not n:
            continue
        rd = fk * 1024.0 * 2.0 / max(fn, 1)          # KiB -> B, gfx950 half-count correction
        wr = wk * 1024.0 / max(wn, 1)
        rec = {"launches": n, "fetch_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
               "hbm_bytes_per_launch": round(rd + wr)}
        if isinstance(k, tuple):
            shapes[f"{short(k[0])} @ {k[1]} workgroups"] = rec
        else:
            out[short(k)] = rec
    doc = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over bench.py; FETCH_SIZE x 2 "
                     "(gfx950 counts 128-B read requests as 64 B), both counters in KiB",
           "kernels": out,
           "by_shape": dict(sorted(shapes.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:60])}
    with open(sys.argv[3], "w") as fh:
        json.dump(doc, fh, indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]:
        print(f"{k[:60]:60s} n={v['launches']:5d} read {v['fetch_bytes_per_launch']/1e6:9.1f} MB  write {v['write_bytes_per_launch']/1e6:9.1f} MB per launch")


if __name__ == "__main__":
    main()
