"""Per-SHAPE rows of a rocprofv3 kernel trace of the headline pass, so that every figure of the bench line's `roofline` can be
recomputed from profiles/ alone (a --stats summary averages a kernel over every problem shape it was launched with):

    cd /tmp && export TMPDIR=/tmp
    AVI_BENCH_MARKERS=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof -- python3 bench.py --legs none \
            --no-roofline --steps 20
    python3 scripts/trace_by_shape.py $(ls -S $(find gpurun_out/prof -name '*kernel_trace.csv') | head -1) 20 \
            > profiles/r04_headline_trace_by_shape.csv

Only the launches of the TIMED REGION are kept: with AVI_BENCH_MARKERS=1 bench.py launches one `where_kernel` in front of and
one behind it (capture-time candidates, warm-up and the parity passes behind the timed region stay out); `passes` = --steps.
Without markers in the trace the `passes` back-to-back passes (delimited by their audio_stats kernel) with the shortest total
duration are taken.  One row per
(kernel, grid): launches per pass, average / min / max duration in us, time per pass in ms, and - from the first and last
timestamp of the kept passes - the measured pass period."""
import csv
import gzip
import io
import sys
from collections import defaultdict

path, passes = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 20
fh = io.TextIOWrapper(gzip.open(path)) if path.endswith(".gz") else open(path)
rows = sorted(csv.DictReader(fh), key=lambda r: int(r["Start_Timestamp"]))
starts = [int(r["Start_Timestamp"]) for r in rows if "audio_stats" in r["Kernel_Name"]]
if len(starts) < passes + 1:
    raise SystemExit(f"the trace holds {len(starts)} passes, {passes} + 1 needed")
marks = [int(r["Start_Timestamp"]) for r in rows if "where_kernel" in r["Kernel_Name"]]
if len(marks) == 2:
    lo, hi = marks[0] + 1, marks[1]                # exactly the timed region
else:
    k = min(range(len(starts) - passes), key=lambda i: starts[i + passes] - starts[i])
    lo, hi = starts[k], starts[k + passes]         # [first kept pass, the pass after the last kept one)


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0]


acc = defaultdict(list)
for r in rows:
    t0 = int(r["Start_Timestamp"])
    # the sampler's branch and the head of pass k run beside pass k's / k + 1's audio branch: a launch belongs to the window
    # its START falls into, which is exact for whole-pass sums over many passes
    if lo <= t0 < hi:
        grid = "x".join(str(int(r[f"Grid_Size_{a}"]) // max(1, int(r[f"Workgroup_Size_{a}"]))) for a in "XYZ")
        acc[(short(r["Kernel_Name"]), grid, r["Workgroup_Size_X"], r["LDS_Block_Size"], r["VGPR_Count"])].append(
            (int(r["End_Timestamp"]) - t0) / 1e3)
w = csv.writer(sys.stdout)
w.writerow(["kernel", "workgroups (x x y x z)", "threads", "lds_bytes", "vgprs", "launches_per_pass", "avg_us", "min_us", "max_us",
            "ms_per_pass"])
for (k, grid, thr, lds, vg), d in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    w.writerow([k, grid, thr, lds, vg, round(len(d) / passes, 2), round(sum(d) / len(d), 2), round(min(d), 2), round(max(d), 2),
                round(sum(d) / passes / 1e3, 4)])
w.writerow(["# timed region / passes, ms" if len(marks) == 2 else "# pass period (audio_stats to audio_stats), ms",
            round((hi - lo) / passes / 1e6, 4), "passes", passes, "", "", "", "", "", ""])
