"""Per-SHAPE rows of a rocprofv3 kernel trace of the headline pass, so that every figure of the bench line's `roofline` can be
recomputed from profiles/ alone (a --stats summary averages a kernel over every problem shape it was launched with):

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof -- python3 bench.py --legs none --no-roofline --steps 20
    python3 scripts/trace_by_shape.py $(ls -S $(find gpurun_out/prof -name '*kernel_trace.csv') | head -1) 20 \
            > profiles/r04_headline_trace_by_shape.csv

Only the LAST `passes` passes of the trace are kept (= the timed region of `bench.py`: passes are delimited by their
audio_stats kernel, the first kernel of the audio branch), so capture-time candidates and warm-up do not mix in.  One row per
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
lo, hi = starts[-passes - 1], starts[-1]          # [first kept pass, the pass after the last kept one)


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
w.writerow(["# pass period (audio_stats to audio_stats), ms", round((hi - lo) / passes / 1e6, 4), "passes", passes, "", "", "", "", "", ""])
