"""Dev tool: the two branches of the sampling pass from a rocprofv3 kernel trace of bench.py:
    rocprofv3 --kernel-trace --output-format csv -d /tmp/p -- python bench.py --no-cpu-baseline --no-train --no-roofline
    python scripts/pass_timeline.py $(ls -S $(find /tmp/p -name '*kernel_trace.csv') | head -1)      (.csv or .csv.gz)
A pass is delimited by its audio_stats kernel (first kernel of the audio branch).  Prints per pass: its length, the sampler's
duration, the end of the audio branch's last encoder projection and of the sampler relative to the next pass's start, and
what runs in the hand-over between two passes."""
import csv, gzip, io, sys

path = sys.argv[1]
fh = io.TextIOWrapper(gzip.open(path)) if path.endswith(".gz") else open(path)
rows = sorted(csv.DictReader(fh), key=lambda r: int(r["Start_Timestamp"]))
t0 = lambda r: int(r["Start_Timestamp"])
t1 = lambda r: int(r["End_Timestamp"])
A = [r for r in rows if "audio_stats" in r["Kernel_Name"]]
for a, b in list(zip(A, A[1:]))[-5:]:
    inside = [r for r in rows if t0(a) - 100000 <= t0(r) < t0(b) - 100000]
    smp = [r for r in inside if "prior_sample" in r["Kernel_Name"]]
    enc = [r for r in inside if "gemm_pp192" in r["Kernel_Name"]]
    if not smp or not enc:
        continue
    prev_end = max((t1(r) for r in rows if t1(r) <= t0(a) and ("gemm_pp192" in r["Kernel_Name"] or "layernorm_kernel<4, true>" in r["Kernel_Name"])), default=None)
    first_side = min((t0(r) for r in inside if "rng_fill" in r["Kernel_Name"] or "prior_sample" in r["Kernel_Name"]), default=t0(a))
    if prev_end is not None:
        print(f"   hand-over: previous pass's last encoder projection ended {(t0(a) - prev_end) / 1e3:.0f} us before this pass's first "
              f"audio kernel; the sampler branch's first kernel starts {(first_side - t0(a)) / 1e3:+.0f} us from it")
    print(f"pass {(t0(b) - t0(a)) / 1e6:.3f} ms | sampler {(t1(smp[0]) - t0(smp[0])) / 1e6:.3f} ms, ends {(t0(b) - t1(smp[0])) / 1e3:.0f} us "
          f"before the next pass | last encoder projection ends {(t0(b) - max(t1(r) for r in enc)) / 1e3:.0f} us before it")
if len(sys.argv) > 2 and len(A) > 3:     # --gap: the hand-over in detail
    b = A[-3]
    last = [r for r in rows if "gemm_pp192" in r["Kernel_Name"] and t1(r) < t0(b)][-1]
    for r in rows[rows.index(last):]:
        if t0(r) > t0(b) + 80000:
            break
        print(f"{(t0(r) - t0(b)) / 1e3:9.1f} us +{(t1(r) - t0(r)) / 1e3:8.1f}  queue {r['Queue_Id']}  {r['Kernel_Name'][:72]}")
