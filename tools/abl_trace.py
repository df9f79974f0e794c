"""GPU-side durations of a conv_bench ablation run: python tools/abl_trace.py <kernel_trace.csv> <reps>
(conv_bench ... abl launches, per debug mask, 5 warm-up + reps timed launches of the same kernel; the host timer of the bench is
launch-rate bound below ~7 us, the kernel-trace durations are not)"""
import csv, sys, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
reps = int(sys.argv[2])
rows = [r for r in rows if "conv" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
masks = [0, 16, 8, 24, 1, 2, 4, 64, 32, 65, 34, 67, 71, 95, 127]
per = 5 + reps
for i, m in enumerate(masks):
    chunk = rows[i * per + 5:(i + 1) * per]
    if not chunk: break
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in chunk]
    gaps = [(int(chunk[j + 1]["Start_Timestamp"]) - int(chunk[j]["End_Timestamp"])) / 1e3 for j in range(len(chunk) - 1)]
    print(f"debug {m:3d}: kernel {statistics.median(d):6.1f} us (min {min(d):5.1f})  gap to next launch {statistics.median(gaps):5.1f} us   {chunk[0]['Kernel_Name'][:60]}")
