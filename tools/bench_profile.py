"""rocprofv3 evidence for bench.py's roofline block, one command per pass as MI355X_MICROARCH.md prescribes (on the GPU box, repo root):

    python tools/bench_profile.py [workload ...]

* `rocprofv3 --kernel-trace --stats -- python bench.py --workload W --no-cpu-baseline --no-secondary` -> gpurun_out/bench_kernel_stats_<W>.txt (the
  per-symbol table) and gpurun_out/bench_kernel_stats.json {W: {steps, ms_per_step, kernels: {symbol: {calls, avg_us, pct}}}} -- the
  in-step average durations bench.py reports next to its stand-alone measurements;
* `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, no trace domains) over `bench.py --roofline-only` ->
  gpurun_out/roofline_pmc.json {key: {traffic_bytes_per_launch, ...}} for the roofline kernels of the default workload.
Copy both into profiles/ (rNN_bench_kernel_stats.json, rNN_roofline_pmc.json, rNN_bench_kernel_stats_<W>.txt; bench.py reads the newest round's)."""
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
ENV = dict(os.environ, TMPDIR="/tmp")


def clean(sym):
    return re.sub(r"\(anonymous namespace\)::|^void |\(.*$", "", sym).strip()


def prof(args, tag, extra):
    d = f"/tmp/bp_{tag}"
    shutil.rmtree(d, ignore_errors=True)
    r = subprocess.run(["rocprofv3", *extra, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.join(ROOT, "bench.py"), *args],
                       env=ENV, cwd="/tmp", capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(r.stdout[-2000:], r.stderr[-2000:])
        raise SystemExit(f"bench.py under rocprofv3 ({tag}) printed no JSON line")
    return d, json.loads(line[-1])


def kernel_stats(workload):
    d, bench = prof(["--workload", workload, "--no-cpu-baseline", "--no-secondary"], "kt", ["--kernel-trace", "--stats"])
    rows = list(csv.DictReader(open(glob.glob(f"{d}/**/*kernel_stats.csv", recursive=True)[0])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    ks, lines = {}, []
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
        n = clean(r["Name"])
        ks[n] = dict(calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3, pct=100.0 * float(r["TotalDurationNs"]) / tot)
        lines.append(f"{n[:100]:100s} calls {int(r['Calls']):7d}  avg {float(r['AverageNs']) / 1e3:9.2f} us  total {float(r['TotalDurationNs']) / 1e6:9.2f} ms  {ks[n]['pct']:5.1f} %")
    nsteps = bench["steps"] + bench["warmup"]
    head = (f"rocprofv3 --kernel-trace --stats -- python bench.py --workload {workload} --no-cpu-baseline --no-secondary\n"
            f"bench line under the profiler: {bench['value']:.0f} img/s, {bench['ms_per_step']:.3f} ms/step, {bench['steps']} timed + {bench['warmup']} warm-up steps\n"
            f"kernel time in the whole run {tot / 1e6:.1f} ms (includes the roofline block's stand-alone launches and start-up)\n")
    open(os.path.join(OUT, f"bench_kernel_stats_{workload}.txt"), "w").write(head + "\n".join(lines) + "\n")
    sys.path.insert(0, ROOT)
    import bench as B
    head = None
    try:
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except Exception:
        pass
    head = head or os.environ.get("CLHIP_HEAD")                   # (the GPU box has no .git: pass the commit in the environment)
    return dict(steps=bench["steps"], warmup=bench["warmup"], ms_per_step_under_profiler=bench["ms_per_step"], kernels=ks, src_hash=B._src_hash(), head=head)


def pmc(workload):
    res = {}
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d, blk = prof(["--workload", workload, "--roofline-only"], ctr, ["--pmc", ctr])
        agg = {}
        for fn in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(fn)):
                if r["Counter_Name"] != ctr:
                    continue
                a = agg.setdefault(clean(r["Kernel_Name"]), [0, 0.0])
                a[0] += 1; a[1] += float(r["Counter_Value"])
        vals[ctr] = {k: (v[1] / v[0], v[0]) for k, v in agg.items()}
    entries = [blk["roofline"]] + blk["roofline_more"]
    for e in entries:
        lab = e["kernel"]
        key = e["pmc_key"]
        sym0 = re.sub(r",", ", ", lab.split(" ")[0])
        syms = (sym0, "wgrad3_reduce_kernel") if key.startswith("wgrad") else (sym0,)
        rd = wr = 0.0
        found = []
        for s_ in syms:
            for k, (v, n) in vals["FETCH_SIZE"].items():
                if k.startswith(s_):
                    rd += v; found.append((k, n))
            for k, (v, n) in vals["WRITE_SIZE"].items():
                if k.startswith(s_):
                    wr += v
        res[key] = dict(kernel=lab, symbols=found, FETCH_SIZE_KB_avg=rd, WRITE_SIZE_KB_avg=wr, traffic_bytes_per_launch=2 * rd * 1024 + wr * 1024,
                        algorithmic_bytes_per_launch=e["algorithmic_bytes_per_launch"],
                        note="FETCH_SIZE doubled per the gfx950 correction in MI355X_MICROARCH.md; counters in KB; the averages include the warm-up launches")
    return res


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    wl = sys.argv[1:] or ["lwf_resnet18_b50_task0"]
    pj = os.path.join(OUT, "bench_kernel_stats.json")
    # (the GPU box starts with an empty gpurun_out/: begin from the committed table so that a partial re-run keeps the other workloads)
    committed = next((c for c in (os.path.join(ROOT, "profiles", f"r0{r}_bench_kernel_stats.json") for r in (5, 4)) if os.path.exists(c)), "")
    stats = json.load(open(pj)) if os.path.exists(pj) else (json.load(open(committed)) if committed else {})
    for w in wl:
        stats[w] = kernel_stats(w)
        json.dump(stats, open(pj, "w"), indent=1)
        print(w, "kernel stats done", flush=True)
    if not any(("vitb16" in w or w in ("ewc_fisher_pass", "herding_b50")) for w in wl[:1]):
        json.dump(pmc(wl[0]), open(os.path.join(OUT, "roofline_pmc.json"), "w"), indent=1)
        print("pmc done")
