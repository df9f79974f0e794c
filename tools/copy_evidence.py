"""copies the outputs of tools/collect_evidence.sh (gpurun_out/) into profiles/ under this round's prefix:  python tools/copy_evidence.py r05"""
import glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pre = sys.argv[1]
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
for fn in glob.glob(os.path.join(G, "bench_kernel_stats_*.txt")):
    shutil.copy(fn, os.path.join(P, f"{pre}_" + os.path.basename(fn)))
for a, b in (("bench_kernel_stats.json", "bench_kernel_stats.json"), ("roofline_pmc.json", "roofline_pmc.json"), ("layer_roofline.md", "layer_roofline.md"),
             ("gemm_vs_blas.txt", "gemm_vs_blas.txt"), ("small_kernels.txt", "small_kernels.txt"), ("inflora_task_boundary.md", "inflora_task_boundary.md"),
             ("b32_ablation.txt", "b32_ablation.txt"), ("step_timeline.txt", "step_timeline.txt"), ("step_timeline_ewc_resnet32.txt", "step_timeline_ewc_resnet32.txt"),
             ("attn_bwd_micro.txt", "attn_bwd_micro.txt"), ("ln_micro.txt", "ln_micro.txt"), ("bn_micro.txt", "bn_micro.txt")):
    if os.path.exists(os.path.join(G, a)):
        shutil.copy(os.path.join(G, a), os.path.join(P, f"{pre}_{b}"))
# the GPU box has no .git: stamp the commit of the kernel sources the in-step tables were taken at (the last commit that touched csrc/ or the header -- the
# tables' own `src_hash` says whether they still match) where bench_profile.py could not
import subprocess
ks = os.path.join(P, f"{pre}_bench_kernel_stats.json")
if os.path.exists(ks):
    d = json.load(open(ks))
    head = subprocess.run(["git", "-C", ROOT, "log", "-1", "--format=%h", "--", "libcontinual_amd/csrc", "include"], capture_output=True, text=True).stdout.strip() or None
    for k, v in d.items():
        if isinstance(v, dict) and v.get("head") is None:
            v["head"] = head
    json.dump(d, open(ks, "w"), indent=1)
lines = {}
for fn in sorted(glob.glob(os.path.join(G, "bench_*.json"))):
    key = os.path.basename(fn)[len("bench_"):-len(".json")]
    if key.startswith("kernel_stats"):
        continue
    txt = open(fn).read().strip().splitlines()
    if txt and txt[-1].startswith("{"):
        lines[key] = json.loads(txt[-1])
json.dump(lines, open(os.path.join(P, f"{pre}_bench_lines.json"), "w"), indent=1)
print("copied; bench lines:", sorted(lines))
