"""kernel resource table of one HIP source: python tools/kres.py libcontinual_amd/csrc/conv4.hip [filter]
(VGPR / AGPR / SGPR, spills, scratch, LDS, occupancy from hipcc -Rpass-analysis=kernel-resource-usage)"""
import re, subprocess, sys
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-c", src, "-o", "/dev/null",
                      "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
cur = None; rows = []
for line in out.splitlines():
    m = re.search(r"remark: [^ ]+ \[|remark: +(.*?) \[-Rpass", line)
    m = re.search(r"remark: (.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        name = subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(anonymous namespace\)::|\(.*$", "", name)}; rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
print(f"{'kernel':60s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'vsp':>4s} {'ssp':>4s} {'scr':>5s} {'occ':>4s} {'LDS':>7s}")
for r in rows:
    if flt in r["name"]:
        print(f"{r['name'][:60]:60s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {r.get('SGPRs','?'):>5s} {r.get('VGPRs Spill','?'):>4s} {r.get('SGPRs Spill','?'):>4s} "
              f"{r.get('ScratchSize [bytes/lane]','?'):>5s} {r.get('Occupancy [waves/SIMD]','?'):>4s} {r.get('LDS Size [bytes/block]','?'):>7s}")
