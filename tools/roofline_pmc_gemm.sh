#!/bin/bash
# PMC + kernel-trace passes for the ViT bench lines' roofline kernel: the fc1 GEMM [M,768] x [3072,768]^T with the bias+GELU epilogue
# (epi 3: writes GELU(x) and GELU'(x)), bf16.  Separate passes as MI355X_MICROARCH.md prescribes.
# usage (GPU box, repo root): bash tools/roofline_pmc_gemm.sh  -> gpurun_out/roofline_pmc_gemm.json  (M = 128*197 and 16*222)
export TMPDIR=/tmp
O=gpurun_out/roofline_pmc_gemm
rm -rf $O; mkdir -p $O
for M in 25216 3552; do
  ARGS="$M 3072 768 3 20"
  rocprofv3 --kernel-trace --stats -d $O/kt$M -o p -- python tools/gemm_micro.py $ARGS > $O/kt$M.log 2>&1
  rocprofv3 --pmc FETCH_SIZE -d $O/rd$M -o p -- python tools/gemm_micro.py $ARGS > $O/rd$M.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $O/wr$M -o p -- python tools/gemm_micro.py $ARGS > $O/wr$M.log 2>&1
done
python - <<PY
import sqlite3, json, glob
def db(d):
    return sqlite3.connect(glob.glob(f"$O/{d}/**/*_results.db", recursive=True)[0])
out = []
for M in (25216, 3552):
    ks = [r for r in db(f"kt{M}").execute("select name,total_calls,average from top_kernels where name like '%gemm%_kernel%'")]
    def pmc(d, name):
        c = db(d)
        cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
        kn = 'kernel_name' if 'kernel_name' in cols else 'name'
        rows = list(c.execute(f"select {kn}, value from counters_collection where counter_name='{name}' and {kn} like '%gemm%_kernel%'"))
        launches = len(rows) // max(1, len(set(r[0] for r in rows)))
        return sum(r[1] for r in rows) / max(1, launches), launches          # the head / tail kernels of one product are summed
    rd, n1 = pmc(f"rd{M}", "FETCH_SIZE"); wr, n2 = pmc(f"wr{M}", "WRITE_SIZE")
    out.append(dict(kernel="clhip_gemm_nt<bf16, bias+GELU> fc1 (gemm8_kernel for the whole rounds, gemm_nt_kernel for the rest / for fewer than 256 tiles)", shape=[M, 3072, 768], dtype="bf16",
                    rocprof=[dict(name=k[0].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0], calls=k[1], avg_us=k[2]) for k in ks],
                    FETCH_SIZE_KB_per_product=rd, WRITE_SIZE_KB_per_product=wr, pmc_launches=[n1, n2],
                    traffic_bytes_per_launch=2 * rd * 1024 + wr * 1024,
                    note="FETCH_SIZE doubled per the gfx950 correction in MI355X_MICROARCH.md; counters in KB; a product may run as a head + a tail kernel, summed"))
json.dump(out, open("gpurun_out/roofline_pmc_gemm.json", "w"), indent=1)
print(json.dumps(out))
PY
