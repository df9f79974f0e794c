#!/bin/bash
# On the GPU box (gpurun -- bash tools/collect_evidence.sh): every figure profiles/r03_* and the READMEs quote -- in-step kernel tables + PMC traffic
# (tools/bench_profile.py), the per-shape roofline table (tools/layer_roofline.py), the bench lines of all workloads, the small-kernel timings.
# Outputs land in gpurun_out/; copy the ones to keep into profiles/.
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python tools/bench_profile.py lwf_resnet18_b50_task0 lwf_resnet18_b50_task1 ewc_resnet32_b50_task1 icarl_resnet32_b50_task1 inflora_vitb16_b20_task1 l2p_vitb16_b10_task1 2>&1 | tail -8
python tools/layer_roofline.py 30 > gpurun_out/layer_roofline.md 2>gpurun_out/layer_roofline.err
for w in lwf_resnet18_b50_task0 lwf_resnet18_b50_task1 ewc_resnet32_b50_task1 icarl_resnet32_b50_task1 inflora_vitb16_b20_task1 l2p_vitb16_b10_task1; do
  python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_$w.json
done
python bench.py --workload ewc_resnet32_b50_task1 --batch 32 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_ewc_resnet32_b32.json
python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_default_with_cpu.json
python tools/small_kernels.py > gpurun_out/small_kernels.txt 2>&1
ls -la gpurun_out | tail -30
