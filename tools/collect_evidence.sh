#!/bin/bash
# On the GPU box (gpurun -- bash tools/collect_evidence.sh): every figure profiles/rNN_* (the current round's) and the READMEs quote -- in-step kernel tables + PMC traffic
# (tools/bench_profile.py), the per-shape roofline table (tools/layer_roofline.py), the bench lines of all workloads, the small-kernel timings.
# Outputs land in gpurun_out/; copy the ones to keep into profiles/.
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
python tools/bench_profile.py lwf_resnet18_b50_task0 lwf_resnet18_b50_task1 ewc_resnet32_b50_task1 icarl_resnet32_b50_task1 lucir_resnet32_b50_task1 ewc_fisher_pass herding_b50 inflora_vitb16_b20_task1 l2p_vitb16_b10_task1 2>&1 | tail -12
python tools/layer_roofline.py 30 > gpurun_out/layer_roofline.md 2>gpurun_out/layer_roofline.err
for w in lwf_resnet18_b50_task0 lwf_resnet18_b50_task1 ewc_resnet32_b50_task1 icarl_resnet32_b50_task1 lucir_resnet32_b50_task1 ewc_fisher_pass herding_b50 inflora_vitb16_b20_task1 l2p_vitb16_b10_task1; do
  python bench.py --workload $w --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > gpurun_out/bench_$w.json
done
python bench.py --workload ewc_resnet32_b50_task1 --batch 32 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_ewc_resnet32_b32.json
# the ViT-B/16 forward + backward at batch 256 (north_star's second roofline target; SURVEY section 8(d) row 4 "+ scale-up run at 256")
python bench.py --workload l2p_vitb16_b10_task1 --batch 256 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_l2p_vitb16_b256.json
python bench.py --workload inflora_vitb16_b20_task1 --batch 256 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_inflora_vitb16_b256.json
python bench.py --workload icarl_resnet32_b50_task1 --batch 32 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_icarl_resnet32_b32.json
python tools/inflora_task_boundary.py 2400 128 > gpurun_out/inflora_task_boundary.md 2>gpurun_out/inflora_task_boundary.err
CLHIP_BN_INPUT_WT=1 python tools/wt_micro.py 256 > gpurun_out/wt_micro.md 2>&1
python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_default_with_cpu.json
python tools/small_kernels.py > gpurun_out/small_kernels.txt 2>&1
# the ViT-B/16 block GEMMs against the vendor library on this box: plain C = A B^T, and the step's fused forms against vendor GEMM + torch's elementwise kernels
python tools/gemm_vs_blas.py 30 > gpurun_out/gemm_vs_blas.txt 2>&1
# the 32-image CifarResNet-32 step (graph replay) with the launch classes removed one by one: timing only, results invalid (CLHIP_PLAN_SKIP: 1 forward BatchNorm
# apply, 2 BatchNorm backward, 4 weight gradients)
for sk in 0 1 2 4 7; do
  echo "PLAN_SKIP=$sk $(CLHIP_PLAN_SKIP=$sk python bench.py --workload ewc_resnet32_b50_task1 --batch 32 --no-cpu-baseline 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*')"
done > gpurun_out/b32_ablation.txt
# every dispatch of ONE step: the headline and the stage-level CifarResNet-32 step
bash tools/step_timeline.sh "A=1" --no-secondary > gpurun_out/step_timeline.txt 2>&1
bash tools/step_timeline.sh "A=1" --workload ewc_resnet32_b50_task1 --no-secondary > gpurun_out/step_timeline_ewc_resnet32.txt 2>&1
python tools/attn_bwd_micro.py 128 50 > gpurun_out/attn_bwd_micro.txt 2>&1
CLHIP_ATTN_BWD=1 python tools/attn_bwd_micro.py 128 50 >> gpurun_out/attn_bwd_micro.txt 2>&1
python tools/ln_micro.py 25216 200 > gpurun_out/ln_micro.txt 2>&1
python tools/bn_micro.py 64 32 256 > gpurun_out/bn_micro.txt 2>&1
ls -la gpurun_out | tail -30
