"""Timing of InfLoRA_OPT's task-boundary work on one MI355X:
the two extra passes over a task's data (before_task: Gram of every attention input -> SVD -> lora_A; after_task: merge, Gram again ->
DualGPM) that the reference runs with a 768 x 768 transfer per layer per batch (core/model/backbone/transformer.py:241-244,
core/model/InfLoRA_opt.py:246-248, 296-299).  ViT-B/16, 224 x 224, random weights, synthetic images resident in HBM.

    python tools/inflora_task_boundary.py [images_per_task=2400] [batch=128]  > gpurun_out/inflora_task_boundary.md
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libcontinual_amd.model as M  # noqa: E402


def sync_time(fn, reps=1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 2400
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    dev = "cuda:0"
    torch.manual_seed(0)
    bb = M.vit_pt_imnet(pretrained=False, attn_layer="MultiHeadAttention_LoRA", lora_rank=10, dtype="bf16")
    m = M.InfLoRA_OPT(bb, dev, init_cls_num=20, inc_cls_num=20, task_num=10, lame=1.0, lamb=0.95, dataset="imagenet-r", use_ca=False, embd_dim=768)
    m._network.to(dev)
    nb = (n_img + B - 1) // B
    def task_data(seed):
        g = torch.Generator(device=dev).manual_seed(seed)
        return [{"image": torch.randn(min(B, n_img - i * B), 3, 224, 224, device=dev, generator=g), "label": torch.zeros(min(B, n_img - i * B), dtype=torch.long, device=dev)}
                for i in range(nb)]
    batches = task_data(1)
    net = m._network
    x = batches[0]["image"]
    with torch.no_grad():
        for _ in range(2):
            net.backbone(x); net.update_input_matrix(x)
        for a in m.attention_modules:
            a.reset_input_matrix()
        t_fwd = sync_time(lambda: net.backbone(x), 10)
        t_gram = sync_time(lambda: net.update_input_matrix(x), 10)
        for a in m.attention_modules:
            a.reset_input_matrix()
        # the Gram launch alone (HIP events around the C call)
        from libcontinual_amd._lib import call
        s = bb.feat._s
        M_rows = x.shape[0] * 197
        h = torch.randn(12, M_rows, 768, device=dev).to(torch.bfloat16)
        G = torch.zeros(12, 768, 768, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            call("clhip_gram_accum_batched", h.data_ptr(), M_rows * 768, 12, G.data_ptr(), M_rows, 768, 0, st)
        ev0.record()
        for _ in range(10):
            call("clhip_gram_accum_batched", h.data_ptr(), M_rows * 768, 12, G.data_ptr(), M_rows, 768, 0, st)
        ev1.record(); torch.cuda.synchronize()
        t_k = ev0.elapsed_time(ev1) / 10 * 1e-3
        del h, G
    flop = 12 * 2.0 * M_rows * 768 * 768
    print(f"# InfLoRA_OPT task boundary on one MI355X: ViT-B/16, bf16, batch {B}, {n_img} images per task\n")
    print("| quantity | time | note |")
    print("|---|---|---|")
    print(f"| forward-only pass, one batch | {t_fwd * 1e3:.2f} ms | `backbone(x)` under no_grad |")
    print(f"| forward + Gram of all 12 attention inputs, one batch | {t_gram * 1e3:.2f} ms | `update_input_matrix(x)`: ratio {t_gram / t_fwd:.3f} (VERDICT r3 target <= 1.15) |")
    print(f"| `gram_mfma_kernel` alone (12 layers x [{M_rows} x 768]^T [{M_rows} x 768], one launch) | {t_k * 1e6:.0f} us | {flop / t_k / 1e12:.0f} TFLOP/s = {flop / t_k / 2.5e15:.2f} of the bf16 MFMA peak |")
    # whole hooks
    t0 = sync_time(lambda: m.before_task(0, None, batches, None))
    t_after0 = sync_time(lambda: m.after_task(0, None, batches, None))
    del batches
    batches1 = task_data(2)                    # a second task's own data (the same images again make the projected Gram rank-deficient)
    t1 = sync_time(lambda: m.before_task(1, None, batches1, None))
    try:
        t_after1 = sync_time(lambda: m.after_task(1, None, batches1, None))
    except Exception as e:                     # numpy's gesdd on the residual of synthetic data (the reference's own host math)
        t_after1 = float("nan")
        print(f"<!-- after_task(1): {type(e).__name__}: {e} -->")
    # what is host math in it: the 12 SVDs
    cur = torch.randn(768, 768)
    cur = cur @ cur.T
    from libcontinual_amd.utils import device_svd
    ts = time.perf_counter(); [torch.linalg.svd(cur, full_matrices=False) for _ in range(2)]; t_svd = (time.perf_counter() - ts) * 6
    device_svd(cur)
    ts = time.perf_counter(); [device_svd(cur) for _ in range(12)]; t_svd_dev = time.perf_counter() - ts
    print(f"| `before_task(0)`: Gram pass over {n_img} images + 12 SVDs of 768 x 768 + lora_A | {t0:.2f} s | {nb} batches x {t_gram * 1e3:.1f} ms = {nb * t_gram:.2f} s of passes; the 12 SVDs: {t_svd_dev:.2f} s on the GPU in fp64 (`utils.device_svd`, what the hooks run) against ~ {t_svd:.1f} s for the reference's `torch.linalg.svd` on this host's cores (fp32) |")
    print(f"| `after_task(0)`: merge + Gram pass + DualGPM (12-36 SVDs, fp64, on the GPU) | {t_after0:.2f} s | |")
    print(f"| `before_task(1)` (projected Gram) | {t1:.2f} s | |")
    print(f"| `after_task(1)` | {t_after1:.2f} s | |")
    print(f"\nDevice -> host traffic of a pass: 12 x 768 x 768 fp32 = 28.3 MB ONCE per hook (when `cur_matrix` is first read), not per batch "
          f"(the reference: {nb} x 28.3 MB per hook and a host-side running mean per batch).")


if __name__ == "__main__":
    main()
