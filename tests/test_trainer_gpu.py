"""End-to-end on the MI355X: the product Trainer + data module + plugins + buffers on a synthetic class-conditional
dataset (no CIFAR-100 on the box): every in-scope method trains through 3 tasks in both compute modes and learns
(accuracy far above chance on the first task; rehearsal methods stay above chance on all 10 classes)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from libcontinual_amd.config import Config     # noqa: E402
from libcontinual_amd.trainer import Trainer   # noqa: E402


def cfg_for(method, backbone, dtype, **over):
    cfg = Config().get_config_dict()
    feat = 512 if backbone == "resnet18" else 64
    kw = {"num_class": 10, "feat_dim": feat, "init_cls_num": 4, "inc_cls_num": 3}
    if method == "EWC":
        kw["lamda"] = 10
    if method == "ICarl":
        kw["task_num"] = 3
    if method == "LUCIR":
        kw.update(lamda=5, K=2, lw_mr=1, dist=0.5)
    if method == "WA":                                # wa.py:222 grows the head by init_cls_num per task: only equal splits make sense
        kw.update(init_cls_num=3)
    cfg.update(dict(dataset="synthetic", image_size=32, init_cls_num=4, inc_cls_num=3, task_num=3, epoch=5, init_epoch=8, batch_size=64,
                    val_per_epoch=10, testing_times=1, num_workers=0, save_path="", synthetic_per_class=200, synthetic_test_per_class=20,
                    seed=7, backbone={"name": backbone, "kwargs": {"num_classes": 10, "dtype": dtype, "args": {"dataset": "cifar100"}}},
                    classifier={"name": method, "kwargs": kw},
                    optimizer={"name": "SGD", "kwargs": {"lr": 0.02, "momentum": 0.9, "weight_decay": 5e-4}},
                    lr_scheduler={"name": "MultiStepLR", "kwargs": {"milestones": [3, 6], "gamma": 0.2}}))
    if method == "WA":
        cfg.update(init_cls_num=3)
    cfg.update(over)
    return cfg


@pytest.mark.parametrize("method,backbone,extra", [
    ("EWC", "cifar_resnet32", {}),
    ("LWF", "resnet18", {}),
    ("ICarl", "cifar_resnet32", {"buffer": {"name": "LinearHerdingBuffer", "kwargs": {"buffer_size": 60, "batch_size": 32}}}),
    ("LUCIR", "resnet32_V2", {"buffer": {"name": "LinearBuffer", "kwargs": {"buffer_size": 60, "batch_size": 32, "strategy": "herding"}}}),
    ("WA", "cifar_resnet32", {"buffer": {"name": "LinearHerdingBuffer", "kwargs": {"buffer_size": 60, "batch_size": 32}}}),
    ("DER", "resnet18", {"buffer": {"name": "LinearBuffer", "kwargs": {"buffer_size": 60, "batch_size": 32, "strategy": "random"}}}),
])
def test_methods_train_end_to_end(method, backbone, extra, monkeypatch):
    res = {}
    for dtype in ("bf16", "f32"):
        monkeypatch.setenv("CLHIP_DTYPE", dtype)                 # DER builds its extractors itself: their dtype comes from the env
        tr = Trainer(0, cfg_for(method, backbone, dtype, **extra), log=lambda *a, **k: None)
        out = tr.train_loop()
        res[dtype] = out
        acc = out["acc_table"]
        assert np.isfinite(acc).all()
        # first task learned (chance 25-33 %): 96-100 % observed; LUCIR's cosine head on this 13-epoch run lands at 86-100 %
        # (the fp32-atomic summation order of the stride-2 / 1x1 weight gradients differs from run to run)
        assert acc[0, 0] > (75.0 if method == "LUCIR" else 90.0), (method, dtype, acc)
        if method in ("ICarl", "LUCIR", "WA", "DER"):             # rehearsal methods fill their buffer
            assert len(tr.buffer.labels) > 0
        if method == "ICarl":                                     # NCM over the herded exemplars keeps the old classes alive
            assert out["batch_last_acc"] > 80.0, (method, dtype, acc)   # 10 classes at the end (92-100 % observed; chance 10 %)
        if method == "DER":                                       # one extractor per task, all but the last frozen
            assert out["batch_last_acc"] > 60.0, (method, dtype, acc)   # frozen extractors keep the old tasks (71-80 % observed)
            assert len(tr.model.convnets) == 3 and tr.model.fc.in_features == 3 * 512
            assert not any(q.requires_grad for q in tr.model.convnets[0].parameters())
        if method == "WA":
            assert out["batch_last_acc"] > 45.0, (method, dtype, acc)   # 9 classes (55-57 % observed: the untrained-head quirk caps it)
            assert tr.model.network.classifier.out_features == 9 and tr.model.old_network is not None
        torch.cuda.synchronize()


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_bic_trains_end_to_end(dtype):
    """BiC through the Trainer: its own SGD recipe and train / validation split (core/trainer.py:297-303), the pre-activation
    backbone on 64 x 64 images, stage 2 on the validation split after every task but the first (:420-455)"""
    cfg = Config().get_config_dict()
    kw = {"num_class": 9, "init_cls_num": 3, "inc_cls_num": 3, "task_num": 3}
    cfg.update(dict(dataset="synthetic", image_size=64, init_cls_num=3, inc_cls_num=3, task_num=3, epoch=8, init_epoch=8, stage2_epoch=3, batch_size=32,
                    val_per_epoch=10, testing_times=1, num_workers=0, save_path="", synthetic_per_class=200, synthetic_test_per_class=20, seed=5,
                    backbone={"name": "cifar_resnet32_V2", "kwargs": {"num_classes": 9, "dtype": dtype, "args": {"dataset": "synthetic"}}},
                    classifier={"name": "bic", "kwargs": kw},
                    buffer={"name": "LinearSpiltBuffer", "kwargs": {"buffer_size": 90, "batch_size": 32, "strategy": "balance_random", "val_ratio": 0.1}},
                    optimizer={"name": "SGD", "kwargs": {"lr": 0.1, "momentum": 0.9, "weight_decay": 2e-4}},
                    lr_scheduler={"name": "MultiStepLR", "kwargs": {"milestones": [100, 150, 200], "gamma": 0.1}}))
    tr = Trainer(0, cfg, log=lambda *a, **k: None)
    out = tr.train_loop()
    acc = out["acc_table"]
    assert np.isfinite(acc).all()
    assert acc[0, 0] > 90.0, acc
    # 9 classes, 90 rehearsal exemplars, 20 test images per class; chance 11 %.  Five runs of the same build on one box gave 80, 80, 52,
    # 44 and 78 % (the stage-2 bias layers of an 8-epoch run sit on a knife edge and the atomic weight-gradient kernels of the 16- and
    # 32-channel layers sum in arrival order), so the bound is "well above chance", not a level
    assert out["batch_last_acc"] > 30.0, acc
    m = tr.model
    assert m.model.classifier.in_features == 256 and m.seen_cls == 9
    ab = [(layer.alpha.item(), layer.beta.item()) for layer in m.bias_layers]
    assert ab[0] == (1.0, 0.0) and ab[1] != (1.0, 0.0) and ab[2] != (1.0, 0.0), ab      # stage 2 trained the layers of tasks 1 and 2 only
    assert tr.optimizer.param_groups[0]["weight_decay"] == pytest.approx(2e-4 * 3 / 3) and tr.optimizer.param_groups[0]["momentum"] == 0.9
    assert [e for e in tr.hook_trace if e[0] == "stage2_epoch"] == [("stage2_epoch", t, e) for t in (1, 2) for e in range(3)]
    assert 0 < len(tr.buffer.train_labels) + len(tr.buffer.val_labels) <= 90 and tr.buffer.total_classes == 9
    assert len(tr.buffer.val_labels) >= 9                         # at least one validation exemplar per class
    torch.cuda.synchronize()


def vit_cfg(method, dtype):
    cfg = Config().get_config_dict()
    bb_kw = {"pretrained": False, "img_size": 32, "patch_size": 8, "embed_dim": 128, "depth": 2, "num_heads": 2, "dtype": dtype}
    if method == "L2P":
        kw = {"init_cls_num": 4, "inc_cls_num": 3, "num_class": 10, "task_num": 3, "feat_dim": 128, "prompt_length": 2, "pool_size": 6, "top_k": 3,
              "pull_constraint_coeff": 1.0}
        opt = {"name": "Adam", "kwargs": {"lr": 0.01, "betas": [0.9, 0.999], "weight_decay": 0}}
    else:
        bb_kw.update(attn_layer="MultiHeadAttention_LoRA", lora_rank=4)
        kw = {"use_ca": False, "dataset": "synthetic", "init_cls_num": 4, "inc_cls_num": 3, "task_num": 3, "lame": 0.9, "lamb": 0.6, "embd_dim": 128}
        opt = {"name": "SGD", "kwargs": {"lr": 0.05, "momentum": 0.9}}
    cfg.update(dict(dataset="synthetic", image_size=32, init_cls_num=4, inc_cls_num=3, task_num=3, epoch=4, init_epoch=6, batch_size=32,
                    val_per_epoch=10, testing_times=1, num_workers=0, save_path="", synthetic_per_class=96, synthetic_test_per_class=16, seed=5,
                    backbone={"name": "vit_pt_imnet", "kwargs": bb_kw}, classifier={"name": method, "kwargs": kw}, optimizer=opt,
                    lr_scheduler={"name": "Constant"}))
    return cfg


def inflora_orig_cfg(dtype):
    cfg = vit_cfg("InfLoRA_OPT", dtype)
    cfg.update(init_cls_num=3, inc_cls_num=3, task_num=3,
               backbone={"name": "SiNet_vit", "kwargs": {"total_sessions": 3, "rank": 4, "init_cls": 3, "embd_dim": 128, "img_size": 32, "patch_size": 8,
                                                         "depth": 2, "num_heads": 2, "dtype": dtype}},
               classifier={"name": "InfLoRA", "kwargs": {"feat_dim": 64, "num_class": 9, "inc_cls_num": 3, "lame": 0.9, "lamb": 0.6, "total_sessions": 3,
                                                         "gram_size": 32}})
    return cfg


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_inflora_original_trains_end_to_end(dtype):
    """the multi-branch InfLoRA through the product Trainer: three tasks, one LoRA pair and one head per task"""
    tr = Trainer(0, inflora_orig_cfg(dtype), log=lambda *a, **k: None)
    out = tr.train_loop()
    acc = out["acc_table"]
    assert np.isfinite(acc).all()
    assert acc[0, 0] > 50.0, (dtype, acc)                         # 3 classes: chance = 33 %
    net = tr.model._network
    assert net.numtask == 3 and len(tr.model.feature_list) == 2
    a0 = net.image_encoder.blocks[0].attn
    assert all(float(a0.lora_B_k[t].weight.detach().abs().max()) > 0 for t in range(3))          # every task trained ITS pair
    trainable = [n for n, q in net.named_parameters() if q.requires_grad]
    assert trainable and all(".2." in n for n in trainable)       # ... and only the last task's pair + head is still trainable
    torch.cuda.synchronize()


@pytest.mark.parametrize("method", ["L2P", "InfLoRA_OPT"])
@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_vit_methods_train_end_to_end(method, dtype):
    """the ViT plugins through the product Trainer (L2P does backward + clip inside observe; InfLoRA_OPT runs its Gram / SVD /
    DualGPM hooks around every task) on a small random-init ViT: finite, task 0 learned above chance"""
    import os
    os.environ.setdefault("PYTHONHASHSEED", "0")
    tr = Trainer(0, vit_cfg(method, dtype), log=lambda *a, **k: None)
    out = tr.train_loop()
    acc = out["acc_table"]
    assert np.isfinite(acc).all()
    assert acc[0, 0] > 40.0, (method, dtype, acc)                 # 4 classes: chance = 25 %
    if method == "InfLoRA_OPT":
        assert len(tr.model.feature_list) == 2 and all(a.apply_lora is False for a in tr.model.attention_modules)
    torch.cuda.synchronize()


@pytest.mark.parametrize("method,extra", [
    ("EWC", {}),
    ("ICarl", {"buffer": {"name": "LinearHerdingBuffer", "kwargs": {"buffer_size": 60, "batch_size": 32}}}),
])
def test_the_trainers_own_batch_32_epochs_replay_from_a_graph(method, extra, recwarn):
    """the step bench.py replays at batch 32 is what the Trainer itself runs -- its EWC and iCaRL epochs at 32 images per step
    (CifarResNet-32, bf16, fused SGD) are captured and replayed, task >= 1 included (penalty term / frozen teacher on its side stream), with no
    "could not be captured" warning; the aborted captures of tests/test_trainer_trace_gpu.py come from that test's host-reading observe() wrapper"""
    tr = Trainer(0, cfg_for(method, "cifar_resnet32", "bf16", batch_size=32, epoch=3, init_epoch=3, task_num=2, **extra), log=lambda *a, **k: None)
    out = tr.train_loop()
    assert np.isfinite(out["acc_table"]).all()
    gs = getattr(tr.model, "_graphed_step", None)
    assert gs is not None and not gs.disabled, "the Trainer's step loop never built a GraphedStep / gave the capture up"
    assert len(gs.graphs) >= 1 and (any(g is not None for g in gs.graphs.values()) or "eager" in gs.choice.values())
    bad = [str(w.message) for w in recwarn.list if "could not be captured" in str(w.message)]
    assert not bad, bad
    torch.cuda.synchronize()
