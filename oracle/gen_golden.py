"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

    python -m oracle.gen_golden            # from the repo root; needs /root/reference

The reference's Python is imported in place through `oracle/ref_shim.py`; nothing of it is copied.
Fixtures hold expected OUTPUTS only (inputs/weights are rebuilt from tags, see oracle/detrand.py).
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

from . import fixtures, ref_shim, scenarios, trainer_scenarios, vit_scenarios


def reference_namespace():
    resnet = ref_shim.load("core.model.backbone.resnet")
    ns = types.SimpleNamespace()
    for n in ("cifar_resnet32", "cifar_resnet32_V2", "resnet18", "resnet32_V2", "CosineLinear", "SplitCosineLinear"):
        setattr(ns, n, getattr(resnet, n))
    ns.EWC = ref_shim.load("core.model.ewc").EWC
    ns.LWF = ref_shim.load("core.model.lwf").LWF
    ns.ICarl = ref_shim.load("core.model.icarl").ICarl
    ns.LUCIR = ref_shim.load("core.model.lucir").LUCIR
    ns.Finetune = ref_shim.load("core.model.finetune").Finetune
    ns.WA = ref_shim.load("core.model.wa").WA
    # der.py:26-27 imports its extractor factories from the backbone PACKAGE and `get_instance` (unused) from core.utils;
    # the shim's packages are empty, so hand it the reference's own objects
    for n in ("resnet18", "resnet34", "resnet50"):
        setattr(sys.modules["core.model.backbone"], n, getattr(resnet, n))
    utils = types.ModuleType("core.utils")
    utils.get_instance = None
    sys.modules.setdefault("core.utils", utils)
    ns.DER = ref_shim.load("core.model.der").DER
    ns.bic = ref_shim.load("core.model.bic").bic
    ns.LinearSpiltBuffer = ref_shim.load("core.model.buffer.linearbuffer").LinearSpiltBuffer
    ns.LinearHerdingBuffer = ref_shim.load("core.model.buffer.linearherdingbuffer").LinearHerdingBuffer
    return ns


def reference_vit_namespace():
    """the reference's ViT classes at the fixture configuration (ViTZoo itself hard-codes ViT-B/16, vit.py:47-51:
    the same object graph is assembled around a small VisionTransformer)"""
    os.environ.setdefault("PYTHONHASHSEED", "0")           # read by InfLoRA_opt.py:55,218
    ref_shim.install_vit_standins()
    tr = ref_shim.load("core.model.backbone.transformer")
    vit = ref_shim.load("core.model.backbone.vit")
    ns = types.SimpleNamespace()

    def make_vit(cfg, attn_layer="MultiHeadAttention", lora_rank=0):
        zoo = vit.ViTZoo.__new__(vit.ViTZoo)
        torch.nn.Module.__init__(zoo)
        kw = {"lora_rank": lora_rank} if lora_rank else {}
        zoo.task_id, zoo.feat_dim = None, cfg["dim"]
        zoo.feat = tr.VisionTransformer(img_size=cfg["img"], patch_size=cfg["patch"], embed_dim=cfg["dim"], depth=cfg["depth"],
                                        num_heads=cfg["heads"], ckpt_layer=0, drop_path_rate=0, attn_layer=attn_layer, **kw)
        zoo.prompt, zoo.prompt_flag = None, ""
        return zoo
    ns.make_vit = make_vit

    def make_sinet(cfg, total_sessions, rank, init_cls):
        """SiNet_vit.__init__ downloads vit_base_patch16_224_in21k (SiNet.py:72-73): the same object graph around a small
        ViT_lora_co"""
        sn = ref_shim.load("core.model.backbone.SiNet")
        net = sn.SiNet_vit.__new__(sn.SiNet_vit)
        torch.nn.Module.__init__(net)
        net.image_encoder = sn.ViT_lora_co(img_size=cfg["img"], patch_size=cfg["patch"], embed_dim=cfg["dim"], depth=cfg["depth"],
                                           num_heads=cfg["heads"], n_tasks=total_sessions, rank=rank)
        net.class_num = init_cls
        net.classifier_pool = torch.nn.ModuleList([torch.nn.Linear(cfg["dim"], init_cls, bias=True) for _ in range(total_sessions)])
        net.classifier_pool_backup = torch.nn.ModuleList([torch.nn.Linear(cfg["dim"], init_cls, bias=True) for _ in range(total_sessions)])
        net.numtask = 0
        return net
    ns.make_sinet = make_sinet
    inflora_mod = ref_shim.load("core.model.InfLoRA")
    # InfLoRA.py hands torch tensors to np.linalg.svd.  Under the numpy 1.x it was written for, linalg results come back as
    # ndarrays (`_makearray` looks for `__array_prepare__`); numpy 2.x looks for `__array_wrap__`, which torch defines, returns
    # Tensors and the method's next line (`feature_list[p].transpose()`, InfLoRA.py:207) raises.  Give the module numpy-1.x
    # behaviour: a `np` whose linalg.svd sees ndarrays.
    real_np = inflora_mod.np
    proxy_linalg = types.SimpleNamespace(**{k: getattr(real_np.linalg, k) for k in dir(real_np.linalg) if not k.startswith("_")})
    proxy_linalg.svd = lambda a, *args, **kw: real_np.linalg.svd(real_np.asarray(a), *args, **kw)

    class _Np1:
        linalg = proxy_linalg

        def __getattr__(self, k):
            return getattr(real_np, k)
    inflora_mod.np = _Np1()
    ns.InfLoRA = inflora_mod.InfLoRA
    ns.L2P = ref_shim.load("core.model.l2p").L2P
    ns.InfLoRA_OPT = ref_shim.load("core.model.InfLoRA_opt").InfLoRA_OPT
    return ns


def main(out_dir=None):
    torch.set_num_threads(8)
    out_dir = out_dir or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    ad = scenarios.PluginAdapter(reference_namespace(), "cpu")
    jobs = {}
    for arch in ("cifar_resnet32", "resnet32_V2", "resnet18", "cifar_resnet32_V2"):
        jobs[f"backbone_{arch}"] = lambda a=arch: scenarios.scenario_backbone(ad, a)
    jobs["ewc"] = lambda: scenarios.scenario_ewc(ad)
    jobs["ewc_fisher"] = lambda: scenarios.scenario_ewc_fisher(ad)
    jobs["lwf_resnet18"] = lambda: scenarios.scenario_lwf(ad)
    jobs["lwf_cifar_resnet32"] = lambda: scenarios.scenario_lwf(ad, dict(arch="cifar_resnet32", feat_dim=64, bs=8))
    jobs["lwf_long"] = lambda: scenarios.scenario_lwf(ad, scenarios.LWF_LONG_CFG)
    jobs["lucir"] = lambda: scenarios.scenario_lucir(ad)
    jobs["wa"] = lambda: scenarios.scenario_wa(ad)
    jobs["der"] = lambda: scenarios.scenario_der(ad)
    jobs["bic"] = lambda: scenarios.scenario_bic(ad)

    def icarl():
        with tempfile.TemporaryDirectory() as d:
            return scenarios.scenario_icarl(ad, d)
    jobs["icarl"] = icarl
    vad = [None]

    def vit_job(fn):
        def run():
            if vad[0] is None:
                vad[0] = vit_scenarios.VitPluginAdapter(reference_vit_namespace(), "cpu")
            return fn(vad[0])
        return run
    jobs["vit_backbone"] = vit_job(vit_scenarios.scenario_vit_backbone)
    jobs["l2p"] = vit_job(vit_scenarios.scenario_l2p)
    jobs["inflora"] = vit_job(vit_scenarios.scenario_inflora)
    jobs["inflora_orig"] = vit_job(vit_scenarios.scenario_inflora_orig)
    # ViT-B/16 at its full geometry through the reference's own classes (batch 2, one forward + backward of the LoRA branch, one L2P step): summaries only
    jobs["vit_b16_full"] = vit_job(vit_scenarios.scenario_vit_full)
    def schedulers():
        """learning-rate sequences of the reference's own scheduler classes (core/scheduler.py:47-124): value at construction,
        then after every `step()`"""
        sc = ref_shim.load("core.scheduler")

        def seq(make, n):
            o = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=0.1)
            s_ = make(o)
            lrs = [o.param_groups[0]["lr"]]
            for _ in range(n):
                s_.step()
                lrs.append(o.param_groups[0]["lr"])
            return np.asarray(lrs, np.float64)
        res = {"cosine_K5": seq(lambda o: sc.CosineSchedule(o, K=5), 6), "cosine_K20": seq(lambda o: sc.CosineSchedule(o, K=20), 21),
               "cosine_K1": seq(lambda o: sc.CosineSchedule(o, K=1), 3), "warmup_2_10": seq(lambda o: sc.CosineAnnealingWarmUp(o, 2, 10), 11),
               "warmup_3_30": seq(lambda o: sc.CosineAnnealingWarmUp(o, 3, 30), 12)}
        o = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=0.1)
        s_ = sc.PatienceSchedule(o, patience=2, factor=2)
        lrs = []
        for v in (1.0, 1.0, 1.0, 0.5, 0.6, 0.7, 0.8):
            s_.step(v)
            lrs.append(o.param_groups[0]["lr"])
        res["patience_2_2"] = np.asarray(lrs, np.float64)
        return res
    jobs["schedulers"] = schedulers
    # whole runs of the reference's own Trainer.train_loop, in its own fp32 arithmetic (oracle/trainer_scenarios.py)
    fp32_jobs = set()
    for tn in trainer_scenarios.SCENARIOS:
        jobs[f"trainer_{tn}"] = lambda n=tn: trainer_scenarios.scenario_trainer(n, reference_namespace())
        fp32_jobs.add(f"trainer_{tn}")
    only = sys.argv[1:]
    for name, fn in jobs.items():
        if only and name not in only:
            continue
        torch.manual_seed(0)
        # fp64 run of the reference = the semantic ground truth (see fixtures.use_dtype)
        with fixtures.use_dtype(torch.float32 if name in fp32_jobs else torch.float64):
            res = fn()
        path = os.path.join(out_dir, name + ".npz")
        np.savez_compressed(path, **res)
        print(f"{name}: {os.path.getsize(path)} bytes", {k: getattr(v, 'shape', None) for k, v in list(res.items())[:4]})


if __name__ == "__main__":
    main()
