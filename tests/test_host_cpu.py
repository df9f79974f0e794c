"""CPU-only tests of the host side: C-ABI library loads and exports every declared symbol (no compute calls
without a GPU), config loader semantics, registry, flat-parameter plumbing of the backbone, buffers, metrics,
schedulers, and that the product refuses to run on CPU (no fallback)."""
import copy
import ctypes
import os

import numpy as np
import pytest
import torch

import libcontinual_amd.model as M
from libcontinual_amd import _lib, optim, ops
from libcontinual_amd.config import Config
from libcontinual_amd.utils import AverageMeter, compute_bwt, compute_frgt, get_instance

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_header_symbol():
    syms = _lib.header_symbols()
    assert len(syms) >= 45
    L = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing
    assert set(_lib._PROTOS) == set(syms), set(_lib._PROTOS) ^ set(syms)
    assert _lib.lib().clhip_version() >= 100
    # no device needed for pure host queries
    assert _lib.lib().clhip_conv_fwd_tiles(256, 32, 32, 64, 64, 3, 1, 1) > 0
    assert _lib.lib().clhip_bn_bwd_ws_floats(1024, 64) > 0


def test_library_exports_nothing_outside_the_header():
    """the product's behaviour has to be a function of include/clhip.h -- every unmangled `clhip_*` export of the shared
    library is declared there (tuning and ablation hooks included: they go through clhip_config)"""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    c_abi = {e for e in exported if e.startswith("clhip_")}
    assert c_abi == set(_lib.header_symbols()), c_abi ^ set(_lib.header_symbols())


def test_clhip_config_round_trip(monkeypatch):
    L = _lib.lib()
    assert L.clhip_config(b"NO_SUCH_SWITCH", b"1") == -1 and b"unknown switch" in L.clhip_last_error()
    assert L.clhip_config(b"BN_ACC_CPT", b"4") == 0 and L.clhip_config(b"CLHIP_BN_ACC_CPT", None) == 0      # prefix accepted, NULL = default
    assert L.clhip_config(b"CONV4_FORCE_CFG", b"nonsense") == -1
    assert L.clhip_config(b"CONV4_FORCE_CFG", None) == 0
    # the getter returns what clhip_config() set (not the environment): callers that flip a switch around a region restore through it
    monkeypatch.setenv("CLHIP_BRANCH_STREAM", "1")
    assert L.clhip_config_get(b"BRANCH_STREAM") is None
    assert L.clhip_config(b"BRANCH_STREAM", b"0") == 0 and L.clhip_config_get(b"BRANCH_STREAM") == b"0" and L.clhip_config_get(b"CLHIP_BRANCH_STREAM") == b"0"
    assert L.clhip_config(b"BRANCH_STREAM", None) == 0 and L.clhip_config_get(b"BRANCH_STREAM") is None


def test_teacher_pass_restores_the_branch_stream_switch(monkeypatch):
    """TeacherPass.result() must put back what was configured (parallel.attach()'s "0"), not erase it"""
    import torch
    from libcontinual_amd import ops
    monkeypatch.delenv("CLHIP_BRANCH_STREAM", raising=False)
    L = _lib.lib()

    class FakeStream:
        def wait_stream(self, other): pass
    class FakeX:
        is_cuda = True
        class device: index = 0
    monkeypatch.setattr(ops, "_SIDE_STREAMS", {0: FakeStream()})
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: FakeStream())
    import contextlib
    monkeypatch.setattr(torch.cuda, "stream", lambda s: contextlib.nullcontext())
    seen = {}
    for prev in (b"0", None, b"2"):
        assert L.clhip_config(b"BRANCH_STREAM", prev) == 0
        tp = ops.TeacherPass(FakeX(), lambda: seen.setdefault("during", L.clhip_config_get(b"BRANCH_STREAM")) and 7)
        assert seen.pop("during") == b"0"
        assert tp.result() == 7
        assert L.clhip_config_get(b"BRANCH_STREAM") == prev
    L.clhip_config(b"BRANCH_STREAM", None)


def test_invalid_arguments_return_error_codes_not_exceptions():
    L = _lib.lib()
    rc = L.clhip_conv_fwd(None, None, None, None, 1, 4, 4, 16, 16, 3, 1, 1, 0, None)
    assert rc == -1 and b"invalid argument" in L.clhip_last_error()
    rc = L.clhip_conv_fwd(1, 1, 1, None, 1, 4, 4, 12, 16, 3, 1, 1, 0, None)      # 12 channels: not a power of two
    assert rc == -1
    with pytest.raises(_lib.ClhipError):
        _lib.call("clhip_sgd_step", None, None, None, 10, 0.1, 0.0, 0.0, 1.0, None, None, 0.0, None)


def test_no_cpu_fallback():
    bb = M.cifar_resnet32()
    with pytest.raises(_lib.ClhipError):
        bb(torch.zeros(2, 3, 32, 32))
    with pytest.raises(_lib.ClhipError):
        ops.linear(torch.zeros(2, 4), torch.zeros(3, 4), None)
    with pytest.raises(_lib.ClhipError):
        o = optim.SGD([torch.nn.Parameter(torch.zeros(4))], lr=0.1)
        o.param_groups[0]["params"][0].grad = torch.zeros(4)
        o.step()


@pytest.mark.parametrize("name,nparam,ntensors", [("cifar_resnet32", 466256, 99), ("resnet18", 11179092, 62), ("resnet32_V2", 466256, 99)])
def test_backbone_surface_matches_reference(name, nparam, ntensors):
    """names / shapes / order of parameters and buffers == the reference's, via the oracle spec"""
    from oracle import nets
    bb = getattr(M, name)(args={"dataset": "cifar100"}) if name == "resnet18" else getattr(M, name)()
    got = [(n, tuple(p.shape)) for n, p in bb.named_parameters()]
    want = [(n, tuple(s)) for n, s in nets.param_shapes(name)]
    assert sorted(got) == sorted(want)
    assert sum(p.numel() for p in bb.parameters()) == nparam and len(got) == ntensors
    assert sorted((n, tuple(b.shape)) for n, b in bb.named_buffers()) == sorted((n, tuple(s)) for n, s in nets.buffer_shapes(name))
    # reference state_dicts load unchanged; conv weights live K,R,S,C in memory
    P, Bf = nets.init_params(name), nets.init_buffers(name)
    bb.load_state_dict({**P, **Bf})
    for n, p in bb.named_parameters():
        assert torch.equal(p.detach(), P[n])
    w = dict(bb.named_parameters())[want[0][0]]
    assert w.is_contiguous(memory_format=torch.channels_last)
    assert bb._ensure_flat(torch.device("cpu")) is False          # still packed after the load
    # deepcopy (teachers) re-packs lazily and does not share storage, plans or workspaces
    t = copy.deepcopy(bb)
    assert t._ensure_flat(torch.device("cpu")) is True
    assert torch.equal(t._flat, bb._flat) and t._flat.data_ptr() != bb._flat.data_ptr()
    assert t._handle is not bb._handle and t._params[0]._clhip_owner() is t
    # .data assignment breaks the view; the next use re-packs and keeps the new value
    w.data = torch.ones_like(w)
    assert bb._ensure_flat(torch.device("cpu")) is True and float(bb._flat[: w.numel()].min()) == 1.0


def test_registry_and_method_construction():
    cfg = {"backbone": {"name": "resnet18", "kwargs": {"num_classes": 100, "args": {"dataset": "cifar100"}}},
           "classifier": {"name": "LWF", "kwargs": {"num_class": 100, "feat_dim": 512, "init_cls_num": 50, "inc_cls_num": 5}},
           "buffer": {"name": "LinearBuffer", "kwargs": {"buffer_size": 0, "batch_size": 128, "strategy": "herding"}}}
    bb = get_instance(M, "backbone", cfg)
    m = get_instance(M, "classifier", cfg, device="cpu", backbone=bb)
    for hook in ("observe", "inference", "before_task", "after_task", "get_parameters"):
        assert callable(getattr(m, hook))
    m.before_task(0, None, None, None)
    assert m.classifier.out_features == 50
    m.before_task(1, None, None, None)
    assert m.classifier.out_features == 55 and m.old_fc.out_features == 50 and m.known_cls_num == 50
    assert not any(p.requires_grad for p in m.old_backbone.parameters())
    m.train()
    assert m.old_backbone.training            # quirk a10: the trainer's model.train() un-freezes the teacher's BN
    buf = get_instance(M, "buffer", cfg)
    assert buf.buffer_size == 0 and buf.strategy == "herding" and buf.is_empty()
    e = M.EWC(M.cifar_resnet32(), 64, 100, device="cpu", init_cls_num=50, inc_cls_num=5, lamda=1000)
    e.before_task(1, None, None, None)
    assert e.network.classifier.out_features == 55
    assert set(e.fisher) == {n for n, _ in e.network.named_parameters()} and e.fisher["classifier.weight"].shape == (50, 64)
    lu = M.LUCIR(M.resnet32_V2(), 64, 100, device="cpu", init_cls_num=50, inc_cls_num=5, lamda=5, K=2, lw_mr=1, dist=0.5)
    lu.task_idx = 1
    lu.network.classifier = M.SplitCosineLinear(64, 50, 5)
    groups = lu.get_parameters({})
    assert groups[1]["lr"] == 0 and groups[0]["lr"] == 0.1 and groups[0]["weight_decay"] == 5e-4


def test_config_loader_semantics(tmp_path, monkeypatch):
    (tmp_path / "config" / "headers").mkdir(parents=True)
    (tmp_path / "config" / "headers" / "model.yaml").write_text("epoch: 7\nbatch_size: 11\n")
    (tmp_path / "my.yaml").write_text("includes:\n  - headers/model.yaml\nbatch_size: 32\noptimizer:\n  name: SGD\n  kwargs:\n    lr: 0.1\n"
                                      "    weight_decay: 5e-4\nclassifier:\n  name: EWC\n  kwargs:\n    lamda: 1000\n")
    monkeypatch.chdir(tmp_path)
    c = Config(str(tmp_path / "my.yaml")).get_config_dict()
    assert c["epoch"] == 7                       # from ./config/ include
    assert c["batch_size"] == 32                 # the file's own keys override its includes
    assert c["optimizer"]["kwargs"]["weight_decay"] == 5e-4 and isinstance(c["optimizer"]["kwargs"]["weight_decay"], float)
    assert c["testing_times"] == 10 and c["seed"] == 1993       # defaults from the packaged headers
    assert "buffer" not in c          # ./config/headers/model.yaml (cwd-relative, config.py:78-80) shadows the packaged one
    assert "includes" not in c
    # the shipped B50-5x10 configs of this repo parse and name resolvable classes
    for f in sorted(os.listdir(os.path.join(ROOT, "config"))):
        if f.endswith(".yaml"):
            monkeypatch.chdir(ROOT)
            cc = Config(os.path.join(ROOT, "config", f)).get_config_dict()
            assert hasattr(M, cc["backbone"]["name"]) and hasattr(M, cc["classifier"]["name"]) and hasattr(M, cc["buffer"]["name"])
            assert cc["init_cls_num"] + (cc["task_num"] - 1) * cc["inc_cls_num"] == cc.get("total_cls_num", 100)


def test_metrics_and_meter():
    acc = np.array([[80.0, 0, 0], [70.0, 75.0, 0], [60.0, 65.0, 90.0]])
    assert compute_bwt(acc, acc[2], 2) == pytest.approx(((60 - 80) * 2) / (2 * 3))
    assert compute_frgt(acc, acc[2], 2) == pytest.approx((80 - 60) / 2)
    m = AverageMeter("t", ["loss", "acc1"])
    m.update("loss", 2.0); m.update("loss", 4.0)
    m.update("acc1", ops.Deferred(torch.tensor(3), 100.0 / 4))          # device-resident value, resolved lazily
    m.update("acc1", torch.tensor(25.0))
    assert m.avg("loss") == 3.0 and m.avg("acc1") == pytest.approx(50.0)


def test_schedulers_match_the_reference():
    """lr sequences against the reference's own classes (tests/golden/schedulers.npz <- `python -m oracle.gen_golden schedulers`).
    The reference applies lr(0) at construction and AGAIN at the first step (scheduler.py:19-20 resets last_epoch to -1)."""
    import os
    import numpy as np
    from libcontinual_amd.scheduler import CosineAnnealingWarmUp, CosineSchedule, PatienceSchedule
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "schedulers.npz"))

    def seq(make, n):
        o = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=0.1)
        s = make(o)
        lrs = [o.param_groups[0]["lr"]]
        for _ in range(n):
            s.step()
            lrs.append(o.param_groups[0]["lr"])
        return np.asarray(lrs)
    for key, make in (("cosine_K5", lambda o: CosineSchedule(o, K=5)), ("cosine_K20", lambda o: CosineSchedule(o, K=20)), ("cosine_K1", lambda o: CosineSchedule(o, K=1)),
                      ("warmup_2_10", lambda o: CosineAnnealingWarmUp(o, 2, 10)), ("warmup_3_30", lambda o: CosineAnnealingWarmUp(o, 3, 30))):
        np.testing.assert_allclose(seq(make, len(ref[key]) - 1), ref[key], rtol=1e-12, atol=1e-15, err_msg=key)
    assert ref["cosine_K5"][0] == ref["cosine_K5"][1]                      # the repeated first value
    o = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=0.1)
    s = PatienceSchedule(o, patience=2, factor=2)
    lrs = []
    for v in (1.0, 1.0, 1.0, 0.5, 0.6, 0.7, 0.8):
        s.step(v)
        lrs.append(o.param_groups[0]["lr"])
    np.testing.assert_allclose(lrs, ref["patience_2_2"], rtol=1e-12)


def test_herding_buffer_bookkeeping():
    b = M.LinearHerdingBuffer(20, 64)
    b.add_data([f"a{i}" for i in range(10)] + [f"b{i}" for i in range(10)], [0] * 10 + [1] * 10)
    b.reduce_old_data(1, 4)              # 20 // 4 = 5 per class, first-k kept
    assert b.labels == [0] * 5 + [1] * 5 and b.images[:2] == ["a0", "a1"] and b.images[5] == "b0"


def test_bic_split_and_buffer_recut_match_the_reference(golden):
    """`bic.spilt_and_update` (bic.py:245-340) runs on the host: 9:1 class-wise split under the global numpy RNG, the two loaders'
    datasets and the re-cut split buffer, against what the reference's own method produced (tests/golden/bic.npz)"""
    import numpy as np
    import libcontinual_amd.model as M
    from oracle import scenarios as sc
    want = golden("bic")
    c = sc.BIC_CFG
    m = M.bic(M.cifar_resnet32_V2(), c["num_class"], device="cpu", task_num=c["task_num"], init_cls_num=c["init"], inc_cls_num=c["inc"])
    assert m.model.classifier.in_features == 256 and len(m.bias_layers) == c["task_num"]
    split = sc.bic_split_plugin(m, M)
    for t, rec in split.items():
        for key, v in zip(sc.BIC_SPLIT_FIELDS, rec):
            if v is not None:
                np.testing.assert_array_equal(np.asarray([int(q) for q in v]), want[f"split{t}_{key}"], err_msg=f"{t}/{key}")
    # bias_forward: every task's slice through its own (alpha, beta)
    import torch
    with torch.no_grad():
        for i, layer in enumerate(m.bias_layers):
            layer.alpha.fill_(1.0 + i); layer.beta.fill_(0.1 * i)
    z = torch.arange(2 * c["num_class"], dtype=torch.float32).reshape(2, c["num_class"])
    out = m.bias_forward(z)
    want_out = torch.cat([(1.0 + i) * z[:, 3 * i:3 * i + 3] + 0.1 * i for i in range(3)], 1)
    assert torch.allclose(out, want_out)


def test_input_pipeline_tables_equal_the_reference_fixture(golden):
    """tests/golden/augment.npz carries the parameter tables parsed out of the reference's core/data/data.py:4-35 (oracle/gen_augment_golden.py);
    the product's transform tables are those numbers, and its CPU transforms reproduce the fixture's test-time outputs exactly"""
    from libcontinual_amd.data import transforms as T
    fx = golden("augment")
    assert np.allclose(fx["mean"], T.CIFAR_MEAN) and np.allclose(fx["std"], T.CIFAR_STD)
    tr = T.cifar_resnet_transform("train", 32)
    kinds = {type(t).__name__: t for t in tr.transforms}
    assert kinds["RandomCrop"].padding == int(fx["padding"][0]) and abs(kinds["ColorJitter"].b - float(fx["brightness"][0])) < 1e-12
    te = T.cifar_resnet_transform("test", 32)
    for k in range(len(fx["cifar_test_expected"])):
        assert np.abs(te(fx["cifar_images"][k]).numpy() - fx["cifar_test_expected"][k]).max() <= 1e-6


def test_resample_oracle_matches_pillow_and_the_fixture(golden):
    """oracle/resample.py restates Pillow's anti-aliased bilinear resize (the arithmetic behind torchvision's RandomResizedCrop on the
    PIL images of the reference's ImageNet-R pipeline).  Pinned: bit-exact against Pillow itself on random sizes (shrinking, enlarging,
    mixed, one-pixel outputs) and against tests/golden/augment_aa.npz (boxes + expected outputs, oracle/gen_augment_aa_golden.py)."""
    from PIL import Image
    from oracle.resample import resize_bilinear, resized_crop
    rs = np.random.RandomState(0)
    for t in range(14):
        H, W = (int(v) for v in rs.randint(2, 180, 2))
        oh, ow = ((48, 48), (1, 7), (int(rs.randint(1, 90)), int(rs.randint(1, 90))))[t % 3]
        img = (rs.rand(H, W, 3) * 255).astype(np.uint8)
        want = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
        assert np.array_equal(resize_bilinear(img, oh, ow), want), (H, W, oh, ow)
    fx = golden("augment_aa")
    for j in (0, 1, 3, 7):                                          # the small outputs (the 224 x 224 case runs on the GPU test)
        top, left, h, w, fl = (int(v) for v in fx["boxes"][j])
        S = int(fx["out_sizes"][j])
        got = resized_crop(fx[f"image_{int(fx['which'][j])}"], top, left, h, w, S, S)
        assert np.array_equal(got[:, ::-1] if fl else got, fx[f"train_expected_u8_{j}"]), j


def test_ragged_store_preload_and_deterministic_head(tmp_path, golden):
    """`preload: true` on a class-folder tree of differently sized images (ImageNet-R): the train split becomes a RaggedStore whose
    views are the decoded images; the test pipeline's Resize(256, BICUBIC) + CenterCrop(224) head is applied once at load (equal to
    the fixture made with the reference's YAML parameters) and only ToTensor stays per batch; the box sampler stays inside every image."""
    from PIL import Image
    from libcontinual_amd.data import transforms as T
    from libcontinual_amd.data.dataset import RaggedStore, get_dataloader, store_hw
    from libcontinual_amd.data.gpu_loader import _rrc_boxes, gpu_plan
    fx = golden("augment_aa")
    n_img = len(fx["hw"])
    for mode in ("train", "test"):
        for k in range(n_img):
            d = tmp_path / mode / f"c{k % 2}"
            d.mkdir(parents=True, exist_ok=True)
            Image.fromarray(fx[f"image_{k}"]).save(d / f"{k}.png")
    S = int(fx["size"][0])
    cfg = dict(data_root=str(tmp_path), dataset="imagenet-r", preload=True, batch_size=4, num_workers=0, task_num=2, init_cls_num=1, inc_cls_num=1,
               class_order=[0, 1],
               train_trfms=[{"RandomResizedCrop": {"size": S, "scale": fx["scale"].tolist(), "ratio": fx["ratio"].tolist()}},
                            {"RandomHorizontalFlip": {"p": float(fx["flip_p"][0])}}, {"ToTensor": {}}],
               test_trfms=[{"Resize": {"size": 256, "interpolation": "BICUBIC"}}, {"CenterCrop": {"size": S}}, {"ToTensor": {}}])
    train = get_dataloader(cfg, "train")
    ds = train.get_loader(0).dataset
    assert isinstance(ds.store, RaggedStore) and store_hw(ds.store) == (None, True) and len(ds.store) == n_img
    order = [k for c in range(2) for k in range(n_img) if k % 2 == c]                   # class folders in label order, files sorted
    for pos, k in enumerate(order):
        assert np.array_equal(ds.store[pos], fx[f"image_{k}"])
    assert gpu_plan(ds.trfms, *store_hw(ds.store))["kind"] == "rrc_aa" and gpu_plan(ds.trfms, (32, 32))["kind"] == "rrc_flip"
    assert gpu_plan(ds.trfms, (500, 375))["kind"] == "rrc_aa" and gpu_plan(T.cifar_resnet_transform("train", 32), None, True) is None
    torch.manual_seed(0)
    assert ds[1]["image"].shape == (3, S, S)                                            # the per-sample CPU path works on the views
    test = get_dataloader(cfg, "test", cls_map=train.cls_map)
    tds = test.get_loader(1)[1].dataset
    assert store_hw(tds.store) == ((S, S), False) and [type(t).__name__ for t in tds.trfms.transforms] == ["ToTensor"]
    assert gpu_plan(tds.trfms, *store_hw(tds.store))["kind"] == "crop_flip"
    pos = order.index(1)
    assert np.array_equal(tds.store[pos], fx["test_expected_u8_1"])
    # vectorised torchvision get_params: inside the image, aspect within `ratio` up to rounding, centre-crop fallback for extreme shapes
    torch.manual_seed(1)
    Hs, Ws = torch.randint(40, 700, (4000,)), torch.randint(40, 700, (4000,))
    Hs[:8], Ws[:8] = 30, 900
    y0, x0, h, w = _rrc_boxes(Hs, Ws, (0.05, 1.0), (0.75, 1.333)).long().unbind(1)
    assert bool(((h > 0) & (w > 0) & (y0 >= 0) & (x0 >= 0) & (y0 + h <= Hs) & (x0 + w <= Ws)).all())
    assert h[:8].tolist() == [30] * 8 and w[:8].tolist() == [40] * 8 and x0[:8].tolist() == [430] * 8
    frac = (h * w).double() / (Hs * Ws).double()
    assert 0.04 < float(frac[8:].min()) and float(frac[8:].max()) <= 1.0 and 0.25 < float(frac[8:].mean()) < 0.55      # first valid of 10 tries: large boxes are rejected more often on elongated images


def test_hw_queue_cap_state_is_reported():
    """the package sets GPU_MAX_HW_QUEUES (3; 4 for a rank of a multi-process job) for processes that import it before the HIP runtime starts and reports what it found
    (hw_queue_cap_state: "ok" it set the cap in time, "user" the caller's own value wins, "late" the runtime was already up -- VERDICT r4 item 7a)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "import libcontinual_amd, os; print(libcontinual_amd.hw_queue_cap_state(), os.environ['GPU_MAX_HW_QUEUES'])"
    env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "WORLD_SIZE")}
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, check=True).stdout
    assert "('ok', '3') 3" in out
    # a rank of a multi-process job (torch.distributed.run sets WORLD_SIZE): RCCL's streams need a fourth queue (round 5, tools/dp_step_micro.py)
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(env, WORLD_SIZE="8"), capture_output=True, text=True, check=True).stdout
    assert "('ok', '4') 4" in out
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(env, WORLD_SIZE="1"), capture_output=True, text=True, check=True).stdout
    assert "('ok', '3') 3" in out
    env["GPU_MAX_HW_QUEUES"] = "2"
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, check=True).stdout
    assert "('user', '2') 2" in out


def test_data_parallel_graph_replay_is_only_legal_on_rccl(monkeypatch):
    """trainer._graph_mode: a reduced step may be captured only when its exchange is made of capturable launches -- the RCCL backend with the in-place
    all-reduce; gloo (host staging), the sharded exchange and CLHIP_DP_GRAPH=0 keep the loop eager (round 5)"""
    from libcontinual_amd import trainer as T

    class Red:
        def __init__(self, exchange):
            self.exchange, self.group, self.world = exchange, None, 2

    monkeypatch.delenv("CLHIP_DP_GRAPH", raising=False)
    assert T._reducer_capturable(None)
    assert not T._reducer_capturable(Red("all_reduce"))            # no process group in this process: not the RCCL backend
    assert not T._reducer_capturable(Red("reduce_scatter"))
    monkeypatch.setenv("CLHIP_DP_GRAPH", "0")
    assert not T._reducer_capturable(Red("all_reduce"))

    class M:
        cuda_graph_safe, grad_reducer = True, None

    class O:
        capture_safe = True
    assert T._graph_mode(M(), Red("all_reduce"), "cuda", O()) is None
    monkeypatch.delenv("CLHIP_DP_GRAPH", raising=False)
    assert T._graph_mode(M(), None, "cuda", O()) == "auto"
    assert T._graph_mode(M(), None, "cpu", O()) is None
