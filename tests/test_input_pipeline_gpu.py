"""GPU input pipeline (csrc/augment.hip + data/gpu_loader.py): the augment kernels vs the CPU transforms on the same images
with the SAME random parameters (exact for crop / flip / brightness / normalise; bilinear RandomResizedCrop vs PIL within one
uint8 step), and the loader contract (coverage of an epoch, sharding, drop-in batch dicts)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from libcontinual_amd._lib import call                         # noqa: E402
from libcontinual_amd.data import ArrayDataset, make_loader    # noqa: E402
from libcontinual_amd.data import transforms as T              # noqa: E402
from libcontinual_amd.data.gpu_loader import GpuBatchLoader, gpu_plan   # noqa: E402

DEV = "cuda"


def st():
    return torch.cuda.current_stream().cuda_stream


def store_of(n, size=32, seed=0):
    return (np.random.RandomState(seed).rand(n, size, size, 3) * 255).astype(np.uint8)


def test_crop_flip_kernel_matches_cpu_transforms():
    store = store_of(20)
    B, S, pad = 12, 32, 4
    g = torch.Generator().manual_seed(1)
    idx = torch.randint(0, 20, (B,), generator=g)
    params = torch.stack([torch.randint(0, 2 * pad + 1, (B,), generator=g), torch.randint(0, 2 * pad + 1, (B,), generator=g),
                          torch.randint(0, 2, (B,), generator=g)], 1).int()
    bright = torch.empty(B).uniform_(0.75, 1.25, generator=g)
    out = torch.empty(B, 3, S, S, device=DEV)
    mean, std = (C.c_float * 3)(*T.CIFAR_MEAN), (C.c_float * 3)(*T.CIFAR_STD)
    sd, idx_d, par_d, bri_d = torch.as_tensor(store).to(DEV), idx.to(DEV), params.to(DEV), bright.to(DEV)     # keep the device copies alive
    call("clhip_augment_crop_flip", sd.data_ptr(), idx_d.data_ptr(), par_d.data_ptr(), bri_d.data_ptr(), out.data_ptr(),
         B, 32, 32, S, pad, mean, std, st())
    torch.cuda.synchronize()
    norm = T.Normalize(T.CIFAR_MEAN, T.CIFAR_STD)
    for b in range(B):
        a = np.pad(store[int(idx[b])], ((pad, pad), (pad, pad), (0, 0)))
        dy, dx, flip = (int(v) for v in params[b])
        a = a[dy:dy + S, dx:dx + S]
        if flip:
            a = a[:, ::-1]
        a = np.clip(a.astype(np.float32) * float(bright[b]), 0, 255).astype(np.uint8)
        want = norm(T.ToTensor()(a))
        assert torch.allclose(out[b].cpu(), want, atol=1e-6), b
    # test-time transform: no crop shift, no flip, no jitter
    p0 = torch.tensor([[0, 0, 0]] * B, dtype=torch.int32).to(DEV)
    call("clhip_augment_crop_flip", sd.data_ptr(), idx_d.data_ptr(), p0.data_ptr(), None, out.data_ptr(), B, 32, 32, S, 0, mean, std, st())
    torch.cuda.synchronize()
    for b in range(B):
        assert torch.allclose(out[b].cpu(), norm(T.ToTensor()(store[int(idx[b])])), atol=1e-6)


def test_rrc_kernel_close_to_pil_bilinear():
    from PIL import Image
    store = store_of(6, 32, seed=3)
    B, S = 6, 64
    params = torch.tensor([[0, 0, 32, 32, 0], [3, 5, 20, 17, 1], [10, 2, 9, 28, 0], [0, 7, 32, 11, 1], [16, 16, 16, 16, 0], [1, 1, 30, 30, 1]], dtype=torch.int32)
    idx = torch.arange(B)
    out = torch.empty(B, 3, S, S, device=DEV)
    mean, std = (C.c_float * 3)(0.0, 0.0, 0.0), (C.c_float * 3)(1.0, 1.0, 1.0)
    sd, idx_d, par_d = torch.as_tensor(store).to(DEV), idx.to(DEV), params.to(DEV)
    call("clhip_augment_rrc_flip", sd.data_ptr(), idx_d.data_ptr(), par_d.data_ptr(), out.data_ptr(), B, 32, 32, S, mean, std, st())
    torch.cuda.synchronize()
    for b in range(B):
        y0, x0, h, w, flip = (int(v) for v in params[b])
        im = Image.fromarray(store[b]).crop((x0, y0, x0 + w, y0 + h)).resize((S, S), Image.BILINEAR)      # crop, then resize
        a = np.asarray(im)
        if flip:
            a = a[:, ::-1]
        want = T.ToTensor()(a)
        diff = (out[b].cpu() - want).abs() * 255
        assert float(diff.max()) <= 1.01 and float(diff.mean()) < 0.35, (b, float(diff.max()), float(diff.mean()))   # PIL: two fixed-point passes with an intermediate uint8 rounding


def test_loader_contract_and_sharding():
    store = store_of(50)
    labels = list(np.arange(50) % 5)
    ds = ArrayDataset(store, list(range(50)), labels, T.cifar_resnet_transform("train", 32), "train")
    assert gpu_plan(ds.trfms)["kind"] == "crop_flip" and gpu_plan(ds.trfms)["brightness"] > 0
    assert gpu_plan(T.create_transforms([{"Resize": {"size": 36}}, {"ToTensor": {}}])) is None        # no GPU plan -> CPU DataLoader
    loader = make_loader(ds, 16, True, 0, DEV)
    assert isinstance(loader, GpuBatchLoader) and len(loader) == 4 and loader.batch_size == 16 and loader.dataset is ds
    torch.manual_seed(0)
    seen = []
    for batch in loader:
        assert batch["image"].is_cuda and batch["image"].dtype == torch.float32 and batch["image"].shape[1:] == (3, 32, 32)
        assert batch["label"].is_cuda and batch["label"].dtype == torch.int64 and torch.isfinite(batch["image"]).all()
        seen += batch["label"].tolist()
    assert sorted(seen) == sorted(labels)                                  # one epoch = every sample once
    parts = [loader.shard(r, 2) for r in range(2)]
    torch.manual_seed(5)
    a = [l for b in parts[0] for l in b["label"].tolist()]
    torch.manual_seed(5)
    b = [l for bt in parts[1] for l in bt["label"].tolist()]
    assert len(a) + len(b) == 50 and parts[0].batch_size == 8
    # test-mode views and copies share the resident store
    import copy
    ds2 = copy.deepcopy(ds)
    ds2.images = ds2.images[:10]; ds2.labels = ds2.labels[:10]
    assert ds2.device_store(DEV) is ds.device_store(DEV) and len(ds.images) == 50
    ev = make_loader(ArrayDataset(store, list(range(50)), labels, T.cifar_resnet_transform("test", 32), "test"), 25, False, 0, DEV)
    first = next(iter(ev))
    want = T.cifar_resnet_transform("test", 32)(store[0])
    assert torch.allclose(first["image"][0].cpu(), want, atol=1e-6) and first["label"].tolist() == labels[:25]


def test_rrc_loader_for_vit_pipeline():
    store = store_of(24)
    ds = ArrayDataset(store, list(range(24)), [0] * 24,
                      T.create_transforms([{"RandomResizedCrop": {"size": 64, "scale": [0.05, 1.0], "ratio": [0.75, 1.3333], "interpolation": "BILINEAR"}},
                                           {"RandomHorizontalFlip": {"p": 0.5}}, {"ToTensor": {}}]), "train")
    loader = make_loader(ds, 8, True, 0, DEV)
    assert isinstance(loader, GpuBatchLoader) and loader.plan["kind"] == "rrc_flip"
    torch.manual_seed(1)
    for batch in loader:
        x = batch["image"]
        assert x.shape == (8, 3, 64, 64) and float(x.min()) >= 0.0 and float(x.max()) <= 1.0


def test_augment_kernels_against_the_committed_reference_fixture(golden):
    """f2 pinned: tests/golden/augment.npz holds images, explicit random parameters and the outputs of the reference's
    CIFAR pipelines (core/data/data.py:4-35: RandomCrop(32, padding 4) + flip + ColorJitter(brightness 63/255) + ToTensor + Normalize;
    RandomResizedCrop(224) + flip + ToTensor) computed by oracle/gen_augment_golden.py with PIL -- the library torchvision itself calls
    for PIL images -- from the parameter tables parsed out of the reference's source.  The crop / flip / jitter / normalise kernel is
    exact (1e-6); the bilinear up-scaling crop is within one uint8 step of PIL's fixed-point two-pass resize."""
    fx = golden("augment")
    assert np.allclose(fx["mean"], T.CIFAR_MEAN) and np.allclose(fx["std"], T.CIFAR_STD)            # the product's tables ARE the reference's
    pad, S = int(fx["padding"][0]), 32
    imgs, B = fx["cifar_images"], len(fx["cifar_images"])
    sd = torch.as_tensor(imgs).to(DEV)
    idx = torch.arange(B).to(DEV)
    par = torch.as_tensor(np.concatenate([fx["cifar_offsets"], fx["cifar_flip"][:, None]], 1).astype(np.int32)).to(DEV)
    bri = torch.as_tensor(fx["cifar_brightness"]).to(DEV)
    out = torch.empty(B, 3, S, S, device=DEV)
    mean, std = (C.c_float * 3)(*[float(v) for v in fx["mean"]]), (C.c_float * 3)(*[float(v) for v in fx["std"]])
    call("clhip_augment_crop_flip", sd.data_ptr(), idx.data_ptr(), par.data_ptr(), bri.data_ptr(), out.data_ptr(), B, 32, 32, S, pad, mean, std, st())
    torch.cuda.synchronize()
    assert float((out.cpu() - torch.as_tensor(fx["cifar_train_expected"])).abs().max()) <= 1e-6
    p0 = torch.zeros(B, 3, dtype=torch.int32, device=DEV)
    call("clhip_augment_crop_flip", sd.data_ptr(), idx.data_ptr(), p0.data_ptr(), None, out.data_ptr(), B, 32, 32, S, 0, mean, std, st())
    torch.cuda.synchronize()
    nt = len(fx["cifar_test_expected"])
    assert float((out[:nt].cpu() - torch.as_tensor(fx["cifar_test_expected"])).abs().max()) <= 1e-6
    # the jitter range the loader draws from is the reference's
    ds = ArrayDataset(imgs, list(range(B)), [0] * B, T.cifar_resnet_transform("train", 32), "train")
    assert abs(gpu_plan(ds.trfms)["brightness"] - float(fx["brightness"][0])) < 1e-9 and gpu_plan(ds.trfms)["pad"] == pad
    # RandomResizedCrop + flip, every case at its own output size (the reference's 224 among them)
    vs = torch.as_tensor(fx["vit_images"]).to(DEV)
    m0, s1 = (C.c_float * 3)(*[float(v) for v in fx["vit_mean"]]), (C.c_float * 3)(*[float(v) for v in fx["vit_std"]])
    for k, size in enumerate(int(v) for v in fx["vit_sizes"]):
        o = torch.empty(1, 3, size, size, device=DEV)
        ik = torch.tensor([k]).to(DEV)
        pk = torch.as_tensor(fx["vit_boxes"][k:k + 1].astype(np.int32)).to(DEV)
        call("clhip_augment_rrc_flip", vs.data_ptr(), ik.data_ptr(), pk.data_ptr(), o.data_ptr(), 1, 32, 32, size, m0, s1, st())
        torch.cuda.synchronize()
        want = torch.as_tensor(fx[f"vit_train_expected_u8_{k}"]).permute(2, 0, 1).float() / 255.0
        diff = (o[0].cpu() - want).abs() * 255
        assert float(diff.max()) <= 1.01 and float(diff.mean()) < 0.35, (k, size, float(diff.max()), float(diff.mean()))


def _aa(flat, offsets, hw, idx, params, S, uniform=None):
    B = len(params)
    max_box = int(np.asarray(params)[:, 2:4].max())
    from libcontinual_amd import _lib
    ws = torch.empty(_lib.lib().clhip_augment_rrc_aa_ws_bytes(B, S, max_box), dtype=torch.uint8, device=DEV)
    out = torch.empty(B, 3, S, S, device=DEV)
    mean, std = (C.c_float * 3)(0.0, 0.0, 0.0), (C.c_float * 3)(1.0, 1.0, 1.0)
    keep = [torch.as_tensor(flat).to(DEV), torch.as_tensor(np.asarray(idx, np.int64)).to(DEV), torch.as_tensor(np.asarray(params, np.int32)).to(DEV)]
    if uniform is None:
        keep += [torch.as_tensor(offsets).to(DEV), torch.as_tensor(hw).to(DEV)]
        call("clhip_augment_rrc_aa", keep[0].data_ptr(), keep[3].data_ptr(), keep[4].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr(), out.data_ptr(),
             ws.data_ptr(), B, 0, 0, S, max_box, mean, std, st())
    else:
        call("clhip_augment_rrc_aa", keep[0].data_ptr(), None, None, keep[1].data_ptr(), keep[2].data_ptr(), out.data_ptr(), ws.data_ptr(), B,
             uniform[0], uniform[1], S, max_box, mean, std, st())
    torch.cuda.synchronize()
    return (out.cpu() * 255).round().permute(0, 2, 3, 1).numpy().astype(np.uint8), out.cpu()


def test_antialiased_rrc_is_bit_exact_on_the_reference_fixture_and_random_ragged_stores(golden):
    """f2 for ImageNet-R: the anti-aliased RandomResizedCrop kernel on a ragged store equals Pillow's resize bit for
    bit -- against tests/golden/augment_aa.npz (boxes and outputs of the reference's YAML pipeline, 224 x 224 case included), against
    the oracle on random images / boxes (shrinking by up to 12x, enlarging, flips), and on a uniform store."""
    from libcontinual_amd.data.dataset import RaggedStore
    from oracle.resample import resized_crop
    fx = golden("augment_aa")
    store = RaggedStore([fx[f"image_{k}"] for k in range(len(fx["hw"]))])
    assert np.array_equal(store.hw, fx["hw"])
    for S in sorted(set(int(v) for v in fx["out_sizes"])):
        sel = [j for j in range(len(fx["boxes"])) if int(fx["out_sizes"][j]) == S]
        got, raw = _aa(store.flat, store.offsets[:-1].copy(), store.hw, fx["which"][sel], fx["boxes"][sel], S)
        for n, j in enumerate(sel):
            assert np.array_equal(got[n], fx[f"train_expected_u8_{j}"]), (S, j)
            assert float((raw[n] - torch.as_tensor(fx[f"train_expected_u8_{j}"]).permute(2, 0, 1).float() / 255.0).abs().max()) <= 1e-7   # ToTensor
    rs = np.random.RandomState(5)
    frames = [(rs.rand(int(rs.randint(8, 400)), int(rs.randint(8, 400)), 3) * 255).astype(np.uint8) for _ in range(9)]
    rag = RaggedStore(frames)
    idx, params = [], []
    for n in range(24):
        k = int(rs.randint(0, 9))
        H, W = frames[k].shape[:2]
        h, w = int(rs.randint(1, H + 1)), int(rs.randint(1, W + 1))
        idx.append(k)
        params.append([int(rs.randint(0, H - h + 1)), int(rs.randint(0, W - w + 1)), h, w, int(rs.randint(0, 2))])
    for S in (32, 7):
        got, _ = _aa(rag.flat, rag.offsets[:-1].copy(), rag.hw, idx, params, S)
        for n, (k, (y0, x0, h, w, fl)) in enumerate(zip(idx, params)):
            want = resized_crop(frames[k], y0, x0, h, w, S, S)
            assert np.array_equal(got[n], want[:, ::-1] if fl else want), (S, n, frames[k].shape, params[n])
    uni = (rs.rand(4, 96, 80, 3) * 255).astype(np.uint8)
    par = [[0, 0, 96, 80, 0], [10, 5, 70, 60, 1], [40, 40, 8, 8, 0], [1, 2, 90, 33, 1]]
    got, _ = _aa(uni.reshape(-1), None, None, [0, 1, 2, 3], par, 24, uniform=(96, 80))
    for n, (y0, x0, h, w, fl) in enumerate(par):
        want = resized_crop(uni[n], y0, x0, h, w, 24, 24)
        assert np.array_equal(got[n], want[:, ::-1] if fl else want), n


def test_ragged_loader_epoch_matches_pillow_for_the_boxes_it_drew():
    """GpuBatchLoader over a RaggedStore (ImageNet-R layout): one epoch covers every sample once, batches are [B,3,S,S] on the device,
    and re-drawing the epoch's boxes with the same seed reproduces every image with Pillow on the host."""
    from PIL import Image
    from libcontinual_amd.data.dataset import RaggedStore
    from libcontinual_amd.data.gpu_loader import _rrc_boxes
    rs = np.random.RandomState(2)
    frames = [(rs.rand(int(rs.randint(40, 260)), int(rs.randint(40, 260)), 3) * 255).astype(np.uint8) for _ in range(21)]
    tr = T.create_transforms([{"RandomResizedCrop": {"size": 48, "scale": [0.05, 1.0], "ratio": [0.75, 1.333]}}, {"RandomHorizontalFlip": {"p": 0.5}},
                              {"ToTensor": {}}])
    ds = ArrayDataset(RaggedStore(frames), list(range(21)), list(range(21)), tr, "train")
    loader = make_loader(ds, 8, True, 0, DEV)
    assert isinstance(loader, GpuBatchLoader) and loader.plan["kind"] == "rrc_aa" and len(loader) == 3
    torch.manual_seed(3)
    batches = [(b["image"].cpu(), b["label"].tolist()) for b in loader]
    assert sorted(l for _, ls in batches for l in ls) == list(range(21)) and batches[0][0].shape == (8, 3, 48, 48)
    # the loader's draw order: permutation, flips, boxes
    torch.manual_seed(3)
    order = torch.randperm(21)
    flip = (torch.rand(21) < 0.5).int()
    hw = torch.as_tensor(ds.store.hw)[order].long()
    boxes = _rrc_boxes(hw[:, 0], hw[:, 1], (0.05, 1.0), (0.75, 1.333))
    pos = 0
    for img, labels in batches:
        for n, l in enumerate(labels):
            assert l == int(order[pos])
            y0, x0, h, w = (int(v) for v in boxes[pos])
            want = np.asarray(Image.fromarray(frames[l]).crop((x0, y0, x0 + w, y0 + h)).resize((48, 48), Image.BILINEAR))
            if int(flip[pos]):
                want = want[:, ::-1]
            assert np.array_equal((img[n] * 255).round().permute(1, 2, 0).numpy().astype(np.uint8), want), (pos, l)
            pos += 1
