"""Golden scenarios (TEST INFRASTRUCTURE, see oracle/__init__).

One scenario = one deterministic script of plugin calls.  It is executed by

* the REFERENCE  (`oracle/gen_golden.py`, build container only)  -> `tests/golden/*.npz`
* the ORACLE     (`tests/test_oracle_golden.py`, CPU)            -> must match the fixtures
* the PRODUCT    (`tests/test_parity_gpu.py`, MI355X)            -> must match within the bf16 tolerance

Reference and product expose the same plugin surface (that is the drop-in boundary), so they share
`PluginAdapter`; the oracle has its own functional API and uses `OracleAdapter`.
"""
import copy
import os

import numpy as np
import torch

from . import detrand
from . import fixtures as fx
from . import methods as om
from . import nets


# ----------------------------------------------------------------------------------- data helpers
class ListLoader:
    """Minimal loader: iterable of batch dicts + the attributes the reference reads
    (`batch_size`, `__len__`, `dataset`, `num_workers`, `pin_memory`)."""

    def __init__(self, batches, batch_size, dataset=None):
        self.batches = batches
        self.batch_size = batch_size
        self.dataset = dataset
        self.num_workers = 0
        self.pin_memory = False

    def __iter__(self):
        for x, y in self.batches:
            yield {"image": x, "label": y}

    def __len__(self):
        return len(self.batches)


_PNG_CACHE = {}


def _decoded(path):
    """the decoded image of a scenario file (the files of a data root never change; an accuracy scenario reads each of its 6 000 files ~12 times per run and the
    GPU suite makes up to 16 runs per scenario: half of a run's wall-clock was PNG decoding).  Transforms get the cached PIL image and must not modify it."""
    st = os.stat(path)
    key = (path, st.st_mtime_ns, st.st_size)                # (a rewritten file is a different entry)
    img = _PNG_CACHE.get(key)
    if img is None:
        from PIL import Image
        img = Image.open(path).convert("RGB")
        img.load()
        if len(_PNG_CACHE) < 20000:
            _PNG_CACHE[key] = img
    return img


class PngDataset(torch.utils.data.Dataset):
    """Class-folder style dataset over PNG files (reference layout docs/tutorials/en/data_module_en.md:15-39):
    `images` are paths relative to data_root/mode, `trfms` maps a PIL image to a tensor."""

    def __init__(self, data_root, mode, images, labels, trfms):
        self.data_root, self.mode = data_root, mode
        self.images, self.labels, self.trfms = list(images), list(labels), trfms

    def __len__(self):
        return len(self.labels)

    def __getitem__(self, i):
        return {"image": self.trfms(_decoded(os.path.join(self.data_root, self.mode, self.images[i]))), "label": int(self.labels[i])}


def png_transform(img):
    a = np.asarray(img, dtype=np.float32) / 255.0              # HWC
    m = np.asarray(fx.CIFAR_MEAN, np.float32)
    s = np.asarray(fx.CIFAR_STD, np.float32)
    return fx._t(((a - m) / s).transpose(2, 0, 1).copy())


def write_class_pngs(root, mode, tag, classes, per_class):
    """-> (images, labels) with files root/mode/<cls>/<k>.png built from detrand data."""
    from PIL import Image
    images, labels = [], []
    for c in classes:
        d = os.path.join(root, mode, f"{c:03d}")
        os.makedirs(d, exist_ok=True)
        pat = detrand.uniform(f"{tag}/pattern{c}", (32, 32, 3), 0.0, 1.0)
        raw = detrand.uniform(f"{tag}/c{c}", (per_class, 32, 32, 3), 0.0, 1.0)
        for k in range(per_class):
            a = np.clip((0.6 * raw[k] + 0.4 * pat) * 255.0, 0, 255).astype(np.uint8)
            rel = os.path.join(f"{c:03d}", f"{k}.png")
            Image.fromarray(a).save(os.path.join(root, mode, rel))
            images.append(rel)
            labels.append(int(c))
    return images, labels


# --------------------------------------------------------------------------------------- adapters
class PluginAdapter:
    """Drives classes with the LibContinual plugin surface found in `ns` (the reference's
    `core.model` modules, or `libcontinual_amd.model`)."""

    kind = "plugin"

    def __init__(self, ns, device="cpu", sgd_factory=None):
        self.ns, self.device = ns, device
        self.sgd_factory = sgd_factory or (lambda params, **kw: torch.optim.SGD(params, **kw))

    def backbone(self, arch, P, Bf):
        if arch == "resnet18":
            bb = self.ns.resnet18(num_classes=100, args={"dataset": "cifar100"})
        else:
            bb = getattr(self.ns, arch)()
        sd = {k: v.clone() for k, v in P.items()}
        sd.update({k: v.clone() for k, v in Bf.items()})
        bb = bb.to(fx._DTYPE[0])
        bb.load_state_dict(sd)
        return bb.to(self.device)

    def to_dev(self, t):
        return t.to(self.device)

    def batch(self, x, y):
        return {"image": x, "label": y}


class OracleAdapter:
    kind = "oracle"
    device = "cpu"


# ------------------------------------------------------------------------------ scenario: backbone
def scenario_backbone(adapter, arch, B=4):
    """features / all parameter grads / running stats for one train-mode batch, then eval features."""
    tag = f"bb/{arch}"
    P, Bf = fx.det_backbone_state(arch, tag)
    x = fx.det_images(tag + "/x", B, size=64 if arch == nets.PREACT else 32)      # ResNet_BIC's feat_dim assumes 64 x 64 inputs

    def feats_of(o):          # ResNet_BIC.forward returns the tensor, the other backbones a dict
        return o["features"] if isinstance(o, dict) else o
    _, feat_dim, _ = nets.arch(arch)
    cw = fx._t(detrand.uniform(tag + "/cw", (B, feat_dim), -1, 1))
    out = {}
    if adapter.kind == "oracle":
        Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        Bb = {k: v.clone() for k, v in Bf.items()}
        f = nets.forward(arch, Pg, Bb, x, True)
        (f * cw).sum().backward()
        grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in Pg.items()}
        bufs = Bb
        fe = nets.forward(arch, Pg, Bb, x, False).detach()
    else:
        bb = adapter.backbone(arch, P, Bf)
        bb.train()
        f = feats_of(bb(adapter.to_dev(x)))
        (f * adapter.to_dev(cw)).sum().backward()
        grads = {k: (p.grad.detach().cpu() if p.grad is not None else torch.zeros(p.shape)) for k, p in bb.named_parameters()}
        bufs = {k: b.detach().cpu() for k, b in bb.named_buffers()}
        bb.eval()
        with torch.no_grad():
            fe = feats_of(bb(adapter.to_dev(x))).detach().cpu()
    out["features_train"] = f.detach().cpu().numpy()
    out["features_eval"] = fe.numpy()
    names, rows = fx.summarize(grads)
    out["grad_names"] = np.asarray(names)
    out["grad_rows"] = rows
    first = nets.param_shapes(arch)[0][0]
    out["grad_stem"] = grads[first].numpy()
    bn_names = [n for n, _ in nets.buffer_shapes(arch) if "running" in n]
    names_b, rows_b = fx.summarize({n: bufs[n] for n in bn_names})
    out["buf_rows"] = rows_b
    return out


# ----------------------------------------------------------------------------------- scenario: EWC
# lr is 5x smaller than config/ewc.yaml's 0.1: with batch 8 and deterministic (non-trained) weights a 0.1 step
# moves the stem by ~7 % per step, which turns fp32 rounding into O(10 %) loss differences two steps later and
# would force meaningless tolerances on the GPU comparisons.
EWC_CFG = dict(arch="cifar_resnet32", feat_dim=64, init=6, inc=2, lamda=1000.0, bs=8,
               lr=0.02, momentum=0.9, wd=5e-4)


def scenario_ewc(adapter):
    """task 0: 2 SGD steps; after_task (Fisher over 3 batches, last one ragged); task 1: 2 steps with
    the penalty active.  Mirrors trainer.py:283-407 call order for method 'EWC'."""
    c = EWC_CFG
    tag = "ewc"
    P, Bf = fx.det_backbone_state(c["arch"], tag)
    w0, b0 = fx.det_linear(tag + "/head0", c["init"], c["feat_dim"])
    w1, b1 = fx.det_linear(tag + "/head1", c["init"] + c["inc"], c["feat_dim"])
    t0 = [fx.det_batch(f"{tag}/t0/{i}", c["bs"], 0, c["init"]) for i in range(2)]
    fb = [fx.det_batch(f"{tag}/fisher/{i}", c["bs"] if i < 2 else 5, 0, c["init"]) for i in range(3)]
    t1 = [fx.det_batch(f"{tag}/t1/{i}", c["bs"], c["init"], c["init"] + c["inc"]) for i in range(2)]
    losses, preds, accs = [], [], []
    if adapter.kind == "oracle":
        net = om.Net(c["arch"], {k: v.clone().requires_grad_(True) for k, v in P.items()},
                     {k: v.clone() for k, v in Bf.items()}, w0.clone().requires_grad_(True), b0.clone().requires_grad_(True))
        m = om.EWC(net, c["init"], c["inc"], c["lamda"])

        def run(batches):
            opt = om.SGD(net.parameters(), c["lr"], c["momentum"], c["wd"])
            for x, y in batches:
                pred, acc, loss = m.observe(x, y, True)
                opt.zero_grad(); loss.backward(); opt.step()
                losses.append(loss.item()); preds.append(pred.numpy()); accs.append(acc)
        m.before_task(0, new_rows=(w0, b0))
        run(t0)
        m.after_task(fb, c["bs"])
        fisher0 = {k: v.clone() for k, v in m.fisher.items()}
        m.before_task(1, new_rows=(w1, b1))
        run(t1)
        params = dict(net.named_parameters())
        bufs = net.Bf
        fisher = fisher0
    else:
        ns = adapter.ns
        bb = adapter.backbone(c["arch"], P, Bf)
        m = ns.EWC(bb, c["feat_dim"], 100, device=adapter.device, init_cls_num=c["init"], inc_cls_num=c["inc"],
                   lamda=c["lamda"]).to(adapter.device)

        def set_head(w, b, old):
            with torch.no_grad():
                m.network.classifier.weight.data[old:] = adapter.to_dev(w[old:])
                m.network.classifier.bias.data[old:] = adapter.to_dev(b[old:])

        def run(batches):
            opt = adapter.sgd_factory(m.get_parameters({}), lr=c["lr"], momentum=c["momentum"], weight_decay=c["wd"])
            m.train()
            for x, y in batches:
                pred, acc, loss = m.observe(adapter.batch(x, y))
                opt.zero_grad(); loss.backward(); opt.step()
                losses.append(float(loss.item())); preds.append(pred.cpu().numpy()); accs.append(acc)
        m.before_task(0, None, None, None)
        set_head(w0, b0, 0)
        # EWC.__init__ snapshots ref_param before we load the head; irrelevant at task 0 (penalty inactive)
        run(t0)
        m.after_task(0, None, ListLoader(fb, c["bs"]), None)
        fisher = {k: v.detach().cpu().clone() for k, v in m.fisher.items()}
        m.before_task(1, None, None, None)
        set_head(w1, b1, c["init"])
        run(t1)
        params = {k: v.detach().cpu() for k, v in m.network.named_parameters()}
        bufs = {k: v.detach().cpu() for k, v in m.network.backbone.named_buffers()}
    out = dict(losses=np.asarray(losses, np.float64), preds=np.stack(preds), accs=np.asarray(accs, np.float64))
    names, rows = fx.summarize(fisher)
    out["fisher_names"], out["fisher_rows"] = np.asarray(names), rows
    out["fisher_head_w"] = fisher["classifier.weight"].numpy()
    out["fisher_bn1"] = fisher["backbone.bn_1.weight"].numpy()
    names, rows = fx.summarize(params)
    out["param_names"], out["param_rows"] = np.asarray(names), rows
    out["head_w"] = params["classifier.weight"].detach().numpy()
    out["rm_last"] = bufs["stage_3.4.bn_b.running_mean"].numpy()
    return out


def fisher_probe_index(name, numel, k=64):
    """the element indices of a Fisher tensor the fixtures keep (all of them for small tensors)"""
    if numel <= k:
        return np.arange(numel)
    return np.sort(detrand.randint(f"fisher_probe/{name}", (k,), 0, numel))


def scenario_ewc_fisher(adapter):
    """The Fisher diagonal of EVERY parameter tensor on fixed (deterministic, untrained) weights: before_task(0), then
    after_task(0) over three batches (the last one ragged) -- ewc.py:147-205 with no optimisation step in front of it, so that a
    comparison tests the Fisher pass itself (forward, per-sample-batch backward, squared-gradient accumulation, 1/N) and not the
    drift of a training trajectory.  Kept per tensor: sum, abs-sum, sq-sum and 64 probed elements."""
    c = EWC_CFG
    tag = "ewc_fisher"
    P, Bf = fx.det_backbone_state(c["arch"], tag)
    w0, b0 = fx.det_linear(tag + "/head0", c["init"], c["feat_dim"])
    fb = [fx.det_batch(f"{tag}/fisher/{i}", c["bs"] if i < 2 else 5, 0, c["init"]) for i in range(3)]
    if adapter.kind == "oracle":
        net = om.Net(c["arch"], {k: v.clone().requires_grad_(True) for k, v in P.items()},
                     {k: v.clone() for k, v in Bf.items()}, w0.clone().requires_grad_(True), b0.clone().requires_grad_(True))
        m = om.EWC(net, c["init"], c["inc"], c["lamda"])
        m.before_task(0, new_rows=(w0, b0))
        m.after_task(fb, c["bs"])
        fisher = {k: v.clone() for k, v in m.fisher.items()}
    else:
        ns = adapter.ns
        bb = adapter.backbone(c["arch"], P, Bf)
        m = ns.EWC(bb, c["feat_dim"], 100, device=adapter.device, init_cls_num=c["init"], inc_cls_num=c["inc"], lamda=c["lamda"]).to(adapter.device)
        m.before_task(0, None, None, None)
        with torch.no_grad():
            m.network.classifier.weight.data[:] = adapter.to_dev(w0)
            m.network.classifier.bias.data[:] = adapter.to_dev(b0)
        m.after_task(0, None, ListLoader(fb, c["bs"]), None)
        fisher = {k: v.detach().cpu().clone() for k, v in m.fisher.items()}
    names, rows = fx.summarize(fisher)
    out = {"fisher_names": np.asarray(names), "fisher_rows": rows}
    probes = []
    for n in names:
        a = fisher[n].detach().double().reshape(-1).numpy()
        idx = fisher_probe_index(n, a.size)
        probes.append(np.pad(a[idx], (0, 64 - len(idx))))
    out["fisher_probes"] = np.stack(probes)
    return out


# ----------------------------------------------------------------------------------- scenario: LwF
LWF_CFG = dict(arch="resnet18", feat_dim=512, init=6, inc=2, bs=4, lr=0.02)
# a longer trajectory through the same plugin: 6 + 6 steps with momentum, weight decay and a learning-rate drop inside each
# task (fresh optimizer per task, trainer.py:294) -- pins the optimizer / scheduler / gradient-buffer handling over many steps
LWF_LONG_CFG = dict(arch="cifar_resnet32", feat_dim=64, bs=32, lr=0.01, momentum=0.9, wd=5e-4, n0=6, n1=6, decay_at=4, tag="lwf_long")


def scenario_lwf(adapter, cfg=None):
    """task 0: 1 step (CE); task 1: 2 steps with CE(new) + 3*KD(T=2) against the frozen copy whose BN is
    (quirk a10) in train mode.  SGD lr .1 without momentum (config/lwf.yaml:14-17)."""
    c = dict(LWF_CFG, **(cfg or {}))
    tag = c.get("tag", "lwf") + "/" + c["arch"]
    P, Bf = fx.det_backbone_state(c["arch"], tag)
    w0, b0 = fx.det_linear(tag + "/head0", c["init"], c["feat_dim"])
    w1, b1 = fx.det_linear(tag + "/head1", c["init"] + c["inc"], c["feat_dim"])
    t0 = [fx.det_batch(f"{tag}/t0/{i}", c["bs"], 0, c["init"]) for i in range(c.get("n0", 1))]
    t1 = [fx.det_batch(f"{tag}/t1/{i}", c["bs"], c["init"], c["init"] + c["inc"]) for i in range(c.get("n1", 2))]
    losses, preds = [], []
    mom, wd, decay_at = c.get("momentum", 0.0), c.get("wd", 0.0), c.get("decay_at")      # decay_at: lr *= 0.1 from that step on

    def lr_of(step):
        return c["lr"] * (0.1 if decay_at is not None and step >= decay_at else 1.0)
    if adapter.kind == "oracle":
        net = om.Net(c["arch"], {k: v.clone().requires_grad_(True) for k, v in P.items()},
                     {k: v.clone() for k, v in Bf.items()}, w0.clone().requires_grad_(True), b0.clone().requires_grad_(True))
        m = om.LWF(net, c["init"], c["inc"])

        def run(batches):
            opt = om.SGD(net.parameters(), c["lr"], mom, wd)
            for i, (x, y) in enumerate(batches):
                opt.lr = lr_of(i)
                pred, acc, loss = m.observe(x, y, True)
                opt.zero_grad(); loss.backward(); opt.step()
                losses.append(loss.item()); preds.append(pred.numpy())
        m.before_task(0, new_rows=(w0, b0)); run(t0)
        m.before_task(1, new_rows=(w1, b1)); run(t1)
        logits = net.logits(t1[0][0], False).detach()
        teacher_rm = m.old.Bf[_last_bn(c["arch"]) + ".running_mean"]
        params = dict(net.named_parameters())
    else:
        ns = adapter.ns
        bb = adapter.backbone(c["arch"], P, Bf)
        m = ns.LWF(bb, c["feat_dim"], 100, device=adapter.device, init_cls_num=c["init"], inc_cls_num=c["inc"]).to(adapter.device)

        def set_head(w, b, old):
            with torch.no_grad():
                m.classifier.weight.data[old:] = adapter.to_dev(w[old:])
                m.classifier.bias.data[old:] = adapter.to_dev(b[old:])

        def run(batches):
            kw = dict(momentum=mom, weight_decay=wd) if (mom or wd) else {}
            opt = adapter.sgd_factory(m.get_parameters({}), lr=c["lr"], **kw)
            sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda e: lr_of(e) / c["lr"])     # stepped per batch here
            m.train()
            for x, y in batches:
                pred, acc, loss = m.observe(adapter.batch(x, y))
                opt.zero_grad(); loss.backward(); opt.step(); sched.step()
                losses.append(float(loss.item())); preds.append(pred.cpu().numpy())
        m.before_task(0, None, None, None); set_head(w0, b0, 0); run(t0)
        m.before_task(1, None, None, None); set_head(w1, b1, c["init"]); run(t1)
        m.eval()
        with torch.no_grad():
            logits = m.classifier(m.backbone(adapter.to_dev(t1[0][0]))["features"]).cpu()
        teacher_rm = dict(m.old_backbone.named_buffers())[_last_bn(c["arch"]) + ".running_mean"].cpu()
        params = {("backbone." + k): v.detach().cpu() for k, v in m.backbone.named_parameters()}
        params["classifier.weight"] = m.classifier.weight.detach().cpu()
        params["classifier.bias"] = m.classifier.bias.detach().cpu()
    out = dict(losses=np.asarray(losses, np.float64), preds=np.stack(preds), logits_eval=logits.numpy(),
               teacher_rm=teacher_rm.numpy())
    names, rows = fx.summarize(params)
    out["param_names"], out["param_rows"] = np.asarray(names), rows
    return out


def _last_bn(arch):
    return nets.arch(arch)[0][-1].bn


# --------------------------------------------------------------------------------- scenario: iCaRL
ICARL_CFG = dict(arch="cifar_resnet32", feat_dim=64, init=4, inc=2, num_class=8, bs=8, lr=0.02, momentum=0.9,
                 wd=5e-4, buffer_size=24, per_class=12)


def scenario_icarl(adapter, tmpdir):
    """task 0 (4 classes x 12 PNG images): 2 steps; after_task: teacher snapshot, herding into a
    LinearHerdingBuffer(24) (6/class), class means; NCM inference; task 1: rehearsal union batches with
    CE + KD; after_task again (buffer reduced to 4/class)."""
    c = ICARL_CFG
    tag = "icarl"
    P, Bf = fx.det_backbone_state(c["arch"], tag)
    hw, hb = fx.det_linear(tag + "/head", c["num_class"], c["feat_dim"])
    root = os.path.join(tmpdir, "icarl_data")
    imgs0, labs0 = write_class_pngs(root, "train", tag, range(0, c["init"]), c["per_class"])
    imgs1, labs1 = write_class_pngs(root, "train", tag, range(c["init"], c["init"] + c["inc"]), c["per_class"])
    timgs, tlabs = write_class_pngs(root, "test", tag + "/test", range(0, c["init"] + c["inc"]), 4)

    def load(images, labels, mode="train"):
        ds = PngDataset(root, mode, images, labels, png_transform)
        xs = torch.stack([ds[i]["image"] for i in range(len(ds))])
        return xs, torch.tensor(labels, dtype=torch.long)

    def batches_of(xs, ys, order):
        out = []
        for s in range(0, len(order), c["bs"]):
            idx = torch.tensor(order[s:s + c["bs"]])
            out.append((xs[idx], ys[idx]))
        return out

    x0, y0 = load(imgs0, labs0)
    order0 = list(np.argsort(detrand.uniform(tag + "/perm0", (len(labs0),)))[: 2 * c["bs"]])
    tx, ty = load(timgs, tlabs, "test")
    res = {}
    losses, preds = [], []
    if adapter.kind == "oracle":
        net = om.Net(c["arch"], {k: v.clone().requires_grad_(True) for k, v in P.items()},
                     {k: v.clone() for k, v in Bf.items()}, hw.clone().requires_grad_(True), hb.clone().requires_grad_(True))
        m = om.ICarl(net, c["init"], c["inc"])

        def run(batches):
            opt = om.SGD(net.parameters(), c["lr"], c["momentum"], c["wd"])
            for x, y in batches:
                pred, acc, loss = m.observe(x, y, True)
                opt.zero_grad(); loss.backward(); opt.step()
                losses.append(loss.item()); preds.append(pred.numpy())

        buf_images, buf_labels = [], []

        def after_task(images, labels, xs, ys, cur_classes):
            nonlocal buf_images, buf_labels
            m.old_network = net.clone(grad=False)
            m.prev_cls_num = m.accu_cls_num
            spc = c["buffer_size"] // m.accu_cls_num
            if m.cur_task_id > 0:                      # reduce_old_data (linearherdingbuffer.py:55-75)
                bi, bl = np.array(buf_images), np.array(buf_labels)
                buf_images, buf_labels = [], []
                for yy in np.unique(bl):
                    sel = bl == yy
                    buf_images.extend(bi[sel][:spc]); buf_labels.extend(bl[sel][:spc])
            # herding over current-class samples only, in class order (linearherdingbuffer.py:83-100)
            keep = [i for cc in cur_classes for i in range(len(labels)) if labels[i] == cc]
            k_imgs = [images[i] for i in keep]; k_labs = [labels[i] for i in keep]
            kx, ky = xs[torch.tensor(keep)], ys[torch.tensor(keep)]
            with torch.no_grad():
                feats = torch.cat([net.features(kx[s:s + 32], False) for s in range(0, len(keep), 32)])
            chosen = om.herding_select(feats, ky, spc)
            buf_images.extend([k_imgs[i] for i in chosen]); buf_labels.extend([int(k_labs[i]) for i in chosen])
            bx, by = load(buf_images, buf_labels)
            with torch.no_grad():
                bf = torch.cat([net.features(bx[s:s + c["bs"]], False) for s in range(0, len(buf_labels), c["bs"])])
            m.class_means = om.class_means_from_buffer(bf, by)
            m.cur_task_id += 1
            return chosen

        m.before_task(0)
        run(batches_of(x0, y0, order0))
        chosen0 = after_task(imgs0, labs0, x0, y0, list(range(0, c["init"])))
        res["chosen0"] = np.asarray(chosen0)
        res["buffer_labels0"] = np.asarray(buf_labels)
        res["class_means0"] = m.class_means.numpy().copy()
        p, a = m.inference(tx[: 4 * c["init"]], ty[: 4 * c["init"]])
        res["ncm_pred0"] = p.numpy()
        m.before_task(1)
        u_imgs, u_labs = imgs1 + list(buf_images), labs1 + list(buf_labels)     # trainer.py:305-312
        ux, uy = load(u_imgs, u_labs)
        order1 = list(np.argsort(detrand.uniform(tag + "/perm1", (len(u_labs),)))[: 2 * c["bs"]])
        run(batches_of(ux, uy, order1))
        chosen1 = after_task(u_imgs, u_labs, ux, uy, list(range(c["init"], c["init"] + c["inc"])))
        res["chosen1"] = np.asarray(chosen1)
        res["buffer_labels1"] = np.asarray(buf_labels)
        res["class_means1"] = m.class_means.numpy().copy()
        p, a = m.inference(tx, ty)
        res["ncm_pred1"] = p.numpy()
        params = dict(net.named_parameters())
    else:
        ns = adapter.ns
        bb = adapter.backbone(c["arch"], P, Bf)
        m = ns.ICarl(bb, c["feat_dim"], c["num_class"], device=adapter.device, init_cls_num=c["init"],
                     inc_cls_num=c["inc"], task_num=2).to(adapter.device)
        with torch.no_grad():
            m.network.classifier.weight.data.copy_(adapter.to_dev(hw)); m.network.classifier.bias.data.copy_(adapter.to_dev(hb))
        buffer = ns.LinearHerdingBuffer(c["buffer_size"], 64)

        def run(batches):
            opt = adapter.sgd_factory(m.get_parameters({}), lr=c["lr"], momentum=c["momentum"], weight_decay=c["wd"])
            m.train()
            for x, y in batches:
                pred, acc, loss = m.observe(adapter.batch(x, y))
                opt.zero_grad(); loss.backward(); opt.step()
                losses.append(float(loss.item())); preds.append(pred.cpu().numpy())

        test_ds = PngDataset(root, "test", timgs, tlabs, png_transform)
        test_loaders = [ListLoader([], c["bs"], test_ds)]
        m.before_task(0, buffer, None, None)
        run(batches_of(x0, y0, order0))
        ds0 = PngDataset(root, "train", imgs0, labs0, png_transform)
        m.after_task(0, buffer, ListLoader([], c["bs"], ds0), test_loaders)
        res["chosen0"] = np.asarray([imgs0.index(p) for p in buffer.images])
        res["buffer_labels0"] = np.asarray([int(v) for v in buffer.labels])
        res["class_means0"] = m.class_means.detach().cpu().numpy().copy()
        m.eval()
        with torch.no_grad():
            p, a = m.inference(adapter.batch(tx[: 4 * c["init"]], ty[: 4 * c["init"]]))
        res["ncm_pred0"] = p.cpu().numpy()
        # NCM decision margins of the plugin under test (best vs second-best distance, relative): lets the parity
        # test tell a wrong decision from a near-tie that a 1e-2 change of the class means may flip
        with torch.no_grad():
            f = m.network.backbone(tx[: 4 * c["init"]].to(adapter.device))["features"].float().cpu().double()
        d = torch.cdist(f, torch.as_tensor(res["class_means0"]).double()) ** 2      # un-normalised features, as in inference
        ds_, _ = d.sort(dim=1)
        res["ncm_margin0"] = ((ds_[:, 1] - ds_[:, 0]) / ds_[:, 1]).numpy()
        # herding decision margins of the run under test (linearherdingbuffer.py:140-161 replayed on its own features, fp64): for every
        # class and every greedy pick the relative gap between the best and the second-best candidate's cost -- a pick may legitimately
        # differ between two implementations only where that gap is at rounding level
        with torch.no_grad():
            f0 = torch.cat([m.network.backbone(x0[i:i + 64].to(adapter.device))["features"].float().cpu().double() for i in range(0, len(y0), 64)])
        f0 = f0 / f0.norm(dim=1).view(-1, 1)
        per_class = len(res["chosen0"]) // c["init"]
        margins = []
        y0n = np.asarray([int(v) for v in y0])
        for cls in range(c["init"]):
            cf = f0[np.where(y0n == cls)[0]].clone()
            mean, run_sum = cf.mean(0, keepdim=True), torch.zeros(1, cf.shape[1], dtype=torch.float64)
            for k in range(min(per_class, cf.shape[0])):
                cost = (mean - (cf + run_sum) / (k + 1)).norm(2, 1)
                two = torch.topk(cost, 2, largest=False)
                margins.append(float((two.values[1] - two.values[0]) / two.values[1]))
                j = int(two.indices[0])
                run_sum += cf[j:j + 1]
                cf[j] = cf[j] + 1e6
        res["herding_margin0"] = np.asarray(margins).reshape(c["init"], -1)
        m.before_task(1, buffer, None, None)
        u_imgs, u_labs = imgs1 + list(buffer.images), labs1 + [int(v) for v in buffer.labels]
        ux, uy = load(u_imgs, u_labs)
        order1 = list(np.argsort(detrand.uniform(tag + "/perm1", (len(u_labs),)))[: 2 * c["bs"]])
        run(batches_of(ux, uy, order1))
        ds1 = PngDataset(root, "train", u_imgs, u_labs, png_transform)
        m.after_task(1, buffer, ListLoader([], c["bs"], ds1), test_loaders)
        new = list(buffer.images)[-(c["buffer_size"] // (c["init"] + c["inc"])) * c["inc"]:]
        kept = [p_ for cc in range(c["init"], c["init"] + c["inc"]) for p_, l_ in zip(u_imgs, u_labs) if l_ == cc]
        res["chosen1"] = np.asarray([kept.index(p_) for p_ in new])
        res["buffer_labels1"] = np.asarray([int(v) for v in buffer.labels])
        res["class_means1"] = m.class_means.detach().cpu().numpy().copy()
        m.eval()
        with torch.no_grad():
            p, a = m.inference(adapter.batch(tx, ty))
        res["ncm_pred1"] = p.cpu().numpy()
        params = {k: v.detach().cpu() for k, v in m.network.named_parameters()}
    res["losses"] = np.asarray(losses, np.float64)
    res["preds"] = np.stack(preds)
    names, rows = fx.summarize(params)
    res["param_names"], res["param_rows"] = np.asarray(names), rows
    return res


# ------------------------------------------------------------------------------------ scenario: WA
WA_CFG = dict(arch="cifar_resnet32", feat_dim=64, init=3, bs=8, lr=0.02, momentum=0.9, wd=2e-4)


class RecordingBuffer:
    """stands in for the herding buffer (covered by the iCaRL scenario): records what the plugin asks of it"""

    def __init__(self):
        self.calls = []

    def reduce_old_data(self, task_idx, total):
        self.calls.append(("reduce", int(task_idx), int(total)))

    def update(self, network, train_loader, trfms, task_idx, total, cur_classes, device):
        self.calls.append(("update", int(task_idx), int(total), [int(v) for v in cur_classes]))


class _Trfms:
    trfms = None


def scenario_wa(adapter):
    """task 0: 2 steps of CE; after_task (teacher snapshot, no alignment); task 1 (classes grow by init_cls_num): 2 steps of
    (1-l) CE + l KD(T=2) with the teacher in train mode; after_task aligns the new head rows.  The optimizer is built from
    `get_parameters`, which (reference quirk) leaves the logits head out."""
    c = WA_CFG
    tag = "wa"
    n0, n1 = c["init"], 2 * c["init"]
    P, Bf = fx.det_backbone_state(c["arch"], tag)
    w0, b0 = fx.det_linear(tag + "/head0", n0, c["feat_dim"])
    w1, b1 = fx.det_linear(tag + "/head1", n1, c["feat_dim"])
    w1 = w1 * 1.7                                   # new rows deliberately off-scale so that the alignment matters
    t0 = [fx.det_batch(f"{tag}/t0/{i}", c["bs"], 0, n0) for i in range(2)]
    t1 = [fx.det_batch(f"{tag}/t1/{i}", c["bs"], 0, n1) for i in range(2)]
    ex, ey = fx.det_batch(f"{tag}/eval", c["bs"], 0, n1)
    losses, preds = [], []
    if adapter.kind == "oracle":
        net = om.Net(c["arch"], {k: v.clone().requires_grad_(True) for k, v in P.items()},
                     {k: v.clone() for k, v in Bf.items()}, None, None)
        m = om.WA(net, c["init"])

        def run(batches):
            opt = om.SGD(m.trainable(), c["lr"], c["momentum"], c["wd"])
            for x, y in batches:
                pred, acc, loss = m.observe(x, y, True)
                opt.zero_grad(); loss.backward(); opt.step()
                losses.append(loss.item()); preds.append(pred.numpy())
        m.before_task((w0, b0)); run(t0); m.after_task()
        m.before_task((w1, b1)); run(t1)
        head_before = net.head_w.detach().clone()
        gamma = m.after_task()
        head_after = net.head_w.detach()
        epred, _ = m.inference(ex, ey)
        logits = net.logits(ex, False).detach()
        teacher_rm = m.old.Bf[_last_bn(c["arch"]) + ".running_mean"]
        params = {"backbone." + k: v for k, v in net.P.items()}
        calls = [("reduce", 0, n0), ("update", 0, n0, list(range(0, n0))), ("reduce", 1, n1), ("update", 1, n1, list(range(n0, n1)))]
    else:
        ns = adapter.ns
        bb = adapter.backbone(c["arch"], P, Bf)
        m = ns.WA(bb, c["feat_dim"], 100, device=adapter.device, init_cls_num=c["init"], inc_cls_num=c["init"]).to(adapter.device)
        buffer = RecordingBuffer()
        tl = [ListLoader([], c["bs"], _Trfms())]

        def set_head(w, b, old):
            m.network.to(adapter.device)
            with torch.no_grad():
                m.network.classifier.weight.data[old:] = adapter.to_dev(w[old:])
                m.network.classifier.bias.data[old:] = adapter.to_dev(b[old:])

        def run(batches):
            opt = adapter.sgd_factory(m.get_parameters({}), lr=c["lr"], momentum=c["momentum"], weight_decay=c["wd"])
            m.train()
            for x, y in batches:
                pred, acc, loss = m.observe(adapter.batch(x, y))
                opt.zero_grad(); loss.backward(); opt.step()
                losses.append(float(loss.item())); preds.append(pred.cpu().numpy())
        m.before_task(0, buffer, None, tl); set_head(w0, b0, 0); run(t0); m.after_task(0, buffer, None, tl)
        m.before_task(1, buffer, None, tl); set_head(w1, b1, n0); run(t1)
        head_before = m.network.classifier.weight.detach().cpu().clone()
        m.after_task(1, buffer, None, tl)
        head_after = m.network.classifier.weight.detach().cpu()
        gamma = (head_after[-1].norm() / head_before[-1].norm())
        m.eval()
        with torch.no_grad():
            epred, _ = m.inference(adapter.batch(ex, ey))
            logits = m.network(adapter.to_dev(ex)).cpu()
        teacher_rm = dict(m.old_network.backbone.named_buffers())[_last_bn(c["arch"]) + ".running_mean"].cpu()
        params = {("backbone." + k): v.detach().cpu() for k, v in m.backbone.named_parameters()}
        calls = buffer.calls
    out = dict(losses=np.asarray(losses, np.float64), preds=np.stack(preds), logits_eval=logits.numpy(), eval_pred=np.asarray(epred.cpu()),
               head_before=head_before.numpy(), head_after=head_after.numpy(), gamma=np.float64(float(gamma)),
               teacher_rm=teacher_rm.numpy(), buffer_calls=np.asarray(repr(calls)))
    names, rows = fx.summarize(params)
    out["param_names"], out["param_rows"] = np.asarray(names), rows
    return out


# ----------------------------------------------------------------------------------- scenario: DER
DER_CFG = dict(arch="resnet18", feat_dim=512, init=4, inc=2, bs=4, lr=0.02, momentum=0.9, wd=2e-4)


def scenario_der(adapter):
    """task 0: 1 step (one extractor, CE); task 1: a second extractor copied from the first, `fc` widened over 1024 features,
    fresh aux head; 2 steps of CE + aux CE.  The frozen first extractor keeps running in train mode (its BN running stats
    move), its parameters do not."""
    c = DER_CFG
    tag = "der"
    D, n0, n1 = c["feat_dim"], c["init"], c["init"] + c["inc"]
    P, Bf = fx.det_backbone_state(c["arch"], tag)
    fc0 = fx.det_linear(tag + "/fc0", n0, D)
    aux0 = fx.det_linear(tag + "/aux0", n0 + 1, D)
    fc1 = fx.det_linear(tag + "/fc1", n1, 2 * D)
    aux1 = fx.det_linear(tag + "/aux1", c["inc"] + 1, D)
    t0 = [fx.det_batch(f"{tag}/t0/{i}", c["bs"], 0, n0) for i in range(1)]
    t1 = [fx.det_batch(f"{tag}/t1/{i}", c["bs"], 0, n1) for i in range(2)]       # rehearsal-style: old and new labels
    ex, ey = fx.det_batch(f"{tag}/eval", c["bs"], 0, n1)
    losses, preds = [], []
    last_rm = _last_bn(c["arch"]) + ".running_mean"
    if adapter.kind == "oracle":
        m = om.DER(c["arch"], c["init"], c["inc"])

        def run(batches):
            opt = om.SGD(m.trainable(), c["lr"], c["momentum"], c["wd"])
            for x, y in batches:
                pred, acc, loss = m.observe(x, y, True)
                opt.zero_grad(); loss.backward(); opt.step()
                losses.append(loss.item()); preds.append(pred.numpy())
        m.before_task(0, fc0, aux0, first=(P, Bf)); run(t0)
        first_after_t0 = {k: v.detach().clone() for k, v in m.P[0].items()}
        m.before_task(1, fc1, aux1); run(t1)
        epred, _ = m.inference(ex, ey)
        logits = F_linear(m.features(ex, False), m.fc_w, m.fc_b).detach()
        frozen_same = all(torch.equal(first_after_t0[k], m.P[0][k]) for k in first_after_t0)
        rm0, rm1 = m.Bf[0][last_rm], m.Bf[1][last_rm]
        params = {"convnets.1." + k: v for k, v in m.P[1].items() if not k.startswith("fc.")}
        params.update({"fc.weight": m.fc_w, "fc.bias": m.fc_b, "aux_fc.weight": m.aux_w, "aux_fc.bias": m.aux_b})
    else:
        ns = adapter.ns
        m = ns.DER(None, D, 100, device=adapter.device, init_cls_num=c["init"], inc_cls_num=c["inc"]).to(adapter.device)

        def fill(lin, w, b, keep_rows=0, keep_cols=0):
            with torch.no_grad():
                w, b = adapter.to_dev(w.clone()), adapter.to_dev(b.clone())
                if keep_rows:
                    w[:keep_rows, :keep_cols] = lin.weight.data[:keep_rows, :keep_cols]
                    b[:keep_rows] = lin.bias.data[:keep_rows]
                lin.weight.data.copy_(w); lin.bias.data.copy_(b)

        def run(batches):
            opt = adapter.sgd_factory(m.get_parameters({}), lr=c["lr"], momentum=c["momentum"], weight_decay=c["wd"])
            m.train()
            for x, y in batches:
                pred, acc, loss = m.observe(adapter.batch(x, y))
                opt.zero_grad(); loss.backward(); opt.step()
                losses.append(float(loss.item())); preds.append(pred.cpu().numpy())
        m.before_task(0, None, None, None)
        sd = {k: v.clone() for k, v in P.items()}
        sd.update({k: v.clone() for k, v in Bf.items()})
        m.convnets[0].load_state_dict(sd)
        m.convnets.to(adapter.device)
        fill(m.fc, *fc0); fill(m.aux_fc, *aux0)
        run(t0)
        first_after_t0 = {k: v.detach().cpu().clone() for k, v in m.convnets[0].named_parameters()}
        m.before_task(1, None, None, None)
        fill(m.fc, *fc1, keep_rows=n0, keep_cols=D); fill(m.aux_fc, *aux1)
        run(t1)
        m.eval()
        with torch.no_grad():
            epred, _ = m.inference(adapter.batch(ex, ey))
            logits = m(adapter.to_dev(ex))["logits"].cpu()
        frozen_same = all(torch.equal(first_after_t0[k], v.detach().cpu()) for k, v in m.convnets[0].named_parameters())
        rm0 = dict(m.convnets[0].named_buffers())[last_rm].cpu()
        rm1 = dict(m.convnets[1].named_buffers())[last_rm].cpu()
        params = {"convnets.1." + k: v.detach().cpu() for k, v in m.convnets[1].named_parameters() if not k.startswith("fc.")}
        params.update({"fc.weight": m.fc.weight.detach().cpu(), "fc.bias": m.fc.bias.detach().cpu(),
                       "aux_fc.weight": m.aux_fc.weight.detach().cpu(), "aux_fc.bias": m.aux_fc.bias.detach().cpu()})
    out = dict(losses=np.asarray(losses, np.float64), preds=np.stack(preds), logits_eval=logits.numpy(), eval_pred=np.asarray(epred.cpu()),
               frozen_same=np.asarray(bool(frozen_same)), rm_frozen=rm0.numpy(), rm_new=rm1.numpy())
    names, rows = fx.summarize(params)
    out["param_names"], out["param_rows"] = np.asarray(names), rows
    return out


# ----------------------------------------------------------------------------------- scenario: BiC
BIC_CFG = dict(arch="cifar_resnet32_V2", feat_dim=256, size=64, init=3, inc=3, task_num=3, num_class=9, bs=6, lr=0.02, momentum=0.9,
               buffer_size=20)


class _ListDataset(torch.utils.data.Dataset):
    """what `spilt_and_update` needs of `dataloader.dataset`: `images` / `labels` lists, deep-copyable, sized"""

    def __init__(self, images, labels):
        self.images, self.labels = list(images), list(labels)

    def __len__(self):
        return len(self.labels)

    def __getitem__(self, i):
        return {"image": torch.zeros(1), "label": int(self.labels[i])}


class _HasDataset:
    def __init__(self, ds):
        self.dataset = ds


def _ints(v):
    return np.asarray([int(t) for t in v], np.int64)


def _bic_task_lists(t):
    """(image ids, labels) of task t in a fixed shuffled order; id = 1000 * class + k"""
    c = BIC_CFG
    imgs, labs = [], []
    for j, n in enumerate({0: [12, 12, 12], 1: [10, 12, 14]}[t]):
        cls = (0 if t == 0 else c["init"]) + j
        imgs += [cls * 1000 + k for k in range(n)]
        labs += [cls] * n
    order = np.random.RandomState(11 + t).permutation(len(labs))
    return [imgs[i] for i in order], [labs[i] for i in order]


def bic_split_plugin(m, ns):
    """`spilt_and_update` of a plugin-surface `bic` object for two tasks under a seeded numpy RNG -> {task: record}"""
    c = BIC_CFG
    buf = ns.LinearSpiltBuffer(c["buffer_size"], "balance_random", 4, 0.1)
    cfg = {"buffer": {"kwargs": {"buffer_size": c["buffer_size"]}}, "batch_size": 4, "num_workers": 0, "init_cls_num": c["init"],
           "inc_cls_num": c["inc"], "gpu_input_pipeline": False}
    m.cls_count = {}
    split = {}
    for t in (0, 1):
        np.random.seed(7 + t)
        imgs, labs = _bic_task_lists(t)
        tr, va = m.spilt_and_update(_HasDataset(_ListDataset(imgs, labs)), buf, t, cfg)
        assert tr.drop_last and tr.batch_size == 4 and (va is None) == (t == 0) and (va is None or (va.batch_size == 100 and not va.drop_last))
        split[t] = (tr.dataset.images, tr.dataset.labels, va.dataset.images if va is not None else None, va.dataset.labels if va is not None else None,
                    list(buf.train_images), list(buf.train_labels), list(buf.val_images), list(buf.val_labels))
    assert buf.total_classes == c["init"] + c["inc"]
    return split


BIC_SPLIT_FIELDS = ("train_i", "train_l", "val_i", "val_l", "buf_ti", "buf_tl", "buf_vi", "buf_vl")


def scenario_bic(adapter):
    """ResNet_BIC(32) on 64 x 64 inputs.  Task 0: 2 steps of CE (weight decay 2e-4 * task_num / 1, core/trainer.py:299-300);
    after_task; inference.  Task 1: 2 distillation steps against the bias-corrected previous model (its BatchNorm in train mode),
    after_task, 3 stage-2 steps of the task's bias layer under the plugin's Adam with everything in eval mode, inference.
    Separately: `spilt_and_update` for two tasks under a seeded numpy RNG (split, loaders' datasets, re-cut buffer)."""
    c = BIC_CFG
    tag = "bic"
    P, Bf = fx.det_backbone_state(c["arch"], tag)
    w, b = fx.det_linear(tag + "/head", c["num_class"], c["feat_dim"])

    def batch(name, lo, hi):
        return fx.det_images(f"{tag}/{name}/x", c["bs"], size=c["size"]), torch.from_numpy(detrand.randint(f"{tag}/{name}/y", (c["bs"],), lo, hi))
    seen1 = c["init"] + c["inc"]
    t0 = [batch(f"t0/{i}", 0, c["init"]) for i in range(2)]
    t1 = [batch(f"t1/{i}", 0, seen1) for i in range(2)]
    v1 = [batch(f"v1/{i}", 0, seen1) for i in range(3)]
    probe = batch("probe", 0, seen1)
    losses, losses2, preds, infer = [], [], [], []

    def wd_of(t):
        return 2e-4 * c["task_num"] / (t + 1)
    # ---- the data side, on plain lists
    split = {}
    task_lists = _bic_task_lists

    if adapter.kind == "oracle":
        net = om.Net(c["arch"], {k: v.clone().requires_grad_(True) for k, v in P.items()}, {k: v.clone() for k, v in Bf.items()},
                     w.clone().requires_grad_(True), b.clone().requires_grad_(True))
        m = om.BiC(net, c["init"], c["inc"], c["task_num"])

        def run(batches, t):
            opt = om.SGD(net.parameters(), c["lr"], c["momentum"], wd_of(t))
            for x, y in batches:
                pred, acc, loss = m.observe(x, y, True)
                opt.zero_grad(); loss.backward(); opt.step()
                losses.append(loss.item()); preds.append(pred.numpy())
        m.before_task(0); run(t0, 0); m.after_task(0)
        infer.append(m.inference(*probe)[0].numpy())
        m.before_task(1); run(t1, 1); m.after_task(1)
        teacher_rm = m.old.Bf["bn.running_mean"].clone()
        for x, y in v1:
            pred, acc, loss = m.stage2(x, y)
            losses2.append(loss.item()); preds.append(pred.numpy())
        infer.append(m.inference(*probe)[0].numpy())
        bias = torch.stack([torch.cat([m.alphas[i].detach(), m.betas[i].detach()]) for i in range(c["task_num"])])
        params = dict(net.named_parameters())
        import types
        buf = types.SimpleNamespace(train_images=[], train_labels=[], val_images=[], val_labels=[], total_classes=0)
        m2 = om.BiC(net, c["init"], c["inc"], c["task_num"])
        for t in (0, 1):
            np.random.seed(7 + t)
            imgs, labs = task_lists(t)
            ti, tl, vi, vl = m2.split_and_update(imgs, labs, buf, t, c["buffer_size"])
            split[t] = (ti, tl, vi, vl, list(buf.train_images), list(buf.train_labels), list(buf.val_images), list(buf.val_labels))
    else:
        ns = adapter.ns
        bb = adapter.backbone(c["arch"], P, Bf)
        m = ns.bic(bb, c["num_class"], device=adapter.device, task_num=c["task_num"], init_cls_num=c["init"], inc_cls_num=c["inc"]).to(adapter.device)
        with torch.no_grad():
            m.model.classifier.weight.copy_(adapter.to_dev(w)); m.model.classifier.bias.copy_(adapter.to_dev(b))

        def run(batches, t):
            opt = adapter.sgd_factory(m.get_parameters({}), lr=c["lr"], momentum=c["momentum"], weight_decay=wd_of(t))
            m.train()
            for layer in m.bias_layers:
                layer.eval()
            for x, y in batches:
                pred, acc, loss = m.observe(adapter.batch(x, y))
                opt.zero_grad(); loss.backward(retain_graph=True); opt.step()
                losses.append(float(loss.item())); preds.append(pred.cpu().numpy())

        def probe_now():
            m.eval()
            with torch.no_grad():
                infer.append(m.inference(adapter.batch(*probe))[0].cpu().numpy())
        m.before_task(0, None, None, None); run(t0, 0); m.after_task(0, None, None, None); probe_now()
        m.before_task(1, None, None, None); run(t1, 1); m.after_task(1, None, None, None)
        teacher_rm = dict(m.previous_model.named_buffers())["backbone.bn.running_mean"].detach().cpu()
        m.eval()
        for layer in m.bias_layers:
            layer.train()
        for x, y in v1:
            pred, acc, loss = m.stage2(adapter.batch(x, y))
            losses2.append(float(loss.item())); preds.append(pred.cpu().numpy())
        probe_now()
        bias = torch.stack([torch.cat([layer.alpha.detach().cpu(), layer.beta.detach().cpu()]) for layer in m.bias_layers])
        params = {k: v.detach().cpu() for k, v in m.model.named_parameters()}
        split = bic_split_plugin(m, ns)
    out = dict(losses=np.asarray(losses, np.float64), losses2=np.asarray(losses2, np.float64), preds=np.stack(preds), infer=np.stack(infer),
               bias=bias.double().numpy(), teacher_rm=teacher_rm.double().numpy())
    names, rows = fx.summarize(params)
    out["param_names"], out["param_rows"] = np.asarray(names), rows
    for t, rec in split.items():
        for key, v in zip(BIC_SPLIT_FIELDS, rec):
            if v is not None:
                out[f"split{t}_{key}"] = _ints(v)
    return out


def F_linear(x, w, b):
    return torch.nn.functional.linear(x, w, b)


# --------------------------------------------------------------------------------- scenario: LUCIR
LUCIR_CFG = dict(arch="resnet32_V2", feat_dim=64, init=6, inc=2, bs=8, lr=0.02, momentum=0.9, wd=5e-4,
                 lamda=5.0, K=2, lw_mr=1.0, dist=0.5, per_class=10)


def scenario_lucir(adapter):
    """task 0: 1 step (CE on the cosine head); before_task(1): teacher snapshot + fc2 imprint from
    class-mean features; task 1: 2 steps with less-forget + CE + margin-ranking (batches mix old and new
    labels), param groups per lucir.py:229-236."""
    c = LUCIR_CFG
    tag = "lucir"
    P, Bf = fx.det_backbone_state(c["arch"], tag)
    b = 1.0 / np.sqrt(c["feat_dim"])
    w0 = fx._t(detrand.uniform(tag + "/w0", (c["init"], c["feat_dim"]), -b, b))
    t0 = [fx.det_batch(f"{tag}/t0/0", c["bs"], 0, c["init"])]
    new_classes = list(range(c["init"], c["init"] + c["inc"]))
    cls_x = {cc: fx.class_images(tag, cc, c["per_class"]) for cc in new_classes}
    t1 = [fx.det_batch(f"{tag}/t1/{i}", c["bs"], 0, c["init"] + c["inc"]) for i in range(2)]
    losses, preds = [], []
    res = {}
    if adapter.kind == "oracle":
        Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        Bb = {k: v.clone() for k, v in Bf.items()}
        m = om.LUCIR(c["arch"], Pg, Bb, w0.clone().requires_grad_(True), torch.ones(1, requires_grad=True),
                     c["init"], c["inc"], c["lamda"], c["K"], c["lw_mr"], c["dist"])

        def run(batches, groups):
            opts = [om.SGD(ps, **kw) for ps, kw in groups]
            for x, y in batches:
                pred, acc, loss = m.observe(x, y, True)
                for o in opts: o.zero_grad()
                loss.backward()
                for o in opts: o.step()
                losses.append(loss.item()); preds.append(pred.numpy())
        m.before_task(0)
        run(t0, [(m.parameters(), dict(lr=c["lr"], momentum=c["momentum"], weight_decay=c["wd"]))])
        with torch.no_grad():
            cf = [nets.forward(c["arch"], Pg, Bb, cls_x[cc], False) for cc in new_classes]
        m.before_task(1, class_feats=cf)
        base = list(m.P.values()) + [m.fc2_w, m.sigma]
        run(t1, [(base, dict(lr=0.1, momentum=c["momentum"], weight_decay=5e-4)),
                 ([m.fc1_w], dict(lr=0.0, momentum=c["momentum"], weight_decay=0.0))])
        res["fc2_imprint_norm"] = np.asarray([m.cur_lamda])
        params = {("backbone." + k): v for k, v in m.P.items()}
        params["classifier.fc1.weight"] = m.fc1_w; params["classifier.fc2.weight"] = m.fc2_w
        params["classifier.sigma"] = m.sigma
    else:
        ns = adapter.ns
        bb = adapter.backbone(c["arch"], P, Bf)
        m = ns.LUCIR(bb, c["feat_dim"], 100, device=adapter.device, init_cls_num=c["init"], inc_cls_num=c["inc"],
                     lamda=c["lamda"], K=c["K"], lw_mr=c["lw_mr"], dist=c["dist"]).to(adapter.device)
        with torch.no_grad():
            m.network.classifier.weight.data.copy_(adapter.to_dev(w0))

        def run(batches):
            opt = adapter.sgd_factory(m.get_parameters({}), lr=c["lr"], momentum=c["momentum"], weight_decay=c["wd"])
            m.train()
            for x, y in batches:
                pred, acc, loss = m.observe(adapter.batch(x, y))
                opt.zero_grad(); loss.backward(); opt.step()
                losses.append(float(loss.item())); preds.append(pred.cpu().numpy())
        m.before_task(0, None, None, None)
        run(t0)

        class _DS(torch.utils.data.Dataset):
            def __init__(s):
                s.images = [(cc, k) for cc in new_classes for k in range(c["per_class"])]
                s.labels = [cc for cc in new_classes for k in range(c["per_class"])]
            def __len__(s): return len(s.labels)
            def __getitem__(s, i):
                im = s.images[i]
                return {"image": cls_x[int(im[0])][int(im[1])], "label": int(s.labels[i])}
        m.before_task(1, None, ListLoader([], c["bs"], _DS()), None)
        res["fc2_imprint_norm"] = np.asarray([m.cur_lamda])
        run(t1)
        m.after_task(1, None, None, None)
        params = {k: v.detach().cpu() for k, v in m.network.named_parameters()}
    res["losses"] = np.asarray(losses, np.float64)
    res["preds"] = np.stack(preds)
    names, rows = fx.summarize(params)
    res["param_names"], res["param_rows"] = np.asarray(names), rows
    res["fc2_w"] = params["classifier.fc2.weight"].detach().numpy()
    return res
