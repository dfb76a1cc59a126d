#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (read-only at /root/reference) on the CPU.

Runs only in the build container (the reference cannot travel to the GPU box).  The reference has no tests,
golden vectors or fixtures of its own, so these files are the pin for oracle/icaf_oracle.py:
    python tests/golden/make_golden.py            # rewrites every fixture

What is recorded per model case: the full Detect output z, class logits, and 2048 seeded samples of every layer's
output (enough to localise a divergence to one layer without shipping MBs of activations).  Weights and inputs
are NOT stored: they are regenerated from icafusion_amd.synth (numpy PCG64 streams keyed by state_dict name).

Third-party modules that are missing here and unused on the forward path (cv2, timm, torchvision, seaborn, thop)
are stubbed; torchvision.ops.nms — used by the reference's non_max_suppression (utils/general.py:591) — is
stubbed with the oracle's greedy core, so the NMS fixtures pin the reference's wrapper logic (candidate
filtering, multi-label expansion, class offsets, max_det) but NOT the greedy core itself ("parity unpinned").
"""
import os
import sys
import types

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

import numpy as np
import torch
import yaml

sys.path.insert(0, REPO)
from icafusion_amd.synth import synth_state_dict, synth_images, synth_tensor      # noqa: E402
from oracle import icaf_oracle as oracle                                             # noqa: E402


class _Dummy:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return None

    def __getattr__(self, k):
        return _Dummy()


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _tv_nms(boxes, scores, thr):
    return torch.from_numpy(oracle.nms_greedy(boxes.numpy(), scores.numpy(), float(thr)))


def import_reference():
    _stub("cv2", setNumThreads=lambda *a: None, ocl=_Dummy())
    _stub("seaborn")
    _stub("thop")
    _stub("timm"); _stub("timm.models"); _stub("timm.models.layers", DropPath=_Dummy)
    tv = _stub("torchvision")
    tv.ops = _stub("torchvision.ops", nms=_tv_nms)
    tv.transforms = _stub("torchvision.transforms")
    tv.utils = _stub("torchvision.utils", save_image=None)
    tv.models = _stub("torchvision.models")
    # the reference's `models` / `utils` packages must win over this repo's drop-in packages of the same name
    sys.path.insert(0, REF)
    for k in [k for k in sys.modules if k == "models" or k.startswith("models.") or k == "utils" or k.startswith("utils.")]:
        del sys.modules[k]
    import models.yolo_test as yt          # noqa
    import models.common as common         # noqa
    import utils.general as general        # noqa
    import utils.metrics as metrics        # noqa
    assert yt.__file__.startswith(REF) and general.__file__.startswith(REF)
    return yt, common, general, metrics


def sample_idx(numel, tag, n=2048):
    g = np.random.default_rng([0x5A3917, tag])
    return g.integers(0, numel, size=min(n, numel))


def model_case(yt, name, yaml_name, batch, h, w, seed, loops=None):
    ref_cfg = os.path.join(REF, "models", "transformer", yaml_name)
    ours = yaml.safe_load(open(os.path.join(REPO, "models", "transformer", yaml_name)))
    assert ours == yaml.safe_load(open(ref_cfg)), "config surface drifted from the reference"
    model = yt.Model(ref_cfg).eval()
    model.load_state_dict(synth_state_dict(model, seed))
    if loops is not None:
        for i in (20, 21, 22):
            model.model[i].crosstransformer[0].loops = loops          # SURVEY §0.2 / models/common.py:744
    rgb, ir = synth_images(batch, h, w, seed)
    rec = {}
    hooks = []
    for i, layer in enumerate(model.model):
        def hook(mod, inp, out, i=i):
            if torch.is_tensor(out):
                flat = out.detach().reshape(-1)
                rec[f"layer{i}"] = flat[torch.from_numpy(sample_idx(flat.numel(), i))].numpy().copy()
                rec[f"layer{i}_shape"] = np.asarray(out.shape)
        hooks.append(layer.register_forward_hook(hook))
    with torch.no_grad():
        z, logits, raws = model(rgb, ir)
    for hk in hooks:
        hk.remove()
    rec.update(z=z.numpy(), logits=logits.numpy())
    for l, r in enumerate(raws):
        flat = r.reshape(-1)
        rec[f"raw{l}"] = flat[torch.from_numpy(sample_idx(flat.numel(), 100 + l))].numpy().copy()
        rec[f"raw{l}_shape"] = np.asarray(r.shape)
    rec["meta"] = np.asarray([batch, h, w, seed, -1 if loops is None else loops])
    rec["yaml"] = np.asarray(yaml_name)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **rec)
    print(name, "z", tuple(z.shape), "saved")
    return z


def dmff_case(common, name, c, va, ha, batch, h, w, seed, loops):
    blk = common.TransformerFusionBlock(c, va, ha).eval()
    for m in blk.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eps = 1e-3                                              # what initialize_weights does in Model()
    sd = {k: (v if k.endswith("num_batches_tracked") else synth_tensor("model.20." + k, v.shape, seed=seed))
          for k, v in blk.state_dict().items()}
    blk.load_state_dict(sd)
    blk.crosstransformer[0].loops = loops
    g = np.random.default_rng([seed, 77, c, h, w])
    rgb = torch.from_numpy(g.normal(0, 1, (batch, c, h, w)).astype(np.float32))
    ir = torch.from_numpy(g.normal(0, 1, (batch, c, h, w)).astype(np.float32))
    with torch.no_grad():
        out = blk([rgb, ir])
        # also record the token tensors that enter / leave the cross transformer (layer-local checkpoints)
        tv = blk.vis_coefficient(blk.avgpool(rgb), blk.maxpool(rgb))
        tok_in = tv.contiguous().view(batch, c, -1).permute(0, 2, 1) + blk.pos_emb_vis
        ti = blk.ir_coefficient(blk.avgpool(ir), blk.maxpool(ir))
        tok_in_ir = ti.contiguous().view(batch, c, -1).permute(0, 2, 1) + blk.pos_emb_ir
        tok_out, tok_out_ir = blk.crosstransformer([tok_in, tok_in_ir])
        att_v, att_i = blk.crosstransformer[0].crossatt([tok_in, tok_in_ir])
    rec = {}
    for j, (k, t) in enumerate([("out", out), ("tok_in", tok_in), ("tok_in_ir", tok_in_ir), ("tok_out", tok_out),
                                ("tok_out_ir", tok_out_ir), ("att_v", att_v), ("att_i", att_i)]):
        flat = t.reshape(-1)
        rec[k] = flat[torch.from_numpy(sample_idx(flat.numel(), 200 + j, 8192))].numpy().copy()
        rec[k + "_shape"] = np.asarray(t.shape)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), meta=np.asarray([c, va, ha, batch, h, w, seed, loops]),
                        **rec)
    print(name, tuple(out.shape), "saved")


def nms_case(general, name, z, **kw):
    pred = torch.from_numpy(z.copy())
    out = general.non_max_suppression(pred, **kw)
    rec = {f"det{i}": o.numpy() for i, o in enumerate(out)}
    rec["n"] = np.asarray(len(out))
    rec["kw"] = np.asarray(repr(sorted(kw.items())))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **rec)
    print(name, [tuple(o.shape) for o in out])


def metrics_case(general, metrics, name):
    g = np.random.default_rng(42)
    n, nc = 400, 3
    tp = g.random((n, 10)) < np.linspace(0.7, 0.1, 10)[None]
    tp = np.logical_and.accumulate(tp, 1)                    # a hit at IoU t implies hits at lower thresholds
    conf = g.random(n).astype(np.float32)
    pcls = g.integers(0, nc, n).astype(np.float32)
    tcls = g.integers(0, nc, 150).astype(np.float32)
    r = metrics.ap_per_class(tp, conf, pcls, tcls)
    a = torch.from_numpy(g.uniform(0, 100, (7, 2)).astype(np.float32))
    b = torch.from_numpy(g.uniform(0, 100, (5, 2)).astype(np.float32))
    box1 = torch.cat((a, a + torch.from_numpy(g.uniform(5, 60, (7, 2)).astype(np.float32))), 1)
    box2 = torch.cat((b, b + torch.from_numpy(g.uniform(5, 60, (5, 2)).astype(np.float32))), 1)
    iou = general.box_iou(box1, box2).numpy()
    coords = torch.from_numpy(g.uniform(-20, 700, (9, 4)).astype(np.float32))
    scaled = general.scale_coords((512, 640), coords.clone(), (480, 720)).numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), tp=tp, conf=conf, pcls=pcls, tcls=tcls, ap=r[5],
                        p=r[3], r=r[4], classes=r[7], box1=box1.numpy(), box2=box2.numpy(), iou=iou,
                        coords=coords.numpy(), scaled=scaled)
    print(name, "ap50", r[5][:, 0])


def match_case(general, name):
    """Pin the TP-matching statement: the reference has it INLINE in its validation loop (test.py:196-230), so those source
    lines are read from the reference tree at generation time, dedented and exec'ed on synthetic detections / labels with
    the reference's own scale_coords / xywh2xyxy / box_iou in scope.  Nothing of the block is stored — only its outputs."""
    import textwrap
    with open(os.path.join(REF, "test.py")) as f:
        lines = f.read().splitlines()
    i0 = next(i for i, l in enumerate(lines) if "# Assign all predictions as incorrect" in l)
    i1 = next(i for i, l in enumerate(lines) if i > i0 and "# Append statistics (correct, conf, pcls, tcls)" in l)
    assert (i0 + 1, i1 + 1) == (196, 229), (i0, i1)                         # the block SURVEY.md / DESIGN.md cite as test.py:196-230
    block = textwrap.dedent("\n".join(lines[i0:i1])).rstrip()
    assert block.lstrip().startswith("# Assign all predictions as incorrect") and block.rstrip().endswith("break"), block
    code = compile(block, "reference test.py:196-230", "exec")
    g = np.random.default_rng(77)
    iouv = torch.linspace(0.5, 0.95, 10)
    rec, T = {}, 24
    for t in range(T):
        H, W = [(320, 320), (544, 672), (160, 128)][t % 3]                  # letterboxed batch shape
        h0, w0 = [(300, 320), (512, 640), (150, 100)][t % 3]                # native image
        gain = min(H / h0, W / w0)
        shapes = [((h0, w0), ((gain, gain), ((W - w0 * gain) / 2, (H - h0 * gain) / 2)))]
        m, n = int(g.integers(0, 7)), int(g.integers(0, 60))
        if t == 0:
            m, n = 0, 5
        if t == 1:
            m, n = 3, 0
        cls = g.integers(0, 3, (m, 1)).astype(np.float32)
        cxy = g.uniform(0.25, 0.75, (m, 2)) * [W, H]
        wh = g.uniform(0.05, 0.3, (m, 2)) * [W, H]
        labels = torch.from_numpy(np.concatenate((cls, cxy, wh), 1).astype(np.float32))          # [cls, x, y, w, h] letterboxed pixels
        if m and n:
            src = labels[g.integers(0, m, n)]
            xyxy = general.xywh2xyxy(src[:, 1:5]) + torch.from_numpy(g.normal(0, 0.04, (n, 4)).astype(np.float32)) * src[:, [3, 4, 3, 4]]
            pcls = torch.where(torch.from_numpy(g.random(n) < 0.8), src[:, 0], torch.from_numpy(g.integers(0, 3, n).astype(np.float32)))
        else:
            a = torch.from_numpy(g.uniform(0, 100, (n, 2)).astype(np.float32))
            xyxy = torch.cat((a, a + 40), 1)
            pcls = torch.from_numpy(g.integers(0, 3, n).astype(np.float32))
        conf = torch.from_numpy(np.sort(g.random(n).astype(np.float32))[::-1].copy())              # NMS output order: descending conf
        pred = torch.cat((xyxy, conf[:, None], pcls[:, None]), 1)
        if n > 4:
            pred[3, :4] = pred[2, :4]                                           # two detections with identical IoU to a label
        img = torch.zeros(1, 6, H, W)
        predn = pred.clone()
        general.scale_coords(img[0].shape[1:], predn[:, :4], shapes[0][0], shapes[0][1])          # test.py:160-161
        ns = dict(torch=torch, pred=pred, predn=predn, labels=labels, nl=len(labels), niou=10, iouv=iouv, device="cpu", img=img,
                  si=0, shapes=shapes, plots=False, scale_coords=general.scale_coords, xywh2xyxy=general.xywh2xyxy,
                  box_iou=general.box_iou, confusion_matrix=None)
        exec(code, ns)
        tbox = general.xywh2xyxy(labels[:, 1:5])
        general.scale_coords(img[0].shape[1:], tbox, shapes[0][0], shapes[0][1])
        rec[f"pred{t}"], rec[f"predn{t}"], rec[f"labels{t}"] = pred.numpy(), predn.numpy(), labels.numpy()
        rec[f"tbox{t}"], rec[f"correct{t}"] = tbox.numpy(), ns["correct"].numpy()
        rec[f"geom{t}"] = np.asarray([H, W, h0, w0, gain, shapes[0][1][1][0], shapes[0][1][1][1]], np.float64)
    rec["n"], rec["iouv"] = np.asarray(T), iouv.numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **rec)
    print(name, "TP flags per trial:", [int(rec[f"correct{t}"][:, 0].sum()) for t in range(T)])


def results_case(general, name):
    """Pin the result-file formats (--save-txt / --save-json): the reference writes them INLINE in its validation loop
    (test.py:162-171 per-image text lines, :184-195 JSON rows, :248-258 result.txt), so those source lines are read from the reference
    tree at generation time, dedented and exec'ed on the detections of tests/golden/match_predictions.npz (one trial = one image) in a
    temporary directory, with the reference's own xyxy2xywh / xyxy2xywh2 in scope.  Only the produced file contents are stored."""
    import json
    import tempfile
    import textwrap
    from pathlib import Path
    with open(os.path.join(REF, "test.py")) as f:
        lines = f.read().splitlines()

    def block(first, last_exclusive, want_range):
        i0 = next(i for i, l in enumerate(lines) if first in l)
        i1 = next(i for i, l in enumerate(lines) if i > i0 and last_exclusive in l)
        assert (i0 + 1, i1) == want_range, (first, i0 + 1, i1)
        return compile(textwrap.dedent("\n".join(lines[i0:i1])).rstrip(), f"reference test.py:{i0 + 1}-{i1}", "exec")
    txt_code = block("# Append to text file", "# W&B logging - Media Panel Plots", (162, 171))
    json_code = block("# Append to pycocotools JSON dictionary", "# Assign all predictions as incorrect", (184, 195))
    i0 = next(i for i, l in enumerate(lines) if "temp = []" in l) - 1
    assert lines[i0].strip() == "if save_txt:" and i0 + 1 == 248, (i0, lines[i0])
    merge_code = compile(textwrap.dedent("\n".join(lines[i0:i0 + 11])).rstrip(), "reference test.py:248-258", "exec")
    assert "ff.write(ii)" in lines[i0 + 10], lines[i0 + 10]
    g = np.load(os.path.join(HERE, "match_predictions.npz"))
    T = int(g["n"])
    stems = [(f"set{t:02d}_V000_I{t * 37:05d}" if t % 2 == 0 else f"{1000 + t * 13}") for t in range(T)]
    labels_list = sorted([s + ".txt" for s in stems] + ["0000.txt", "zz_extra.txt"])
    out = {"stems": stems, "labels_list": labels_list, "runs": {}}
    for save_conf in (True, False):
        with tempfile.TemporaryDirectory() as d:
            labels_dir = Path(d) / "labels"
            labels_dir.mkdir()
            jdict = []
            for t in range(T):
                pred, predn = torch.from_numpy(g[f"pred{t}"]), torch.from_numpy(g[f"predn{t}"])
                geom = g[f"geom{t}"]
                shapes = [((int(geom[2]), int(geom[3])), None)]
                ns = dict(torch=torch, save_txt=True, save_json=True, save_conf=save_conf, labels_list=labels_list, path=Path("/data/visible/test") / (stems[t] + ".jpg"),
                          shapes=shapes, si=0, predn=predn, pred=pred, labels_dir=labels_dir, jdict=jdict, is_coco=False, coco91class=None,
                          xyxy2xywh=general.xyxy2xywh, xyxy2xywh2=general.xyxy2xywh2)
                exec(txt_code, ns)
                exec(json_code, ns)
            exec(merge_code, dict(save_txt=True, os=os, labels_dir=labels_dir))
            files = {n: open(labels_dir / n).read() for n in sorted(os.listdir(labels_dir))}
            out["runs"]["conf" if save_conf else "noconf"] = {"files": files, "jdict": jdict}
    with open(os.path.join(HERE, name + ".json"), "w") as f:
        json.dump(out, f)
    print(name, "files:", len(out["runs"]["conf"]["files"]), "json rows:", len(out["runs"]["conf"]["jdict"]))


def pool_window_case(common, name):
    """Pin AdaptivePool2d's window rule (models/common.py:868-891) over a sweep of feature-map sizes, not just the four DMFF fixtures:
    for the three anchor grids of the shipped yamls and ~1,200 (h, w) each — every height from the grid size to 170, eight scattered
    widths per height, sizes below the grid in one dimension included where the reference still runs — the reference module's output
    shape and the float64 sum of its avg / max outputs on a seeded tensor."""
    rows = []
    for va, ha in ((20, 20), (16, 16), (10, 10)):
        avg, mx = common.AdaptivePool2d(va, ha, "avg"), common.AdaptivePool2d(va, ha, "max")
        sizes = [(h, ha - 3 + ((h * 7 + j * 13) % (174 - ha))) for h in range(va - 3, 171) for j in range(8)]
        sizes += [(va, ha), (va - 1, ha), (va, ha - 2), (va - 3, ha - 3), (va // 2, ha // 2)]     # at or below the grid on both sides: identity
        for h, w in sizes:
            g = np.random.default_rng([va, h, w])
            x = torch.from_numpy(g.normal(0, 1, (1, 2, h, w)).astype(np.float32))
            try:
                a, m = avg(x), mx(x)
            except Exception:                           # one side above the grid and the other below it: stride 0, torch raises
                rows.append((va, ha, h, w, -1, -1, 0.0, 0.0))
                continue
            assert a.shape == m.shape
            rows.append((va, ha, h, w, a.shape[2], a.shape[3], float(a.double().sum()), float(m.double().sum())))
    rows = np.asarray(rows, np.float64)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), rows=rows)
    print(name, rows.shape, "raising:", int((rows[:, 4] < 0).sum()))


DATASET_CASES = [dict(n=7, size=(96, 128), nc=3, seed=9, img_size=160, batch=2, mixed=True),         # both orientations, ragged last batch
                 dict(n=10, size=(100, 150), nc=2, seed=10, img_size=320, batch=4, mixed=True),
                 dict(n=5, size=(128, 160), nc=1, seed=11, img_size=640, batch=8, mixed=False)]       # KAIST-like 4:5 frames, one batch: 544x672 rule


def dataset_case(name):
    """Pin the paired validation set's metadata path: the reference's LoadMultiModalImagesAndLabels.__init__ (utils/datasets.py:690-880:
    file discovery, visible -> labels rule, label parsing, aspect-ratio sort, rectangular batch shapes with pad 0.5) is run on synthetic
    paired folders written by tests/test_frontends.py::make_dataset; stored: file order, native shapes, batch shapes, labels.  Version
    shims only (`np.int`, `torch.load(weights_only=False)` for the label cache the class writes next to the labels); __getitem__ needs
    OpenCV and is not touched."""
    import functools
    import tempfile
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from test_frontends import make_dataset
    import utils.datasets as rd
    assert rd.__file__.startswith(REF)
    if not hasattr(np, "int"):
        np.int = int
    real_load = torch.load
    rd.torch.load = functools.partial(real_load, weights_only=False)
    rec = {}
    try:
        for k, c in enumerate(DATASET_CASES):
            with tempfile.TemporaryDirectory() as d:
                rgb_dir, ir_dir = make_dataset(d, n=c["n"], size=c["size"], nc=c["nc"], seed=c["seed"], mixed=c["mixed"])
                ds = rd.LoadMultiModalImagesAndLabels(rgb_dir, ir_dir, c["img_size"], c["batch"], augment=False, hyp=None, rect=True,
                                                      cache_images=False, single_cls=False, stride=32, pad=0.5, image_weights=False, prefix="")
                assert [os.path.basename(f) for f in ds.img_files_rgb] == [os.path.basename(f) for f in ds.img_files_ir]
                rec[f"files{k}"] = np.asarray([os.path.basename(f) for f in ds.img_files_rgb])
                rec[f"label_files{k}"] = np.asarray([os.path.relpath(f, d) for f in ds.label_files_rgb])
                rec[f"shapes{k}"] = np.asarray(ds.shapes_rgb)                      # (w, h) per image, sorted order
                rec[f"batch_shapes{k}"] = np.asarray(ds.batch_shapes_rgb)          # (h, w) per batch
                rec[f"batch{k}"] = np.asarray(ds.batch_rgb)
                rec[f"labels{k}"] = np.concatenate([np.concatenate((np.full((len(l), 1), i, np.float32), l), 1) for i, l in enumerate(ds.labels_rgb)])
                print(name, k, rec[f"files{k}"].tolist(), rec[f"batch_shapes{k}"].tolist())
    finally:
        rd.torch.load = real_load
    import json
    rec["cases"] = np.asarray(json.dumps(DATASET_CASES))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **rec)


def half_case(yt, name):
    """The reference's own 16-bit behaviour (detect_twostream.py:33-40 / test.py:73-75: `.fuse()` then `.half()`): the real reference model,
    fused, cast to bf16 / fp16 and run on the CPU on the seeded inputs of two fixtures, next to its fp32 output.  Stored: the 16-bit
    outputs and their deviation from fp32 — the yardstick the 16-bit parity bounds of the HIP path are derived from (the 16-bit mode of
    the oracle, `OracleModel(dtype=)`, is checked against these numbers in tests/test_oracle_vs_golden.py)."""
    rec = {}
    cases = [("s", "yolov5s_Transfusion_kaist.yaml", 1, 320, 320, 1), ("l", "yolov5l_Transfusion_VEDAI.yaml", 1, 320, 320, 3)]
    for tag, yaml_name, batch, h, w, seed in cases:
        ref_cfg = os.path.join(REF, "models", "transformer", yaml_name)
        rgb, ir = synth_images(batch, h, w, seed)
        for dn, dt in (("bf16", torch.bfloat16), ("f16", torch.float16)):
            model = yt.Model(ref_cfg).eval()
            model.load_state_dict(synth_state_dict(model, seed))
            model = model.fuse().eval()
            with torch.no_grad():
                z32 = model(rgb, ir)[0]
                z16 = model.to(dt)(rgb.to(dt), ir.to(dt))[0].float()
            d = (z16 - z32).abs()
            rec[f"{tag}_{dn}_z16"] = z16.numpy()
            rec[f"{tag}_{dn}_dev"] = np.asarray([d[..., :4].max(), d[..., :4].mean(), d[..., 4:].max(), d[..., 4:].mean()], np.float64)
            rec[f"{tag}_z32"] = z32.numpy()
            print(name, tag, dn, "box max / mean, conf max / mean:", rec[f"{tag}_{dn}_dev"].tolist())
        rec[f"{tag}_meta"] = np.asarray([batch, h, w, seed])
        rec[f"{tag}_yaml"] = np.asarray(yaml_name)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **rec)


def main():
    torch.set_num_threads(os.cpu_count())
    yt, common, general, metrics = import_reference()
    if "--match-only" in sys.argv:                    # TP matching of the validation loop (test.py:196-230), added in round 2
        match_case(general, "match_predictions")
        return
    if "--m-only" in sys.argv:                        # yolov5m widths (48 / 96 / 192 / 384 / 768: no powers of two), added later
        model_case(yt, "model_m_kaist_320_b1", "yolov5m_Transfusion_kaist.yaml", 1, 320, 320, seed=13)
        return
    if "--n-only" in sys.argv:                        # yolov5n + DMFF (16-channel stem, C = 64 / 128 / 256 fusion blocks), added later
        model_case(yt, "model_n_flir_352x320_b2", "yolov5n_Transfusion_FLIR.yaml", 2, 352, 320, seed=14)
        return
    if "--half-only" in sys.argv:                     # the real reference run in bf16 / fp16 on the CPU
        half_case(yt, "reference_16bit")
        return
    if "--dataset-only" in sys.argv:                  # rect validation set metadata from the reference's dataset class
        dataset_case("rect_dataset")
        return
    if "--pool-only" in sys.argv:                     # AdaptivePool2d window rule over a sweep of sizes
        pool_window_case(common, "adaptive_pool_windows")
        return
    if "--results-only" in sys.argv:                  # the --save-txt / --save-json file formats (test.py:162-171, 184-195, 248-258)
        results_case(general, "result_files")
        return
    if "--rect-only" in sys.argv:                     # the shape real KAIST validation batches have under the reference's rect protocol
        model_case(yt, "model_s_kaist_544x672_b1", "yolov5s_Transfusion_kaist.yaml", 1, 544, 672, seed=15)     # (512x640 frames,
        return                                                                                                 # pad 0.5 -> 544x672; DMFF windows (11,8) / (4,12) / (8,3))
    if "--fusion-variants-only" in sys.argv:          # the NiNfusion / Add fixtures, added later
        model_case(yt, "model_s_add_kaist_320_b1", "yolov5s_Add_kaist.yaml", 1, 320, 320, seed=11)
        model_case(yt, "model_n_ninfusion_flir_320_b2", "yolov5n_NiNfusion_FLIR.yaml", 2, 320, 320, seed=12)
        return
    z_s = model_case(yt, "model_s_kaist_320_b2", "yolov5s_Transfusion_kaist.yaml", 2, 320, 320, seed=1)
    model_case(yt, "model_s_kaist_640_b1", "yolov5s_Transfusion_kaist.yaml", 1, 640, 640, seed=0)
    model_case(yt, "model_s_kaist_384x320_loops3", "yolov5s_Transfusion_kaist.yaml", 1, 384, 320, seed=2, loops=3)
    z_l = model_case(yt, "model_l_vedai_320_b1", "yolov5l_Transfusion_VEDAI.yaml", 1, 320, 320, seed=3)
    dmff_case(common, "dmff_c128_20x20_in40x40", 128, 20, 20, 2, 40, 40, seed=5, loops=1)
    dmff_case(common, "dmff_c256_16x16_in40x40_overlap", 256, 16, 16, 1, 40, 40, seed=6, loops=1)
    dmff_case(common, "dmff_c128_20x20_in64x80_rect_loops3", 128, 20, 20, 1, 64, 80, seed=7, loops=3)
    dmff_case(common, "dmff_c512_10x10_in10x10_identity", 512, 10, 10, 2, 10, 10, seed=8, loops=1)
    nms_case(general, "nms_s_conf25", z_s.numpy(), conf_thres=0.25, iou_thres=0.45)
    nms_case(general, "nms_s_conf30", z_s.numpy(), conf_thres=0.30, iou_thres=0.3)
    nms_case(general, "nms_s_conf001_iou5", z_s.numpy(), conf_thres=0.001, iou_thres=0.5)
    nms_case(general, "nms_l_multilabel", z_l.numpy(), conf_thres=0.001, iou_thres=0.5, multi_label=True)
    nms_case(general, "nms_l_agnostic_classes", z_l.numpy(), conf_thres=0.3, iou_thres=0.6, agnostic=True,
             classes=[0, 2, 5])
    metrics_case(general, metrics, "metrics_ap")
    model_case(yt, "model_s_add_kaist_320_b1", "yolov5s_Add_kaist.yaml", 1, 320, 320, seed=11)
    model_case(yt, "model_n_ninfusion_flir_320_b2", "yolov5n_NiNfusion_FLIR.yaml", 2, 320, 320, seed=12)
    model_case(yt, "model_m_kaist_320_b1", "yolov5m_Transfusion_kaist.yaml", 1, 320, 320, seed=13)
    model_case(yt, "model_n_flir_352x320_b2", "yolov5n_Transfusion_FLIR.yaml", 2, 352, 320, seed=14)
    match_case(general, "match_predictions")
    model_case(yt, "model_s_kaist_544x672_b1", "yolov5s_Transfusion_kaist.yaml", 1, 544, 672, seed=15)
    results_case(general, "result_files")
    pool_window_case(common, "adaptive_pool_windows")
    dataset_case("rect_dataset")
    half_case(yt, "reference_16bit")


if __name__ == "__main__":
    main()
