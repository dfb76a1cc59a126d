"""Front ends around the hot path: image loading without OpenCV, the paired validation set, TP
matching, and — on the GPU — detect_twostream.py / test.py end to end on a synthetic paired dataset."""
import os
import sys

import numpy as np
import pytest
import torch

from icafusion_amd.utils import datasets as D
from icafusion_amd.utils.general import scale_coords, xywh2xyxy
from icafusion_amd.utils.metrics import match_predictions
from oracle import icaf_oracle as oracle

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_dataset(root, n=4, size=(96, 128), nc=1, seed=0, mixed=True):
    """The layout of the shipped data/multispectral/*.yaml files: root/visible/test/*.png, root/infrared/test/*.png and
    root/labels/test/*.txt (YOLO: cls cx cy w h) — labels are found by the reference's visible -> labels rule."""
    g = np.random.default_rng(seed)
    for mod in ("visible", "infrared", "labels"):
        os.makedirs(os.path.join(root, mod, "test"), exist_ok=True)
    for i in range(n):
        h, w = size if (i % 2 == 0 or not mixed) else (size[1], size[0])
        for mod in ("visible", "infrared"):
            D.imwrite_bgr(os.path.join(root, mod, "test", f"im{i:03d}.png"), g.integers(0, 256, (h, w, 3), dtype=np.uint8))
        k = int(g.integers(1, 5))
        lab = np.concatenate((g.integers(0, nc, (k, 1)).astype(np.float32), g.uniform(0.2, 0.8, (k, 2)), g.uniform(0.1, 0.3, (k, 2))), 1)
        np.savetxt(os.path.join(root, "labels", "test", f"im{i:03d}.txt"), lab, fmt="%g")
    return os.path.join(root, "visible", "test"), os.path.join(root, "infrared", "test")


def test_img2label_paths_follow_the_reference_rule():
    """utils/datasets.py:391-401: the first 'visible' (else 'infrared') becomes 'labels', the extension 'txt'."""
    got = D.img2label_paths(["/data/kaist/visible/test/set06_V000_I00019.jpg", "/data/kaist/infrared/train/a.b.png",
                             "/d/visible/x/visible/y.jpeg"])
    assert got == ["/data/kaist/labels/test/set06_V000_I00019.txt", "/data/kaist/labels/train/a.b.txt", "/d/labels/x/visible/y.txt"]
    with pytest.raises(ValueError, match="visible"):
        D.img2label_paths(["/data/kaist/images/test/x.jpg"])


def test_missing_labels_fail_loudly(tmp_path):
    rgb_dir, ir_dir = make_dataset(str(tmp_path), n=2)
    os.rename(os.path.join(str(tmp_path), "labels"), os.path.join(str(tmp_path), "annotations"))
    with pytest.raises(FileNotFoundError, match="no label file"):
        D.create_dataloader_rgb_ir(rgb_dir, ir_dir, 128, 2)


def test_rectangular_batches_follow_the_reference_protocol(tmp_path):
    """test.py:100 -> rect=True, pad=0.5: images sorted by h / w, one shape per batch = ceil(shape * img_size / stride + pad)
    * stride (utils/datasets.py:826-849); KAIST's 512x640 frames at img_size 640 become 544x672 batches."""
    rgb_dir, ir_dir = make_dataset(str(tmp_path / "kaist"), n=3, size=(512, 640), mixed=False)
    loader, ds = D.create_dataloader_rgb_ir(rgb_dir, ir_dir, 640, 2, 32, None, pad=0.5, rect=True)
    assert ds.batch_shapes.tolist() == [[544, 672], [544, 672]]
    shapes_seen = []
    for img6, targets, paths, shapes in loader:
        shapes_seen.append(tuple(img6.shape))
        (h0, w0), ((rh, rw), (pw, ph)) = shapes[0]
        assert (h0, w0, rh, rw, pw, ph) == (512, 640, 1.0, 1.0, 16.0, 16.0)
        assert (img6[0, :, :16] == 114).all() and (img6[0, :, :, :16] == 114).all() and (img6[0, :, -16:] == 114).all()
    assert shapes_seen == [(2, 6, 544, 672), (1, 6, 544, 672)]          # ragged last batch
    # mixed orientations: portrait images sort behind landscape ones and each batch takes the shape of its own members
    rgb_dir, ir_dir = make_dataset(str(tmp_path / "mixed"), n=4, size=(96, 128), mixed=True)
    loader, ds = D.create_dataloader_rgb_ir(rgb_dir, ir_dir, 128, 2, 32, None, pad=0.5, rect=True)
    assert [os.path.basename(f) for f in ds.rgb] == ["im000.png", "im002.png", "im001.png", "im003.png"]
    assert ds.batch_shapes.tolist() == [[128, 160], [160, 128]]             # ceil([0.75, 1] * 128 / 32 + 0.5) * 32
    for (img6, targets, paths, shapes), want in zip(loader, [(2, 6, 128, 160), (2, 6, 160, 128)]):
        assert tuple(img6.shape) == want
    # a down-scaled set: longest side -> img_size by area averaging, no up-scaling afterwards (scaleup=False)
    loader, ds = D.create_dataloader_rgb_ir(rgb_dir, ir_dir, 64, 4, 32, None, pad=0.0, rect=False)
    img6, targets, paths, shapes = next(iter(loader))
    assert tuple(img6.shape) == (4, 6, 64, 64) and shapes[0][1][0] == (0.5, 0.5)


def test_resize_area_is_the_exact_box_average():
    a = np.random.default_rng(4).integers(0, 256, (8, 12, 3), dtype=np.uint8)
    half = D.resize_area(a, (6, 4))
    ref = a.reshape(4, 2, 6, 2, 3).astype(np.float32).mean((1, 3))
    assert np.abs(half.astype(np.float32) - np.floor(ref + 0.5)).max() == 0
    assert (D.resize_area(np.full((9, 7, 3), 77, np.uint8), (3, 4)) == 77).all()
    frac = D.resize_area(np.arange(5, dtype=np.uint8).reshape(1, 5, 1) * 50, (2, 1))      # 5 -> 2 columns: 2.5 pixels each
    assert frac.reshape(-1).tolist() == [int(np.floor((0 + 50 + 100 * 0.5) / 2.5 + 0.5)), int(np.floor((100 * 0.5 + 150 + 200) / 2.5 + 0.5))]


@pytest.mark.parametrize("shape,new", [((512, 640), 640), ((480, 640), (544, 672)), ((100, 50), 64), ((30, 300), 128)])
def test_letterbox_geometry(shape, new):
    img = np.random.default_rng(1).integers(0, 256, (*shape, 3), dtype=np.uint8)
    out, ratio, (dw, dh) = D.letterbox(img, new)
    nh, nw = (new, new) if isinstance(new, int) else new
    assert out.shape == (nh, nw, 3)
    r = min(nh / shape[0], nw / shape[1])
    assert ratio == (r, r)
    uw, uh = int(round(shape[1] * r)), int(round(shape[0] * r))
    assert dw == (nw - uw) / 2 and dh == (nh - uh) / 2
    top, left = int(round(dh - 0.1)), int(round(dw - 0.1))
    if top:
        assert (out[:top] == 114).all()
    if left:
        assert (out[:, :left] == 114).all()
    if (uw, uh) == (shape[1], shape[0]):                      # no resize: pixels are copied verbatim
        assert (out[top:top + uh, left:left + uw] == img).all()


def test_resize_bilinear_matches_half_pixel_filter():
    import torch.nn.functional as F
    a = np.random.default_rng(2).integers(0, 256, (37, 53, 3), dtype=np.uint8)
    t = F.interpolate(torch.from_numpy(a).permute(2, 0, 1)[None].float(), size=(80, 96), mode="bilinear", align_corners=False)
    ref = torch.floor(t[0].permute(1, 2, 0) + 0.5).numpy()
    assert np.abs(D.resize_bilinear(a, (96, 80)).astype(np.float32) - ref).max() <= 1.0
    assert (D.resize_bilinear(np.full((9, 7, 3), 200, np.uint8), (31, 15)) == 200).all()


def test_load_images_and_paired_set(tmp_path):
    rgb_dir, ir_dir = make_dataset(str(tmp_path), n=3)
    items = list(D.LoadImages(rgb_dir, img_size=128))
    assert len(items) == 3 and [os.path.basename(i[0]) for i in items] == ["im000.png", "im001.png", "im002.png"]
    path, img, img0, cap = items[0]
    assert img.shape == (3, 128, 128) and img.dtype == np.uint8 and img0.shape == (96, 128, 3) and cap is None
    assert (img[::-1, 16] == img0[0].T).all()                 # CHW RGB of the (unresized) BGR original below the pad rows
    loader, ds = D.create_dataloader_rgb_ir(rgb_dir, ir_dir, 128, 2)
    img6, targets, paths, shapes = next(iter(loader))
    assert img6.shape == (2, 6, 128, 128) and img6.dtype == torch.uint8 and targets.shape[1] == 6
    assert set(targets[:, 0].tolist()) <= {0.0, 1.0}
    # label round trip: letterboxed-normalised -> pixels -> scale_coords back == the file's boxes in native pixels
    for si in range(2):
        lab = targets[targets[:, 0] == si, 1:].clone()
        box = xywh2xyxy(lab[:, 1:5] * torch.tensor([128, 128, 128, 128.0]))
        (h0, w0), pad = shapes[si]
        scale_coords((128, 128), box, (h0, w0), pad)
        raw = np.loadtxt(os.path.join(str(tmp_path), "labels", "test", os.path.basename(paths[si])[:-4] + ".txt"), ndmin=2)
        want = xywh2xyxy(torch.from_numpy(raw[:, 1:5]).float() * torch.tensor([w0, h0, w0, h0.__float__()]))
        want[:, [0, 2]] = want[:, [0, 2]].clamp(0, w0)
        want[:, [1, 3]] = want[:, [1, 3]].clamp(0, h0)
        assert torch.allclose(box, want, atol=1e-3)


def test_match_predictions_equals_oracle():
    g = np.random.default_rng(3)
    iouv = np.linspace(0.5, 0.95, 10)
    for trial in range(20):
        m, n = int(g.integers(0, 6)), int(g.integers(0, 40))
        gt_xy = g.uniform(0, 80, (m, 2)); gt = np.concatenate((g.integers(0, 3, (m, 1)), gt_xy, gt_xy + g.uniform(10, 40, (m, 2))), 1)
        base = gt[g.integers(0, max(m, 1), n), 1:5] if m else g.uniform(0, 100, (n, 4))
        det = np.concatenate((base + g.normal(0, 3, (n, 4)), g.uniform(0, 1, (n, 1)), g.integers(0, 3, (n, 1))), 1).astype(np.float32)
        a = match_predictions(det, gt.astype(np.float32), iouv)
        b = oracle.match_predictions(det, gt.astype(np.float32), iouv)
        assert (a == b).all()


@pytest.mark.gpu
def test_detect_twostream_and_test_py_end_to_end(tmp_path):
    """detect_twostream.py writes one label line per NMS survivor; test.py's mAP on the synthetic set equals the mAP of
    the CPU oracle's detections on the same letterboxed batches (fp32 build, north_star: within 0.1)."""
    sys.path.insert(0, REPO)
    import detect_twostream as dt
    import test as val
    import yaml
    from icafusion_amd.models.yolo import Model
    from icafusion_amd.synth import synth_state_dict
    # 120x128 / 128x120 images at img-size 320 -> rectangular batches of 320x352 / 352x320: the smallest side must stay >= 320,
    # or the P5 map (stride 32) has fewer than the DMFF's 10x10 anchors and the model raises — as the reference does
    rgb_dir, ir_dir = make_dataset(str(tmp_path), n=4, size=(120, 128), nc=3, seed=5)
    cfg_path = os.path.join(REPO, "models", "transformer", "yolov5s_Transfusion_FLIR.yaml")
    opt = dt.parse_opt(["--cfg", cfg_path, "--source1", rgb_dir, "--source2", ir_dir, "--img-size", "320", "--conf-thres", "0.3",
                        "--save-txt", "--save-conf", "--project", str(tmp_path / "runs"), "--name", "exp"])
    out_dir = dt.detect(opt)
    txts = sorted((out_dir / "labels").glob("*.txt"))
    assert len(txts) >= 1 and len(list(out_dir.glob("*_rgb.png"))) == 4 and len(list(out_dir.glob("*_ir.png"))) == 4
    rows = np.loadtxt(txts[0], ndmin=2)
    assert rows.shape[1] == 6 and (rows[:, 1:5] >= 0).all() and (rows[:, 1:5] <= 1).all()

    data = {"val_rgb": rgb_dir, "val_ir": ir_dir, "nc": 3, "names": ["person", "car", "bicycle"]}
    cfg = yaml.safe_load(open(cfg_path))
    model = Model(cfg).eval()
    sd = synth_state_dict(model, seed=0)
    model.load_state_dict(sd)
    (mp, mr, map50, map_, *_), maps, _ = val.test(data, batch_size=2, imgsz=320, model=model.to("cuda:0"))
    # oracle detections on the same batches through the same statistics
    loader, _ = D.create_dataloader_rgb_ir(rgb_dir, ir_dir, 320, 2, 32, None, pad=0.5, rect=True)      # test.py:100
    om = oracle.OracleModel(cfg, sd)
    iouv = np.linspace(0.5, 0.95, 10)
    tp, conf, pcls, tcls = [], [], [], []
    for img6, targets, paths, shapes in loader:
        f = img6.float() / 255.0
        z = om.forward(f[:, :3].contiguous(), f[:, 3:].contiguous())[0].numpy()
        dets = oracle.non_max_suppression(z, 0.001, 0.5, multi_label=True)
        H, W = img6.shape[2:]                                       # rectangular batch shape
        targets[:, 2:] *= torch.tensor([W, H, W, H])
        for si, d in enumerate(dets):
            lab = targets[targets[:, 0] == si, 1:]
            dn = torch.from_numpy(d.copy()); scale_coords((H, W), dn[:, :4], shapes[si][0], shapes[si][1])
            tb = xywh2xyxy(lab[:, 1:5]); scale_coords((H, W), tb, shapes[si][0], shapes[si][1])
            tp.append(oracle.match_predictions(dn.numpy(), torch.cat((lab[:, :1], tb), 1).numpy(), iouv))
            conf.append(d[:, 4]); pcls.append(d[:, 5]); tcls.append(lab[:, 0].numpy())
    ap, _ = oracle.ap_per_class(np.concatenate(tp), np.concatenate(conf), np.concatenate(pcls), np.concatenate(tcls))
    assert abs(100 * map50 - 100 * ap[:, 0].mean()) <= 0.1 and abs(100 * map_ - 100 * ap.mean()) <= 0.1


def test_paired_validation_set_metadata_equals_the_reference_class(tmp_path):
    """File discovery, the visible -> labels rule, label parsing, the aspect-ratio sort and the rectangular batch shapes of the
    reference's LoadMultiModalImagesAndLabels (utils/datasets.py:690-880), recorded by tests/golden/make_golden.py --dataset-only
    from the reference class itself on the same synthetic folders make_dataset() writes — including the KAIST-like case whose
    single batch comes out as 544x672."""
    import json
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rect_dataset.npz"))
    cases = json.loads(str(g["cases"]))
    assert len(cases) == 3
    for k, c in enumerate(cases):
        root = str(tmp_path / f"set{k}")
        rgb_dir, ir_dir = make_dataset(root, n=c["n"], size=tuple(c["size"]), nc=c["nc"], seed=c["seed"], mixed=c["mixed"])
        _, ds = D.create_dataloader_rgb_ir(rgb_dir, ir_dir, c["img_size"], c["batch"], 32, None, pad=0.5, rect=True)
        assert [os.path.basename(f) for f in ds.rgb] == g[f"files{k}"].tolist(), k
        assert [os.path.basename(f) for f in ds.ir] == g[f"files{k}"].tolist()
        assert [os.path.relpath(f, root) for f in ds.label_files] == g[f"label_files{k}"].tolist()
        assert ds.batch_shapes.tolist() == g[f"batch_shapes{k}"].tolist(), k
        assert ds.batch.tolist() == g[f"batch{k}"].tolist()
        mine = np.concatenate([np.concatenate((np.full((len(l), 1), i, np.float32), l), 1) for i, l in enumerate(ds.labels)])
        np.testing.assert_array_equal(mine, g[f"labels{k}"])
    assert g["batch_shapes2"].tolist() == [[544, 672]]
