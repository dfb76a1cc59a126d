"""--save-txt / --save-json outputs of the validation loop (icafusion_amd/utils/results.py) against files the reference's own
test.py statements produced (tests/golden/result_files.json, made by tests/golden/make_golden.py --results-only by exec'ing
test.py:162-171, :184-195 and :248-258 on the detections of match_predictions.npz), and the small path helpers beside them."""
import json
import os

import numpy as np
import pytest

from helpers import load_golden
from icafusion_amd.utils.general import check_img_size, increment_path, xyxy2xywh2
from icafusion_amd.utils.results import ResultWriter, frame_index, label_listing

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("save_conf", [True, False])
def test_result_files_equal_the_reference_statements(tmp_path, save_conf):
    want = json.load(open(os.path.join(HERE, "golden", "result_files.json")))
    g = load_golden("match_predictions")
    run = want["runs"]["conf" if save_conf else "noconf"]
    w = ResultWriter(tmp_path / "exp", save_txt=True, save_conf=save_conf, save_json=True, label_names=want["labels_list"],
                     weights=["runs/train/exp6/weights/best.pt"])
    for t, stem in enumerate(want["stems"]):
        pred, predn = g[f"pred{t}"], g[f"predn{t}"]
        w.add(f"/data/visible/test/{stem}.jpg", predn[:, :4], pred[:, 4], pred[:, 5])
    result_txt, pred_json = w.close()
    got = {n: open(tmp_path / "exp" / "labels" / n).read() for n in sorted(os.listdir(tmp_path / "exp" / "labels"))}
    assert sorted(got) == sorted(run["files"])                       # an image without detections leaves no file
    for n in got:
        assert got[n] == run["files"][n], n
    assert result_txt.name == "result.txt" and pred_json.name == "best_predictions.json"
    assert json.load(open(pred_json)) == run["jdict"]


def test_result_writer_modes(tmp_path):
    with pytest.raises(ValueError):
        ResultWriter(tmp_path / "a", save_txt=True)                  # frame numbers need the label listing
    w = ResultWriter(tmp_path / "b", save_json=True, weights=None)
    w.add("x/000123.png", np.array([[1.0, 2.0, 11.0, 22.0]]), [0.5], [2])
    assert w.close() == (None, tmp_path / "b" / "_predictions.json")
    assert json.load(open(tmp_path / "b" / "_predictions.json")) == [{"image_id": 123, "category_id": 2, "bbox": [1.0, 2.0, 10.0, 20.0], "score": 0.5}]
    w = ResultWriter(tmp_path / "c")                                 # nothing requested: only the run directory exists
    w.add("x/i.png", np.zeros((0, 4)), [], [])
    assert w.close() == (None, None) and os.listdir(tmp_path / "c") == []
    with pytest.raises(ValueError):
        frame_index(["a.txt"], "b")                                  # image without a label file: as the reference's list.index
    os.makedirs(tmp_path / "lab")
    for n in ("b.txt", "a.txt", "10.txt"):
        open(tmp_path / "lab" / n, "w").close()
    assert label_listing(tmp_path / "lab") == ["10.txt", "a.txt", "b.txt"]


def test_path_helpers(tmp_path, capsys):
    p = increment_path(tmp_path / "exp")
    assert p == tmp_path / "exp"
    p.mkdir()
    assert increment_path(tmp_path / "exp") == tmp_path / "exp2"
    (tmp_path / "exp2").mkdir()
    (tmp_path / "exp7").mkdir()
    assert increment_path(tmp_path / "exp") == tmp_path / "exp8"                     # one past the highest sibling, not the first gap
    assert increment_path(tmp_path / "exp", exist_ok=True) == tmp_path / "exp"
    q = increment_path(tmp_path / "runs" / "test" / "exp", mkdir=True)
    assert q.is_dir()
    (tmp_path / "r.txt").write_text("x")
    assert increment_path(tmp_path / "r.txt") == tmp_path / "r2.txt"                 # a file keeps its suffix
    assert check_img_size(640, 32) == 640 and check_img_size(650, 32) == 672
    assert "must be multiple of max stride" in capsys.readouterr().out
    np.testing.assert_array_equal(xyxy2xywh2(np.array([[1.0, 2.0, 5.0, 9.0]])), [[1.0, 2.0, 4.0, 7.0]])


def _root_test_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("icaf_root_test", os.path.join(os.path.dirname(HERE), "test.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_summary_table_and_metrics():
    """test.py's statistics -> metrics + the reference's table layout (test.py:287-313), including the empty cases."""
    val = _root_test_module()
    g = np.random.default_rng(3)
    iouv_n = 10

    def image(n, tcls, nc):
        conf = np.sort(g.random(n))[::-1]
        pcls = g.integers(0, nc, n).astype(np.float64)
        correct = np.zeros((n, iouv_n), bool)
        hit = g.random(n) < 0.6
        depth = g.integers(1, iouv_n + 1, n)
        for i in range(n):
            if hit[i] and pcls[i] in tcls:
                correct[i, :depth[i]] = True                  # a TP at threshold t is a TP at every lower one
        return correct, conf, pcls, list(tcls)

    # one class: TP / FP / FN / F1 columns
    stats = [image(12, [0.0, 0.0, 0.0], 1), image(0, [0.0], 1)[:3] + ([0.0],), image(7, [0.0], 1)]
    stats[1] = (np.zeros((0, 10), bool), np.zeros(0), np.zeros(0), [0.0])
    (mp, mr, map50, map_), maps, lines = val.summarize(stats, 1, ["person"], seen=3)
    from icafusion_amd.utils.metrics import ap_per_class
    cat = [np.concatenate([np.asarray(s[k]) for s in stats], 0) for k in range(4)]
    tp, fp, fn, p, r, ap, f1, cls = ap_per_class(*cat)
    assert (mp, mr, map50, map_) == (p.mean(), r.mean(), ap[:, 0].mean(), ap.mean()) and maps.tolist() == [ap[0].mean()]
    assert lines[0].split() == ["Class", "Images", "Labels", "TP", "FP", "FN", "F1", "P", "R", "mAP@.5", "mAP@.5:.95"]
    row = lines[1].split()
    assert row[:3] == ["all", "3", "5"] and len(row) == 11 and float(row[3]) == pytest.approx(tp.sum(), rel=1e-3)
    # several classes: mAP@.75 column, per-class rows when verbose
    stats = [image(20, [0.0, 2.0, 2.0], 3), image(15, [1.0], 3)]
    (mp, mr, map50, map_), maps, lines = val.summarize(stats, 3, ["person", "car", "bicycle"], seen=2, verbose=True)
    assert lines[0].split() == ["Class", "Images", "Labels", "P", "R", "mAP@.5", "mAP@.75", "mAP@.5:.95"]
    assert [l.split()[0] for l in lines[1:]] == ["all", "person", "car", "bicycle"] and lines[1].split()[2] == "4"
    assert len(maps) == 3 and 0.0 <= map_ <= map50 <= 1.0
    assert len(val.summarize(stats, 3, ["a", "b", "c"], seen=2)[2]) == 2
    # nothing detected at all / detections but no true positive
    (mp, mr, map50, map_), maps, lines = val.summarize([], 1, ["person"], seen=0)
    assert (mp, mr, map50, map_) == (0.0, 0.0, 0.0, 0.0) and maps.tolist() == [0.0] and lines[1].split()[:3] == ["all", "0", "0"]
    none = [(np.zeros((4, 10), bool), np.array([0.9, 0.8, 0.7, 0.6]), np.zeros(4), [0.0, 0.0])]
    (mp, mr, map50, map_), maps, lines = val.summarize(none, 2, ["a", "b"], seen=1)
    assert (mp, mr, map50, map_) == (0.0, 0.0, 0.0, 0.0) and lines[1].split()[:3] == ["all", "1", "2"]


def test_root_test_rejects_what_is_not_built():
    val = _root_test_module()
    with pytest.raises(NotImplementedError):
        val.test({"nc": 1}, augment=True)


def test_validation_loop_on_cpu_with_the_oracle_behind_it(tmp_path, monkeypatch):
    """Root test.py's loop, statistics, result files and return value with the three device calls (forward, NMS, TP matching) replaced by
    their CPU oracle statements: the host logic around the kernels — label scaling, writer wiring, frame numbers, table — runs without
    a GPU.  The metrics must equal an independent evaluation of the same oracle detections."""
    import sys
    import torch
    import yaml
    sys.path.insert(0, HERE)
    from test_frontends import make_dataset
    from helpers import REPO
    from icafusion_amd.models.yolo import Model
    from icafusion_amd.synth import synth_state_dict
    from icafusion_amd.utils import datasets as D
    from icafusion_amd.utils.general import scale_coords, xywh2xyxy
    from icafusion_amd.utils.metrics import match_predictions
    from oracle import icaf_oracle as oracle
    val = _root_test_module()
    rgb_dir, ir_dir = make_dataset(str(tmp_path / "set"), n=3, size=(120, 128), nc=3, seed=5)
    cfg = yaml.safe_load(open(os.path.join(REPO, "models", "transformer", "yolov5s_Transfusion_FLIR.yaml")))
    sd = synth_state_dict(Model(cfg), seed=0)
    om = oracle.OracleModel(cfg, sd)
    MAX_DET = 300

    class FakeModel:
        stride = torch.tensor([8.0, 16.0, 32.0])

        def parameters(self):
            yield torch.zeros(1)

        def forward_u8(self, img6):
            f = img6.float() / 255.0
            return (om.forward(f[:, :3].contiguous(), f[:, 3:].contiguous())[0],)

    def fake_nms(out, conf_thres, iou_thres, multi_label=False, agnostic=False, **kw):
        dets = oracle.non_max_suppression(out.numpy(), conf_thres, iou_thres, multi_label=multi_label, agnostic=agnostic)
        det = torch.zeros((len(dets), MAX_DET, 6))
        for i, d in enumerate(dets):
            det[i, :len(d)] = torch.from_numpy(d)
        return det, torch.tensor([len(d) for d in dets], dtype=torch.int32), None

    def fake_match(det, count, labels, label_off, iouv, scale=None, predn=None, stream_ptr=None):
        B = det.shape[0]
        correct = torch.zeros((B, MAX_DET, iouv.numel()), dtype=torch.uint8)
        for b in range(B):
            n = int(count[b])
            d = det[b, :n].clone()
            gain, px, py, w0, h0 = scale[b].tolist()
            scale_coords(None, d[:, :4], (h0, w0), ((gain, gain), (px, py)))
            if predn is not None:
                predn[b, :n] = d[:, :4]
            lab = labels[int(label_off[b]):int(label_off[b + 1])]
            correct[b, :n] = torch.from_numpy(match_predictions(d.numpy(), lab.numpy(), iouv.numpy()).astype(np.uint8))
        return correct

    monkeypatch.setattr(val, "nms_device", fake_nms)
    monkeypatch.setattr(val.ops, "match_predictions", fake_match)
    data = {"val_rgb": rgb_dir, "val_ir": ir_dir, "nc": 3, "names": ["person", "car", "bicycle"]}
    run = tmp_path / "runs" / "exp"
    (mp, mr, map50, map_, *_), maps, tt = val.test(data, weights=["w/best.pt"], batch_size=2, imgsz=320, conf_thres=0.3, model=FakeModel(),
                                                   save_txt=True, save_conf=True, save_json=True, save_dir=run, verbose=True)
    # independent evaluation of the same detections
    loader, ds = D.create_dataloader_rgb_ir(rgb_dir, ir_dir, 320, 2, 32, None, pad=0.5, rect=True)
    iouv = np.linspace(0.5, 0.95, 10)
    tp, conf, pcls, tcls, per_image = [], [], [], [], {}
    for img6, targets, paths, shapes in loader:
        f = img6.float() / 255.0
        dets = oracle.non_max_suppression(om.forward(f[:, :3].contiguous(), f[:, 3:].contiguous())[0].numpy(), 0.3, 0.5, multi_label=True)
        H, W = img6.shape[2:]
        targets[:, 2:] *= torch.tensor([W, H, W, H])
        for si, d in enumerate(dets):
            lab = targets[targets[:, 0] == si, 1:]
            dn = torch.from_numpy(d.copy()); scale_coords((H, W), dn[:, :4], shapes[si][0], shapes[si][1])
            tb = xywh2xyxy(lab[:, 1:5]); scale_coords((H, W), tb, shapes[si][0], shapes[si][1])
            tp.append(oracle.match_predictions(dn.numpy(), torch.cat((lab[:, :1], tb), 1).numpy(), iouv))
            conf.append(d[:, 4]); pcls.append(d[:, 5]); tcls.append(lab[:, 0].numpy())
            per_image[os.path.splitext(os.path.basename(paths[si]))[0]] = dn.numpy()
    assert sum(len(c) for c in conf) > 0
    if np.concatenate(tp).any():
        ap, _ = oracle.ap_per_class(np.concatenate(tp), np.concatenate(conf), np.concatenate(pcls), np.concatenate(tcls))
        assert map50 == pytest.approx(ap[:, 0].mean(), abs=1e-6) and map_ == pytest.approx(ap.mean(), abs=1e-6)
    else:
        assert (map50, map_) == (0.0, 0.0)
    # result files: one text file per image with detections, frame = 1-based rank of its label file, native-space top-left + size
    listing = sorted(os.listdir(os.path.join(str(tmp_path / "set"), "labels", "test")))
    files = sorted(os.listdir(run / "labels"))
    assert "result.txt" in files and files == sorted([s + ".txt" for s, d in per_image.items() if len(d)] + ["result.txt"])
    total = 0
    for stem, d in per_image.items():
        if not len(d):
            continue
        rows = np.loadtxt(run / "labels" / (stem + ".txt"), delimiter=",", ndmin=2)
        assert rows.shape == (len(d), 6) and (rows[:, 0] == listing.index(stem + ".txt") + 1).all()
        np.testing.assert_allclose(rows[:, 1:3], d[:, :2], rtol=1e-5, atol=1e-4)                 # %g keeps 6 significant digits
        np.testing.assert_allclose(rows[:, 3:5], d[:, 2:4] - d[:, :2], rtol=1e-5, atol=1e-4)
        np.testing.assert_allclose(rows[:, 5], d[:, 4], rtol=1e-5)
        total += len(d)
    assert len(open(run / "labels" / "result.txt").read().splitlines()) == total
    rows = json.load(open(run / "best_predictions.json"))
    assert len(rows) == total and {r["image_id"] for r in rows} == {s for s, d in per_image.items() if len(d)}
    assert all(set(r) == {"image_id", "category_id", "bbox", "score"} and r["category_id"] in (0, 1, 2) for r in rows)


def test_detect_twostream_loop_on_cpu_with_the_oracle_behind_it(tmp_path, monkeypatch):
    """detect_twostream.py's host side (paired loaders, letterbox, box scaling, label files, annotated images, run directory numbering)
    with forward + NMS replaced by the CPU oracle."""
    import sys
    import torch
    import yaml
    sys.path.insert(0, HERE)
    from test_frontends import make_dataset
    from helpers import REPO
    sys.path.insert(0, REPO)
    import detect_twostream as dt
    from icafusion_amd.models.yolo import Model
    from icafusion_amd.synth import synth_state_dict
    from oracle import icaf_oracle as oracle
    rgb_dir, ir_dir = make_dataset(str(tmp_path / "set"), n=3, size=(120, 128), nc=3, seed=6)
    cfg = yaml.safe_load(open(os.path.join(REPO, "models", "transformer", "yolov5s_Transfusion_FLIR.yaml")))
    om = oracle.OracleModel(cfg, synth_state_dict(Model(cfg), seed=0))

    class FakeModel:
        stride = torch.tensor([8.0, 16.0, 32.0])
        names = ["person", "car", "bicycle"]

        def forward_u8(self, img6):
            f = img6.float() / 255.0
            return (om.forward(f[:, :3].contiguous(), f[:, 3:].contiguous())[0],)

    def fake_nms(pred, conf_thres, iou_thres, classes=None, agnostic=False, **kw):
        return [torch.from_numpy(d.copy()) for d in oracle.non_max_suppression(pred.numpy(), conf_thres, iou_thres, classes=classes, agnostic=agnostic)]

    monkeypatch.setattr(dt, "load_model", lambda opt, device: FakeModel())
    monkeypatch.setattr(dt, "non_max_suppression", fake_nms)
    monkeypatch.setattr(dt, "select_device", lambda d: torch.device("cpu"))
    args = ["--source1", rgb_dir, "--source2", ir_dir, "--img-size", "320", "--conf-thres", "0.3", "--save-txt", "--save-conf",
            "--project", str(tmp_path / "runs"), "--name", "exp", "--hide-conf"]
    out = dt.detect(dt.parse_opt(args))
    assert out == tmp_path / "runs" / "exp" and len(list(out.glob("*_rgb.png"))) == 3 and len(list(out.glob("*_ir.png"))) == 3
    txts = sorted((out / "labels").glob("*.txt"))
    assert txts, "conf 0.3 leaves detections on the synthetic weights"
    for t in txts:
        rows = np.loadtxt(t, ndmin=2)                                    # cls cx cy w h conf, normalised to the native image
        assert rows.shape[1] == 6 and (rows[:, 1:5] >= 0).all() and (rows[:, 1:5] <= 1).all() and (rows[:, 5] >= 0.3).all()
        assert set(rows[:, 0].astype(int)) <= {0, 1, 2}
    assert dt.detect(dt.parse_opt(args)) == tmp_path / "runs" / "exp2"                  # the run directory is numbered, not overwritten
    assert dt.detect(dt.parse_opt(args + ["--exist-ok", "--nosave"])) == tmp_path / "runs" / "exp"


def test_single_cls_reaches_the_dataset(tmp_path, monkeypatch):
    """`test(..., single_cls=True)` on a multi-class label set: the reference hands `opt` to the dataloader, whose dataset zeroes the label
    classes (utils/datasets.py:465-466, test.py:100) while the loop zeroes the detections' (test.py:141-142) - with the flag stopping at
    the loop, labels 1..n never match class-0 detections and P / R / mAP are silently wrong."""
    import sys
    import torch
    import yaml
    sys.path.insert(0, HERE)
    from test_frontends import make_dataset
    from helpers import REPO
    from icafusion_amd.models.yolo import Model
    from icafusion_amd.synth import synth_state_dict
    from icafusion_amd.utils.general import scale_coords
    from icafusion_amd.utils.metrics import match_predictions
    from oracle import icaf_oracle as oracle
    val = _root_test_module()
    rgb_dir, ir_dir = make_dataset(str(tmp_path / "set"), n=2, size=(120, 128), nc=3, seed=8)
    cfg = yaml.safe_load(open(os.path.join(REPO, "models", "transformer", "yolov5s_Transfusion_kaist.yaml")))       # nc = 1 head
    om = oracle.OracleModel(cfg, synth_state_dict(Model(cfg), seed=0))
    seen = {}
    real = val.create_dataloader_rgb_ir

    def spy(*a, **kw):
        loader, ds = real(*a, **kw)
        seen["opt"] = a[5]
        seen["classes"] = sorted({float(c) for lab in ds.labels for c in lab[:, 0]})
        return loader, ds

    class FakeModel:
        stride = torch.tensor([8.0, 16.0, 32.0])

        def parameters(self):
            yield torch.zeros(1)

        def forward_u8(self, img6):
            f = img6.float() / 255.0
            return (om.forward(f[:, :3].contiguous(), f[:, 3:].contiguous())[0],)

    def fake_nms(out, conf_thres, iou_thres, multi_label=False, agnostic=False, **kw):
        assert agnostic
        dets = oracle.non_max_suppression(out.numpy(), conf_thres, iou_thres, multi_label=multi_label, agnostic=agnostic)
        det = torch.zeros((len(dets), 300, 6))
        for i, d in enumerate(dets):
            det[i, :len(d)] = torch.from_numpy(d)
        return det, torch.tensor([len(d) for d in dets], dtype=torch.int32), None

    def fake_match(det, count, labels, label_off, iouv, scale=None, predn=None, stream_ptr=None):
        assert float(labels[:, 0].abs().max()) == 0.0                  # every label arrives as class 0
        correct = torch.zeros((det.shape[0], 300, iouv.numel()), dtype=torch.uint8)
        for b in range(det.shape[0]):
            n = int(count[b])
            d = det[b, :n].clone()
            gain, px, py, w0, h0 = scale[b].tolist()
            scale_coords(None, d[:, :4], (h0, w0), ((gain, gain), (px, py)))
            lab = labels[int(label_off[b]):int(label_off[b + 1])]
            correct[b, :n] = torch.from_numpy(match_predictions(d.numpy(), lab.numpy(), iouv.numpy()).astype(np.uint8))
        return correct

    monkeypatch.setattr(val, "create_dataloader_rgb_ir", spy)
    monkeypatch.setattr(val, "nms_device", fake_nms)
    monkeypatch.setattr(val.ops, "match_predictions", fake_match)
    data = {"val_rgb": rgb_dir, "val_ir": ir_dir, "nc": 3, "names": ["person", "car", "bicycle"]}
    (mp, mr, map50, map_, *_), maps, _ = val.test(data, batch_size=2, imgsz=320, conf_thres=0.3, single_cls=True, model=FakeModel())
    assert seen["opt"].single_cls is True and seen["classes"] == [0.0]
    assert maps.shape == (1,)
