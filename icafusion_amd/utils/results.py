"""Result files of the validation loop, in the formats the reference's test.py leaves behind (the data format on the far side
of the hot path):

* `--save-txt`: one `<image stem>.txt` per image under `<run>/labels/`, one line per detection
  `frame,x1,y1,w,h[,conf]` — frame = 1-based position of the image's label file in the sorted label directory, box in native
  image pixels as top-left corner + size, every number printed with `%g`, comma separated (test.py:162-171); after the loop all
  files are concatenated in name order into `<run>/labels/result.txt` (test.py:248-258) — the input of the KAIST miss-rate
  evaluator (`evaluation_script/`, whose call site the reference has disabled);
* `--save-json`: `<run>/<weights stem>_predictions.json`, a list of `{image_id, category_id, bbox [x, y, w, h], score}` with the box
  rounded to 3 and the score to 5 decimals, image_id an int when the stem is numeric (test.py:184-195, 330-335).

Host-side Python on the (B, max_det, ...) blocks the device kernels return; nothing here touches the GPU."""
import json
import os
from pathlib import Path

import numpy as np


def frame_index(label_names, stem):
    """0-based position of `<stem>.txt` in the sorted listing of the label directory (test.py:165, 397-399).  ValueError when the
    image has no label file, as the reference's list.index raises."""
    return label_names.index(str(stem) + ".txt")


def label_listing(label_dir):
    """Sorted file names of a label directory (test.py:397-399: `os.listdir` + `sort`)."""
    names = os.listdir(label_dir)
    names.sort()
    return names


class ResultWriter:
    """Collects the per-image outputs of one validation run.

    add(path, predn, conf, cls): predn (n, 4) xyxy in native image pixels, conf (n,), cls (n,) — numpy or nested lists.
    close() writes result.txt / the JSON file and returns their paths (None where the output is off)."""

    def __init__(self, save_dir, save_txt=False, save_conf=True, save_json=False, label_names=None, weights=None):
        self.save_dir = Path(save_dir)
        self.save_txt, self.save_conf, self.save_json = bool(save_txt), bool(save_conf), bool(save_json)
        self.labels_dir = self.save_dir / "labels"
        (self.labels_dir if self.save_txt else self.save_dir).mkdir(parents=True, exist_ok=True)
        if self.save_txt and label_names is None:
            raise ValueError("save_txt needs the sorted listing of the label directory (frame numbers are positions in it)")
        self.label_names = list(label_names) if label_names is not None else None
        w = weights[0] if isinstance(weights, (list, tuple)) and weights else weights
        self.weights_stem = Path(w).stem if w else ""
        self.jdict = []

    def add(self, path, predn, conf, cls):
        path = Path(path)
        predn = np.asarray(predn, dtype=np.float32).reshape(-1, 4)
        conf, cls = np.asarray(conf, dtype=np.float32).reshape(-1), np.asarray(cls, dtype=np.float32).reshape(-1)
        if self.save_txt and len(predn):            # the reference opens the file per detection: no detections, no file
            frame = frame_index(self.label_names, path.stem) + 1
            with open(self.labels_dir / (path.stem + ".txt"), "a") as f:
                wh = predn[:, 2:4] - predn[:, 0:2]                 # float32, as the reference's xyxy2xywh2 on a float32 tensor
                for (x1, y1), (w, h), c in zip(predn[:, 0:2].tolist(), wh.tolist(), conf.tolist()):
                    line = (frame, x1, y1, w, h) + ((c,) if self.save_conf else ())
                    f.write(("%g," * len(line)).rstrip(",") % line + "\n")
        if self.save_json:
            image_id = int(path.stem) if path.stem.isnumeric() else path.stem
            for (x1, y1, x2, y2), c, k in zip(predn.tolist(), conf.tolist(), cls.tolist()):
                # xyxy -> centre form -> top-left (test.py:188-189: the same float32 arithmetic, then rounding)
                w, h = np.float32(x2) - np.float32(x1), np.float32(y2) - np.float32(y1)
                cx, cy = (np.float32(x1) + np.float32(x2)) / np.float32(2), (np.float32(y1) + np.float32(y2)) / np.float32(2)
                box = [float(cx - w / np.float32(2)), float(cy - h / np.float32(2)), float(w), float(h)]
                self.jdict.append({"image_id": image_id, "category_id": int(k), "bbox": [round(v, 3) for v in box],
                                   "score": round(float(c), 5)})

    def close(self):
        result_txt = pred_json = None
        if self.save_txt:
            files = sorted(os.listdir(self.labels_dir))
            lines = []
            for name in files:
                with open(self.labels_dir / name) as f:
                    lines += f.readlines()
            result_txt = self.labels_dir / "result.txt"
            with open(result_txt, "a") as f:
                f.writelines(lines)
        if self.save_json and self.jdict:
            pred_json = self.save_dir / f"{self.weights_stem}_predictions.json"
            with open(pred_json, "w") as f:
                json.dump(self.jdict, f)
        return result_txt, pred_json
