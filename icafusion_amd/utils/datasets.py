"""Image loading for the two-stream front ends, without OpenCV (cv2 is not in this image).

Same entry points and return conventions as the reference's utils/datasets.py for the inference / validation paths:
`letterbox` (:1404-1444 — note the reference's `auto` / `scaleFill` branches are commented out, so every image is padded
to the full `new_shape`), `LoadImages` (:172-241, image files only) and a paired RGB+IR validation set that yields the
6-channel uint8 batches `test.py:115-123` consumes (a compact stand-in for LoadMultiModalImagesAndLabels :690-1024:
square letterbox, no augmentation, no rectangular batching, no label cache).

Decoding goes through PIL, resizing is a half-pixel-centre bilinear filter evaluated in float32 and rounded to nearest:
the same sampling geometry as cv2.INTER_LINEAR, which however evaluates it in 11-bit fixed point — results can differ
from OpenCV's by one grey level.  Arrays are BGR HWC uint8 like cv2.imread's, so downstream code is unchanged."""
import glob
import os
from pathlib import Path

import numpy as np
import torch

IMG_FORMATS = ("bmp", "jpg", "jpeg", "png", "tif", "tiff", "dng", "webp", "mpo")


def imread_bgr(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.ascontiguousarray(np.asarray(im.convert("RGB"))[:, :, ::-1])


def imwrite_bgr(path, img):
    from PIL import Image
    Image.fromarray(np.ascontiguousarray(img[:, :, ::-1])).save(path)


def resize_bilinear(img, new_wh):
    """(H, W, C) uint8 -> (new_h, new_w, C) uint8, bilinear with half-pixel centres and edge clamping."""
    h, w = img.shape[:2]
    nw, nh = new_wh
    if (nw, nh) == (w, h):
        return img

    def taps(n_in, n_out):
        src = (np.arange(n_out, dtype=np.float32) + 0.5) * (n_in / n_out) - 0.5
        i0 = np.floor(src).astype(np.int64)
        frac = (src - i0).astype(np.float32)
        return np.clip(i0, 0, n_in - 1), np.clip(i0 + 1, 0, n_in - 1), frac
    y0, y1, fy = taps(h, nh)
    x0, x1, fx = taps(w, nw)
    f = img.astype(np.float32)
    top = f[y0][:, x0] * (1 - fx)[None, :, None] + f[y0][:, x1] * fx[None, :, None]
    bot = f[y1][:, x0] * (1 - fx)[None, :, None] + f[y1][:, x1] * fx[None, :, None]
    out = top * (1 - fy)[:, None, None] + bot * fy[:, None, None]
    return np.clip(np.floor(out + 0.5), 0, 255).astype(np.uint8)


def letterbox(img, new_shape=(640, 640), color=(114, 114, 114), auto=True, scaleFill=False, scaleup=True, stride=32):
    """Resize keeping the aspect ratio, then pad to `new_shape` with `color` (reference utils/datasets.py:1404-1444).
    `auto` / `scaleFill` / `stride` are accepted and — exactly as in the reference, whose branches for them are
    commented out — have no effect.  Returns (image, (ratio_w, ratio_h), (pad_w, pad_h)) with per-side paddings."""
    shape = img.shape[:2]
    if isinstance(new_shape, int):
        new_shape = (new_shape, new_shape)
    r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
    if not scaleup:
        r = min(r, 1.0)
    new_unpad = int(round(shape[1] * r)), int(round(shape[0] * r))
    dw, dh = (new_shape[1] - new_unpad[0]) / 2, (new_shape[0] - new_unpad[1]) / 2
    if shape[::-1] != new_unpad:
        img = resize_bilinear(img, new_unpad)
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    out = np.empty((img.shape[0] + top + bottom, img.shape[1] + left + right, img.shape[2]), np.uint8)
    out[...] = np.asarray(color, np.uint8)
    out[top:top + img.shape[0], left:left + img.shape[1]] = img
    return out, (r, r), (dw, dh)


def _list_images(path):
    p = str(Path(path).absolute())
    if "*" in p:
        files = sorted(glob.glob(p, recursive=True))
    elif os.path.isdir(p):
        files = sorted(glob.glob(os.path.join(p, "*.*")))
    elif os.path.isfile(p):
        files = [p]
    else:
        raise FileNotFoundError(f"{p} does not exist")
    return [f for f in files if f.rsplit(".", 1)[-1].lower() in IMG_FORMATS]


class LoadImages:
    """Iterate the image files of a folder / glob / single path: yields (path, CHW RGB uint8 letterboxed, BGR original,
    None) like the reference's LoadImages.__next__ (utils/datasets.py:205-241).  Video files are out of scope."""

    def __init__(self, path, img_size=640, stride=32):
        self.files = _list_images(path)
        self.nf = len(self.files)
        self.img_size, self.stride, self.mode, self.cap = img_size, stride, "image", None
        if self.nf == 0:
            raise FileNotFoundError(f"no images found in {path} (supported: {IMG_FORMATS})")

    def __iter__(self):
        self.count = 0
        return self

    def __len__(self):
        return self.nf

    def __next__(self):
        if self.count == self.nf:
            raise StopIteration
        path = self.files[self.count]
        self.count += 1
        img0 = imread_bgr(path)
        img = letterbox(img0, self.img_size, stride=self.stride)[0]
        img = np.ascontiguousarray(img[:, :, ::-1].transpose(2, 0, 1))
        return path, img, img0, self.cap


def img2label_paths(img_paths):
    """.../images/x.jpg -> .../labels/x.txt (reference utils/datasets.py:84-87)."""
    sa, sb = os.sep + "images" + os.sep, os.sep + "labels" + os.sep
    return [sb.join(p.rsplit(sa, 1)).rsplit(".", 1)[0] + ".txt" for p in img_paths]


class PairedValSet:
    """RGB + IR validation pairs with YOLO txt labels.  `__getitem__` -> (6xHxW uint8 tensor = cat(rgb, ir) as
    utils/datasets.py:1022-1024, labels (n, 6) [0, cls, cx, cy, w, h] normalised to the letterboxed image, rgb path,
    ((h0, w0), ((ratio, ratio), (pad_w, pad_h))) for scale_coords)."""

    def __init__(self, path_rgb, path_ir, img_size=640, label_paths=None):
        self.rgb, self.ir = _list_images(path_rgb), _list_images(path_ir)
        assert len(self.rgb) == len(self.ir) and self.rgb, f"{len(self.rgb)} RGB vs {len(self.ir)} IR images"
        self.img_size = img_size
        self.label_files = label_paths or img2label_paths(self.rgb)

    def __len__(self):
        return len(self.rgb)

    def __getitem__(self, i):
        a0, b0 = imread_bgr(self.rgb[i]), imread_bgr(self.ir[i])
        a, ratio, pad = letterbox(a0, self.img_size)
        b = letterbox(b0, self.img_size)[0]
        h0, w0 = a0.shape[:2]
        lab = np.zeros((0, 5), np.float32)
        if os.path.isfile(self.label_files[i]):
            with open(self.label_files[i]) as f:
                rows = [ln.split() for ln in f.read().strip().splitlines() if ln.strip()]
            if rows:
                lab = np.array(rows, np.float32).reshape(-1, 5)
        out = np.zeros((len(lab), 6), np.float32)
        if len(lab):                    # normalised xywh of the original image -> normalised xywh of the letterboxed one
            H, W = a.shape[:2]
            out[:, 1] = lab[:, 0]
            out[:, 2] = (lab[:, 1] * w0 * ratio[0] + pad[0]) / W
            out[:, 3] = (lab[:, 2] * h0 * ratio[1] + pad[1]) / H
            out[:, 4] = lab[:, 3] * w0 * ratio[0] / W
            out[:, 5] = lab[:, 4] * h0 * ratio[1] / H
        chw = lambda x: np.ascontiguousarray(x[:, :, ::-1].transpose(2, 0, 1))      # noqa: E731
        img6 = torch.from_numpy(np.concatenate((chw(a), chw(b)), 0))
        return img6, torch.from_numpy(out), self.rgb[i], ((h0, w0), (ratio, pad))

    @staticmethod
    def collate_fn(batch):
        img, label, path, shapes = zip(*batch)
        for i, l in enumerate(label):
            l[:, 0] = i                 # image index inside the batch (reference :1053-1058)
        return torch.stack(img, 0), torch.cat(label, 0), path, shapes


def create_dataloader_rgb_ir(path_rgb, path_ir, imgsz, batch_size, *_, **kw):
    """Reference signature (utils/datasets.py:102-129) reduced to what validation needs: -> (loader, dataset)."""
    ds = PairedValSet(path_rgb, path_ir, imgsz)
    loader = torch.utils.data.DataLoader(ds, batch_size=batch_size, shuffle=False, num_workers=int(kw.get("workers", 0)),
                                         collate_fn=PairedValSet.collate_fn)
    return loader, ds
