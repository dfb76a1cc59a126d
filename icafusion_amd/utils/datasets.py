"""Image loading for the two-stream front ends, without OpenCV (cv2 is not in this image).

Same entry points and return conventions as the reference's utils/datasets.py for the inference / validation paths:
`letterbox` (:1404-1444 — note the reference's `auto` / `scaleFill` branches are commented out, so every image is padded
to the full `new_shape`), `LoadImages` (:172-241, image files only) and the paired RGB+IR validation set that yields the
6-channel uint8 batches `test.py:115-123` consumes, following LoadMultiModalImagesAndLabels' evaluation protocol
(:690-1024): longest side resized to `img_size` (:1116-1122), RECTANGULAR batches — images sorted by aspect ratio, one
letterbox shape per batch, `ceil(shape * img_size / stride + pad) * stride` (:826-849; test.py:100 passes rect=True,
pad=0.5, which makes KAIST's 512x640 frames 544x672 batches) — `scaleup=False`, labels re-normalised to the letterboxed
image (:961-986), label files found by replacing the `visible` / `infrared` path component with `labels` (:391-401).
Not carried over: augmentation / mosaic (training), the label cache file, image caching.

Decoding goes through PIL.  Up-sampling is a half-pixel-centre bilinear filter evaluated in float32 and rounded to nearest
(cv2.INTER_LINEAR's geometry; OpenCV evaluates it in 11-bit fixed point, so results can differ by one grey level);
down-sampling is the exact pixel-area average (cv2.INTER_AREA's definition, float32).  Arrays are BGR HWC uint8 like
cv2.imread's, so downstream code is unchanged."""
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


def resize_area(img, new_wh):
    """(H, W, C) uint8 -> (new_h, new_w, C) uint8 by pixel-area averaging (the definition of cv2.INTER_AREA for
    down-scaling, which load_image_rgb_ir uses when r < 1: utils/datasets.py:1119-1122): output pixel j covers the input
    interval [j * s, (j + 1) * s), s = n_in / n_out, and averages the input pixels weighted by their overlap with it."""
    h, w = img.shape[:2]
    nw, nh = new_wh
    if (nw, nh) == (w, h):
        return img

    def weights(n_in, n_out):
        s = n_in / n_out
        lo = np.arange(n_out, dtype=np.float64) * s
        hi = lo + s
        px = np.arange(n_in, dtype=np.float64)
        ov = np.clip(np.minimum(hi[:, None], px[None] + 1.0) - np.maximum(lo[:, None], px[None]), 0.0, None)
        return (ov / s).astype(np.float32)                       # (n_out, n_in), rows sum to 1
    wy, wx = weights(h, nh), weights(w, nw)
    f = img.astype(np.float32)
    out = np.einsum("oh,hwc->owc", wy, f, optimize=True)
    out = np.einsum("pw,owc->opc", wx, out, optimize=True)
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
    """.../visible/<split>/x.jpg or .../infrared/<split>/x.jpg -> .../labels/<split>/x.txt: the first `visible` (else
    `infrared`) in the path becomes `labels`, the extension `txt` (reference utils/datasets.py:391-401).  The shipped
    data/multispectral/*.yaml files use exactly this layout."""
    out = []
    for x in img_paths:
        parts = x.split("/")
        if "visible" in parts:
            sa = "visible"
        elif "infrared" in parts:
            sa = "infrared"
        else:
            raise ValueError(f"{x}: paired datasets keep images under a 'visible' or 'infrared' directory and labels under "
                             "'labels' (reference utils/datasets.py:391-401)")
        out.append("txt".join(x.replace(sa, "labels", 1).rsplit(x.split(".")[-1], 1)))
    return out


def _image_files(path):
    """Directory (recursive), glob, or a text file listing image paths ('./' = relative to the list) — :709-722."""
    if isinstance(path, (list, tuple)):
        return sorted(f for p in path for f in _image_files(p))
    p = Path(path)
    if p.is_dir():
        files = glob.glob(str(p / "**" / "*.*"), recursive=True)
    elif p.is_file() and p.suffix.lower()[1:] not in IMG_FORMATS:
        parent = str(p.parent) + os.sep
        with open(p) as f:
            files = [ln.replace("./", parent) if ln.startswith("./") else ln for ln in f.read().strip().splitlines()]
    else:
        return _list_images(path)
    return sorted(f for f in files if f.rsplit(".", 1)[-1].lower() in IMG_FORMATS)


def _read_labels(path):
    if not os.path.isfile(path):
        return None
    with open(path) as f:
        rows = [ln.split() for ln in f.read().strip().splitlines() if ln.strip()]
    lab = np.array(rows, np.float32).reshape(-1, 5) if rows else np.zeros((0, 5), np.float32)
    if len(lab) and ((lab < 0).any() or (lab[:, 1:] > 1).any()):
        raise ValueError(f"{path}: labels must be non-negative, normalised [cls cx cy w h] rows (reference :1049-1052)")
    return lab


class PairedValSet:
    """RGB + IR validation pairs with YOLO txt labels.  `__getitem__` -> (6xHxW uint8 tensor = cat(rgb, ir) as
    utils/datasets.py:1022-1024, labels (n, 6) [0, cls, cx, cy, w, h] normalised to the letterboxed image, rgb path,
    ((h0, w0), ((h / h0, w / w0), (pad_w, pad_h))) for scale_coords)."""

    def __init__(self, path_rgb, path_ir, img_size=640, batch_size=16, rect=False, pad=0.0, stride=32, single_cls=False,
                 label_paths=None):
        self.rgb, self.ir = _image_files(path_rgb), _image_files(path_ir)
        if not self.rgb or len(self.rgb) != len(self.ir):
            raise FileNotFoundError(f"{len(self.rgb)} RGB images under {path_rgb} vs {len(self.ir)} IR images under {path_ir}")
        self.img_size, self.rect, self.stride = img_size, bool(rect), stride
        self.label_files = list(label_paths) if label_paths is not None else img2label_paths(self.rgb)
        labels = [_read_labels(f) for f in self.label_files]
        self.n_missing = sum(l is None for l in labels)
        if self.n_missing == len(labels):
            # the reference prints "No labels found" and asserts (:781) unless augmenting: validation without a single label
            # file means the paths are wrong, and every metric would silently come out as zero
            raise FileNotFoundError(f"no label file found for any of {len(labels)} images, e.g. {self.label_files[0]}")
        self.labels = [np.zeros((0, 5), np.float32) if l is None else l for l in labels]
        if single_cls:
            for l in self.labels:
                l[:, 0] = 0
        n = len(self.rgb)
        self.batch = np.floor(np.arange(n) / batch_size).astype(np.int64)      # batch index of each image (:800-803)
        self.batch_shapes = None
        if self.rect:
            from PIL import Image
            shapes = []
            for f in self.rgb:
                with Image.open(f) as im:
                    shapes.append(im.size)                                    # (w, h), header only
            s = np.array(shapes, dtype=np.float64)
            ar = s[:, 1] / s[:, 0]                                            # aspect ratio h / w
            order = ar.argsort()
            self.rgb = [self.rgb[i] for i in order]
            self.ir = [self.ir[i] for i in order]                             # pairs share their shape (:851-859 sorts IR by its own)
            self.label_files = [self.label_files[i] for i in order]
            self.labels = [self.labels[i] for i in order]
            ar = ar[order]
            nb = int(self.batch[-1]) + 1
            bshapes = [[1.0, 1.0]] * nb
            for i in range(nb):
                ari = ar[self.batch == i]
                mini, maxi = ari.min(), ari.max()
                if maxi < 1:
                    bshapes[i] = [maxi, 1.0]
                elif mini > 1:
                    bshapes[i] = [1.0, 1.0 / mini]
            self.batch_shapes = np.ceil(np.array(bshapes) * img_size / stride + pad).astype(np.int64) * stride     # (h, w)

    def __len__(self):
        return len(self.rgb)

    def __getitem__(self, i):
        a, b = imread_bgr(self.rgb[i]), imread_bgr(self.ir[i])
        h0, w0 = a.shape[:2]
        r = self.img_size / max(h0, w0)                 # longest side -> img_size (load_image_rgb_ir, :1116-1122)
        if r != 1:
            resize = resize_area if r < 1 else resize_bilinear
            a, b = resize(a, (int(w0 * r), int(h0 * r))), resize(b, (int(w0 * r), int(h0 * r)))
        h, w = a.shape[:2]
        shape = tuple(int(v) for v in self.batch_shapes[self.batch[i]]) if self.rect else self.img_size
        a, ratio, pad = letterbox(a, shape, auto=False, scaleup=False)
        b = letterbox(b, shape, auto=False, scaleup=False)[0]
        lab = self.labels[i]
        out = np.zeros((len(lab), 6), np.float32)
        if len(lab):        # normalised xywh of the file -> pixel xyxy of the letterboxed image -> normalised xywh of it (:961-986)
            H, W = a.shape[:2]
            gw, gh = ratio[0] * w, ratio[1] * h
            x1, y1 = gw * (lab[:, 1] - lab[:, 3] / 2) + pad[0], gh * (lab[:, 2] - lab[:, 4] / 2) + pad[1]
            x2, y2 = gw * (lab[:, 1] + lab[:, 3] / 2) + pad[0], gh * (lab[:, 2] + lab[:, 4] / 2) + pad[1]
            out[:, 1] = lab[:, 0]
            out[:, 2], out[:, 3] = (x1 + x2) / 2 / W, (y1 + y2) / 2 / H
            out[:, 4], out[:, 5] = (x2 - x1) / W, (y2 - y1) / H
        chw = lambda x: np.ascontiguousarray(x[:, :, ::-1].transpose(2, 0, 1))      # noqa: E731
        img6 = torch.from_numpy(np.concatenate((chw(a), chw(b)), 0))
        return img6, torch.from_numpy(out), self.rgb[i], ((h0, w0), ((h / h0, w / w0), pad))

    @staticmethod
    def collate_fn(batch):
        img, label, path, shapes = zip(*batch)
        for i, l in enumerate(label):
            l[:, 0] = i                 # image index inside the batch (reference :1053-1058)
        return torch.stack(img, 0), torch.cat(label, 0), path, shapes


def create_dataloader_rgb_ir(path1, path2, imgsz, batch_size, stride=32, opt=None, hyp=None, augment=False, cache=False, pad=0.0,
                             rect=False, rank=-1, world_size=1, workers=0, image_weights=False, quad=False, prefix="", sampler=None):
    """Reference signature (utils/datasets.py:102-129) -> (loader, dataset); test.py:100 calls it with rect=True, pad=0.5.
    augment / hyp / image_weights / quad belong to training and must be off."""
    if augment or image_weights or quad:
        raise NotImplementedError("training-time loading (augment / image_weights / quad) is outside the inference hot path")
    ds = PairedValSet(path1, path2, imgsz, batch_size, rect=rect, pad=pad, stride=int(stride),
                      single_cls=bool(getattr(opt, "single_cls", False)))
    batch_size = min(batch_size, len(ds))
    loader = torch.utils.data.DataLoader(ds, batch_size=batch_size, shuffle=False, num_workers=int(workers),
                                         collate_fn=PairedValSet.collate_fn)
    return loader, ds
