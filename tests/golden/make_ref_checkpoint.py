#!/usr/bin/env python3
"""Pickle a checkpoint with the REAL reference classes, exactly as its training loop does, plus the reference's own output
for it — the fixture behind the `attempt_load` drop-in tests.

    python tests/golden/make_ref_checkpoint.py                       # rewrites tests/golden/ref_ckpt_yolov5n_flir_fp16.{pt,npz}
    python tests/golden/make_ref_checkpoint.py --yaml X.yaml --seed S --out /tmp/x.pt [--no-golden]

The reference saves WHOLE pickled `Model` objects: `{'epoch', 'best_fitness', 'model': deepcopy(model).half(), 'ema', 'updates',
'optimizer', 'wandb_id'}` (train.py:424-435), reduced by `strip_optimizer` (utils/general.py:610-624: ema -> model, the
other keys None, epoch -1, half precision) for the files it publishes.  The pickle stream therefore names
`models.yolo_test.Model`, `models.common.Conv`, ... — the import paths this repo re-exports — and restores each object's
__dict__ WITHOUT running this repo's constructors.  The golden output is what the reference's own loading recipe
(models/experimental.py:113-134: `ckpt['model'].float().fuse().eval()`) computes on the CPU for a seeded input.

Runs only in the build container (needs /root/reference); the .pt / .npz it writes are committed test fixtures.
"""
import argparse
import os
import sys
from copy import deepcopy

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

import numpy as np   # noqa: E402
import torch         # noqa: E402

import make_golden as mg                                            # noqa: E402
from icafusion_amd.synth import synth_images, synth_state_dict      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--yaml", default="yolov5n_Transfusion_FLIR.yaml")
    ap.add_argument("--seed", type=int, default=21)
    ap.add_argument("--out", default=os.path.join(HERE, "ref_ckpt_yolov5n_flir_fp16.pt"))
    ap.add_argument("--no-golden", action="store_true")
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--height", type=int, default=320)
    ap.add_argument("--width", type=int, default=352)
    args = ap.parse_args()
    yt, common, general, metrics = mg.import_reference()
    model = yt.Model(os.path.join(mg.REF, "models", "transformer", args.yaml))
    model.load_state_dict(synth_state_dict(model, args.seed))
    model.nc, model.names = model.yaml["nc"], [f"class{i}" for i in range(model.yaml["nc"])]    # train.py:230-236 attach these
    model.gr = 1.0
    ckpt = {"epoch": -1, "best_fitness": None, "model": deepcopy(model).half(), "ema": None, "updates": None,
            "optimizer": None, "wandb_id": None}                      # a published (strip_optimizer'ed) checkpoint
    for p in ckpt["model"].parameters():
        p.requires_grad = False
    torch.save(ckpt, args.out)
    print(args.out, f"{os.path.getsize(args.out) / 2 ** 20:.1f} MiB")
    if args.no_golden:
        return
    ref = torch.load(args.out, map_location="cpu", weights_only=False)["model"].float().fuse().eval()
    rgb, ir = synth_images(args.batch, args.height, args.width, args.seed)
    with torch.no_grad():
        z, logits, raws = ref(rgb, ir)
    out = args.out[:-3] + ".npz"
    np.savez_compressed(out, z=z.numpy(), logits=logits.numpy(), meta=np.asarray([args.batch, args.height, args.width, args.seed]),
                        yaml=np.asarray(args.yaml), names=np.asarray(ref.names), stride=ref.stride.numpy())
    print(out, tuple(z.shape))


if __name__ == "__main__":
    main()
