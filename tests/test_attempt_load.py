"""The drop-in's primary entry point: `attempt_load` of checkpoints pickled by the REAL reference (models/experimental.py:113-134,
train.py:424-435 — whole `Model` objects whose __dict__ is restored without running this repo's constructors).

CPU: the committed fixture (tests/golden/ref_ckpt_yolov5n_flir_fp16.pt, written by tests/golden/make_ref_checkpoint.py with the
reference's own classes) un-pickles onto the HIP-backed classes and builds execution plans; where /root/reference exists a
fresh yolov5s checkpoint is pickled by the reference in a subprocess and loaded the same way.
GPU: `attempt_load(fixture)(rgb, ir)` reproduces the reference's own output for that checkpoint (fp32 1e-3)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from helpers import GOLDEN, REPO
from icafusion_amd.models.common import Conv, HipModule, TransformerFusionBlock
from icafusion_amd.models.yolo import Detect, Model
from icafusion_amd.synth import synth_images

FIXTURE = os.path.join(GOLDEN, "ref_ckpt_yolov5n_flir_fp16.pt")
REF = "/root/reference"


def _attempt_load(path, device="cpu"):
    from models.experimental import attempt_load          # the import path detect_twostream.py:36 / test.py:64 use
    return attempt_load(path, map_location=device)


def _check_loaded(m, yaml_names):
    assert type(m) is Model and isinstance(m, HipModule) and not m.training
    assert not any(hasattr(c, "bn") for c in m.modules() if type(c) is Conv), "attempt_load returns a fused model"
    assert next(m.parameters()).dtype == torch.float32
    assert isinstance(m.model[-1], Detect) and m.model[-1].anchor_grid.dtype == torch.float32
    assert len(m.names) == m.yaml["nc"] and [float(s) for s in m.stride] == [8.0, 16.0, 32.0]
    # nothing the reference does not know is in the un-pickled __dict__ ...
    assert "compute_dtype" not in m.__dict__ and "fuse_tail" not in next(b for b in m.modules() if isinstance(b, TransformerFusionBlock)).__dict__
    # ... and every execution switch still resolves (class-level defaults) and can be set per instance
    assert (m.compute_dtype, m.autotune, m.use_graph, m.pair_streams, m.branch_dmff, m.fold_upsample, m.static_outputs) == \
           (None, False, False, True, True, False, False)
    for dtype in (torch.float32, torch.bfloat16):
        plan = m.build_plan(2, 320, 352, "cpu", dtype)
        names = [l.name for l in plan.launches]
        assert len(names) > 40 and names[-1] == ("detect_decode" if dtype == torch.float32 else "detect_conv+decode")
        assert names.count("cross_attention") + names.count("dmff_attn_mlp") == 3
        assert ("dmff_attn_mlp" in names or "dmff_proj_mlp" in names) == (dtype != torch.float32)     # 16-bit: the fused block kernels (two- or three-launch form)
    m.compute_dtype, m.use_graph = torch.bfloat16, True           # what test.py / detect_twostream.py do after loading
    assert m.__dict__["compute_dtype"] is torch.bfloat16
    return m


def test_attempt_load_reference_pickle_fixture_cpu():
    m = _check_loaded(_attempt_load(FIXTURE), None)
    assert m.yaml["nc"] == 3 and m.names == ["class0", "class1", "class2"]
    sd = m.state_dict()
    assert "model.20.crosstransformer.0.ln_input.weight" in sd and "model.20.crosstransformer.0.mlp.0.weight" in sd   # dead params kept
    assert "model.0.conv.bias" in sd and "model.0.bn.weight" not in sd                                                    # fused


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree only exists in the build container")
def test_attempt_load_fresh_reference_pickle_cpu(tmp_path):
    """Pickle yolov5s_Transfusion_kaist with the real reference (subprocess: its `models` package shadows ours), load it here."""
    out = tmp_path / "last.pt"
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    subprocess.run([sys.executable, os.path.join(GOLDEN, "make_ref_checkpoint.py"), "--yaml", "yolov5s_Transfusion_kaist.yaml",
                    "--seed", "5", "--out", str(out), "--no-golden"], check=True, cwd=str(tmp_path), env=env, capture_output=True)
    m = _check_loaded(_attempt_load(str(out)), None)
    names = [l.name for l in m.build_plan(2, 320, 320, "cpu", torch.bfloat16).launches]
    assert names[0] == "stem+conv3x3s2+1x1" and names[1] == "bottleneck+cv3"      # the fused yolov5s launches survive un-pickling


def test_ensemble_of_two_checkpoints_cpu():
    from models.experimental import Ensemble, attempt_load
    e = attempt_load([FIXTURE, FIXTURE], map_location="cpu")
    assert isinstance(e, Ensemble) and len(e) == 2 and e.names == e[-1].names


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_attempt_load_forward_matches_reference_output(dtype):
    g = np.load(FIXTURE[:-3] + ".npz", allow_pickle=False)
    batch, h, w, seed = [int(v) for v in g["meta"]]
    m = _attempt_load(FIXTURE, "cuda:0")
    assert next(m.parameters()).is_cuda
    rgb, ir = synth_images(batch, h, w, seed)
    if dtype != torch.float32:
        m.compute_dtype = dtype
    z, logits, raws = m(rgb.cuda(), ir.cuda())
    z, ref = z.float().cpu().numpy(), g["z"]
    assert z.shape == ref.shape
    box = np.abs(z[..., :4] - ref[..., :4]).max()
    conf = np.abs(z[..., 4:] - ref[..., 4:]).max()
    print(f"attempt_load {dtype}: max box err {box:.4g} px, max score err {conf:.3g}")
    if dtype == torch.float32:                 # north_star: fp32 1e-3 (boxes relative to the coordinate scale)
        assert box <= 1e-3 * max(1.0, np.abs(ref[..., :4]).max()) and conf <= 1e-3
        assert np.abs(logits.cpu().numpy() - g["logits"]).max() <= 1e-3 * max(1.0, np.abs(g["logits"]).max())
    else:                                      # the checkpoint's own storage type: see tests/test_gpu_parity16.py for the bound
        assert box <= 2.0 and conf <= 5e-3
