"""16-bit parity on the configurations that are benchmarked (BASELINE.json configs 2-5 at their per-GPU shard sizes): a slice
of every full-size bf16 / fp16 batch against the fp32 CPU oracle, with the tolerance DERIVED from the reference's own 16-bit
deviation on the same weights and inputs (tools/parity16.py: the oracle evaluated by torch in the same 16-bit type, which is
what `model.half()` does in detect_twostream.py:40 / test.py:73-75), plus the bf16 / fp16 mAP@50 delta through the same
ap_per_class on 16 images at 640x640 of a detector with separated scores (north_star: within 0.1, asserted as stated).  Every measured number is printed and appended to
gpurun_out/parity_16bit_tests.jsonl; profiles/parity_16bit.json is the committed copy of a full tools/parity16.py run.

Why a multiple of the reference's deviation, and why 1.5: both pipelines keep activations in the 16-bit type between layers
and accumulate in fp32, so both carry rounding noise of the same magnitude through ~40 layers; they differ in WHERE they round
(fused launches keep some intermediates in fp32 / LDS; the reference rounds after every torch op and also decodes boxes in the
16-bit type, where pixel coordinates near 640 have a 4 px (bf16) / 0.5 px (fp16) grid).  The HIP path is therefore expected
BELOW the reference's own deviation; 1.5x leaves room for a different noise realisation in the maximum over 10^5-10^6 values.
"""
import json
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
import parity16                                     # noqa: E402

FACTOR = 1.5


def _record(rec):
    print(json.dumps(rec))
    try:
        os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
        with open(os.path.join(REPO, "gpurun_out", "parity_16bit_tests.jsonl"), "a") as f:
            f.write(json.dumps(rec) + "\n")
    except OSError:
        pass


@pytest.mark.parametrize("name", list(parity16.CONFIGS))
def test_full_size_16bit_slice_vs_fp32_oracle(name):
    slow = name.startswith("c5")              # yolov5l at 1280x1280 in fp16 on the host CPU: ~6 minutes for the yardstick alone
    rec = parity16.measure(name, with_reference16=not slow)
    if slow:                                  # ... so it comes from the committed full run of tools/parity16.py (same weights, inputs, image)
        with open(os.path.join(REPO, "profiles", "parity_16bit.json")) as f:
            done = {c["config"]: c for c in json.load(f)["configs"]}
        assert done[name]["images_compared"] == rec["images_compared"]
        rec["reference16_vs_oracle_fp32"] = done[name]["reference16_vs_oracle_fp32"]
        rec["reference16_source"] = "profiles/parity_16bit.json"
    _record(rec)
    hip, ref = rec["hip16_vs_oracle_fp32"], rec["reference16_vs_oracle_fp32"]
    for k in ("box_px_max", "box_px_mean", "score_max", "score_mean"):
        assert hip[k] <= FACTOR * ref[k], f"{name}: {k} = {hip[k]:.4g} exceeds {FACTOR} x the reference's own 16-bit deviation {ref[k]:.4g}"


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_16bit_map50_within_a_tenth_of_the_fp32_oracle(dtype):
    """north_star: mAP@50 within +-0.1 (percent) of the reference on identical inputs - asserted as stated, no multiplier.
    16 images 640x640 through test.py's protocol (conf 0.001, IoU 0.5, the same ap_per_class) on the planted detector of
    tools/parity16.py: scores separated the way a trained detector's are (background ~0.003, objects 0.3-0.85), mAP@50 ~ 80.
    The recipe itself is checked first: the REFERENCE evaluated in the same 16-bit type must stay within 0.1 as well
    (tests/test_oracle_vs_golden.py runs that half on the CPU for a smaller batch)."""
    rec = parity16.measure_map(dtype)
    _record(rec)
    assert rec["images"] >= 16 and rec["objects"] >= 300
    assert rec["map50_oracle_fp32"] > 50.0
    assert abs(rec["map50_delta_reference16"]) <= 0.1, f"recipe not separated: the reference in {dtype} moves mAP@50 by {rec['map50_delta_reference16']}"
    assert abs(rec["map50_delta"]) <= 0.1, f"{dtype}: mAP@50 {rec['map50_hip16']} vs fp32 oracle {rec['map50_oracle_fp32']}"


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_16bit_map_on_ten_pixel_objects_where_box_error_counts(dtype):
    """the recipe above plants 30-60 pixel boxes, which no 16-bit box error can push across IoU 0.5.  RECIPES["p3_small"]
    plants at the P3 DMFF block and reads out on the P3 level's smallest anchor: ~10 x 10 pixel objects, for which two pixels of error cost a
    third of the IoU.  That the metric now depends on localisation is shown by the reference itself: evaluated in bf16 (boxes decoded INTO a
    bf16 tensor: a 2-4 pixel grid beyond x = 256) it keeps mAP@50 but loses several points of mAP@.5:.95.  The HIP path (16-bit features,
    fp32 Detect decode) must keep mAP@50 within 0.1 AND may not lose more mAP@.5:.95 than the reference's own 16-bit mode does."""
    rec = parity16.measure_map(dtype, recipe="p3_small")
    _record(rec)
    box = rec["object_box_px"]
    assert rec["objects"] >= 1000 and 8.0 <= box["w_median"] <= 16.0 and 8.0 <= box["h_median"] <= 16.0
    assert rec["map50_oracle_fp32"] > 50.0
    assert abs(rec["map50_delta"]) <= 0.1, f"{dtype}: mAP@50 {rec['map50_hip16']} vs fp32 oracle {rec['map50_oracle_fp32']} on 10-pixel objects"
    assert rec["map_delta"] >= min(rec["map_delta_reference16"], 0.0) - 0.5, \
        f"{dtype}: mAP@.5:.95 moves by {rec['map_delta']} (HIP) vs {rec['map_delta_reference16']} (the reference in {dtype})"
