"""Fused DMFF block kernels (dmff_fused.hip: icaf_dmff_ln_qkv + icaf_dmff_attn_mlp, 2 launches per iteration) against the fp32
oracle statement of CrossTransformerBlock (reference models/common.py:737-759) and against the per-layer launches they replace
(LayerNorm, QKV / out-projection / MLP GEMMs, cross_attn_kernel: 7 launches) in the same 16-bit type."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import load_golden, sample_idx                         # noqa: E402
from icafusion_amd.engine import Plan                                # noqa: E402
from icafusion_amd.models.common import CrossTransformerBlock, TransformerFusionBlock   # noqa: E402
from icafusion_amd.synth import synth_tensor                         # noqa: E402
from oracle import icaf_oracle as oracle                             # noqa: E402

DEV = "cuda:0"


def make_block(C, heads, loops, seed):
    blk = CrossTransformerBlock(C, C, C, heads, 4, 0.1, 0.1, loops_num=loops).eval()
    sd = {k: synth_tensor("model.20.crosstransformer.0." + k, v.shape, seed=seed) for k, v in blk.state_dict().items()}
    blk.load_state_dict(sd)
    return blk, {"b." + k: v for k, v in sd.items()}


def run_block(blk, tok, B, N, dtype, fused, max_c=512, res32=True):
    blk.fuse_block, blk.fuse_max_c = fused, max_c        # (the plan uses the two-launch kernels up to C = 128 by default; they are built to 512)
    blk.res32 = res32                                     # loops > 1, three-launch form: fp32 token stream between iterations (round 5)
    blk.fuse_fp32 = fused                                 # fp32: the parity instantiation of the same template (C <= 128)
    blk.invalidate()
    plan = Plan(DEV, dtype)
    t = plan.tokens(2, B * N, tok.shape[2])
    t.copy_(tok.to(DEV).to(dtype))
    out = blk.emit_tokens(plan, t, B, N)
    plan.run()
    torch.cuda.synchronize()
    return out.float().cpu(), [l.name for l in plan.launches]


# (C, heads, N, B, loops): the three yolov5s levels, a yolov5n level (dk = 8), yolov5m levels (dk = 24 / 48, C not a power of two),
# ragged token counts, several iterations
SHAPES = [(128, 8, 400, 3, 1), (256, 8, 256, 2, 1), (512, 8, 100, 3, 1), (64, 8, 400, 2, 1), (192, 8, 400, 1, 1), (384, 8, 256, 1, 1),
          (128, 8, 77, 2, 3), (256, 8, 400, 1, 2), (512, 8, 64, 1, 1), (128, 4, 130, 1, 1)]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", SHAPES)
def test_fused_block_vs_oracle_and_per_layer_launches(shape, dtype):
    C, heads, N, B, loops = shape
    blk, sd = make_block(C, heads, loops, seed=C + N)
    blk = blk.to(DEV)
    g = np.random.default_rng(C * 1000 + N)
    tok = torch.from_numpy(g.normal(0.2, 0.8, (2, B * N, C)).astype(np.float32))
    tq = tok.to(dtype).float()                                      # both paths start from the same 16-bit tokens
    rv, ri = oracle.cross_transformer(tq[0].reshape(B, N, C), tq[1].reshape(B, N, C), sd, "b", heads, loops)
    ref = torch.stack((rv.reshape(B * N, C), ri.reshape(B * N, C)))
    fused, names_f = run_block(blk, tok, B, N, dtype, True)
    plain, names_p = run_block(blk, tok, B, N, dtype, False)
    assert names_f == ["dmff_ln_qkv", "dmff_attn_mlp"] * loops and len(names_p) == 7 * loops
    scale = ref.abs().max().item()
    e_f, e_p = (fused - ref).abs().max().item() / scale, (plain - ref).abs().max().item() / scale
    m_f, m_p = (fused - ref).abs().mean().item() / scale, (plain - ref).abs().mean().item() / scale
    print(f"C={C} N={N} B={B} loops={loops} {dtype}: fused max {e_f:.3e} mean {m_f:.3e} | per-layer max {e_p:.3e} mean {m_p:.3e}")
    assert torch.isfinite(fused).all()
    # same rounding points as the per-layer launches (only fp32 summation orders differ): the error against the fp32 oracle must
    # be of the same size, and the two 16-bit results must agree to a few units of the storage type's precision
    assert e_f <= 1.5 * e_p + 1e-4 and m_f <= 1.25 * m_p + 1e-5
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    assert (fused - plain).abs().max().item() / scale <= 24 * ulp


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(256, 8, 256, 2, 1), (512, 8, 100, 3, 1), (256, 8, 400, 1, 2), (512, 8, 64, 1, 1), (256, 4, 77, 3, 3), (512, 16, 130, 2, 2),
                                   (128, 8, 400, 3, 1), (128, 8, 77, 2, 3), (128, 4, 130, 1, 2)])
def test_wide_block_three_launches_vs_oracle_and_per_layer_launches(shape, dtype):
    """C = 256 / 512 (what the plan runs at P4 / P5) and C = 128 (the four-wavefront build): icaf_dmff_wide_ln_qkv + icaf_cross_attention +
    icaf_dmff_wide_proj_mlp per iteration."""
    C, heads, N, B, loops = shape
    blk, sd = make_block(C, heads, loops, seed=C + N)
    blk = blk.to(DEV)
    g = np.random.default_rng(C * 1000 + N + 1)
    tok = torch.from_numpy(g.normal(0.2, 0.8, (2, B * N, C)).astype(np.float32))
    tq = tok.to(dtype).float()
    rv, ri = oracle.cross_transformer(tq[0].reshape(B, N, C), tq[1].reshape(B, N, C), sd, "b", heads, loops)
    ref = torch.stack((rv.reshape(B * N, C), ri.reshape(B * N, C)))
    wide, names_w = run_block(blk, tok, B, N, dtype, True, max_c=64)
    plain, names_p = run_block(blk, tok, B, N, dtype, False)
    assert [n for n in names_w if n != "dmff_proj_mlp_reduce"] == ["dmff_ln_qkv", "cross_attention", "dmff_proj_mlp"] * loops and len(names_p) == 7 * loops
    scale = ref.abs().max().item()
    e_w, e_p = (wide - ref).abs().max().item() / scale, (plain - ref).abs().max().item() / scale
    m_w, m_p = (wide - ref).abs().mean().item() / scale, (plain - ref).abs().mean().item() / scale
    print(f"C={C} N={N} B={B} loops={loops} {dtype}: three launches max {e_w:.3e} mean {m_w:.3e} | per-layer max {e_p:.3e} mean {m_p:.3e}")
    assert torch.isfinite(wide).all()
    assert e_w <= 1.5 * e_p + 1e-4 and m_w <= 1.25 * m_p + 1e-5
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    assert (wide - plain).abs().max().item() / scale <= 24 * ulp


def rounded_reference(tok, sd, heads, loops, dtype, B, N, C, res32=False):
    """oracle.cross_transformer in float64 with the storage roundings of a 16-bit implementation and nothing else: parameters of the
    Linear layers rounded to the storage type (the packed weights), LayerNorm parameters / biases / coefficients in fp32 as the kernels
    hold them, every stored tensor round-tripped through the type."""
    def rt(t):
        return t.to(dtype).double()
    sd64 = {}
    for k, v in sd.items():
        is_w = k.endswith(".weight") and v.dim() == 2
        sd64[k] = rt(v.float()) if is_w else v.double()
    tq = rt(tok)
    rv, ri = oracle.cross_transformer(tq[0].reshape(B, N, C), tq[1].reshape(B, N, C), sd64, "b", heads, loops, store=rt, res32=res32)
    return torch.stack((rv.reshape(B * N, C), ri.reshape(B * N, C)))


# (max, mean) of |kernel - rounded_reference| / max|ref|.  Measured on the MI355X over the five shapes below (printed by the test;
# profiles/r04_dmff_wide_abs_error.txt): bf16 max 3.9e-3 ... 1.1e-2 (three iterations), mean 2.2e-4 ... 5.0e-4; f16 max 6.2e-4 ... 1.4e-3, mean
# 2.8e-5 ... 6.2e-5 — one to three units in the last place of the storage type on the largest elements (the rounding of the attention
# probabilities, fp32 accumulation order, the exp2 / erf approximations).  The bounds are 2 x the largest value measured.
ABS_BOUND = {torch.bfloat16: (2.3e-2, 1.0e-3), torch.float16: (2.8e-3, 1.3e-4)}


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(256, 8, 256, 2, 1), (512, 8, 100, 3, 1), (512, 8, 64, 1, 2), (128, 8, 400, 2, 1), (256, 4, 77, 3, 3)])
def test_wide_block_against_the_oracle_in_absolute_terms(shape, dtype):
    """VERDICT r3 #5 / next #6a: the three-launch kernels (what the bench runs at P4 / P5) held to oracle.cross_transformer DIRECTLY, not to
    their per-layer siblings: the oracle evaluated in float64 with exactly the storage roundings of a 16-bit implementation
    (rounded_reference) leaves only the kernels' own arithmetic error, which is bounded in absolute terms (fractions of max|ref|);
    the distance to the unrounded fp32 oracle is printed beside it."""
    C, heads, N, B, loops = shape
    blk, sd = make_block(C, heads, loops, seed=C + N)
    blk = blk.to(DEV)
    g = np.random.default_rng(C * 1000 + N + 3)
    tok = torch.from_numpy(g.normal(0.2, 0.8, (2, B * N, C)).astype(np.float32))
    ref_r = rounded_reference(tok, sd, heads, loops, dtype, B, N, C, res32=loops > 1)      # (these shapes all take the fp32 token stream when loops > 1)
    tq = tok.to(dtype).float()
    rv, ri = oracle.cross_transformer(tq[0].reshape(B, N, C), tq[1].reshape(B, N, C), sd, "b", heads, loops)
    ref = torch.stack((rv.reshape(B * N, C), ri.reshape(B * N, C)))
    wide, names_w = run_block(blk, tok, B, N, dtype, True, max_c=64)
    assert "dmff_proj_mlp" in names_w and "cross_attention" in names_w
    scale = ref.abs().max().item()
    d = (wide.double() - ref_r).abs()
    e_max, e_mean = d.max().item() / scale, d.mean().item() / scale
    e_fp32 = (wide - ref).abs().max().item() / scale
    print(f"C={C} N={N} B={B} loops={loops} {dtype}: vs float64 oracle with storage roundings max {e_max:.3e} mean {e_mean:.3e} | vs fp32 oracle max {e_fp32:.3e}")
    assert torch.isfinite(wide).all()
    assert e_max <= ABS_BOUND[dtype][0] and e_mean <= ABS_BOUND[dtype][1]


@pytest.mark.parametrize("ksplit", [1, 2, 4])
@pytest.mark.parametrize("shape", [(512, 8, 100, 3, 1), (256, 8, 77, 2, 2), (512, 16, 130, 2, 2)])
def test_wide_block_hidden_split_equals_unsplit(shape, ksplit, monkeypatch):
    """icaf_dmff_wide_proj_mlp_split + icaf_dmff_wide_reduce: the MLP's hidden columns over 2 / 4 workgroups per tile, fp32 partial
    sums added in slice order.  Against the fp32 oracle the error is the unsplit launch's; the two 16-bit results agree to rounding
    (only the fp32 association of the fc2 sum differs); the result is bit-stable from run to run (no atomics, no arrival order)."""
    from icafusion_amd import ops
    C, heads, N, B, loops = shape
    blk, sd = make_block(C, heads, loops, seed=C + N)
    blk = blk.to(DEV)
    g = np.random.default_rng(C * 1000 + N + 2)
    tok = torch.from_numpy(g.normal(0.2, 0.8, (2, B * N, C)).astype(np.float32))
    tq = tok.to(torch.bfloat16).float()
    rv, ri = oracle.cross_transformer(tq[0].reshape(B, N, C), tq[1].reshape(B, N, C), sd, "b", heads, loops)
    ref = torch.stack((rv.reshape(B * N, C), ri.reshape(B * N, C)))
    # (res32 off: the fp32 token stream of loops > 1 is not built for every split — this test is about the split alone)
    monkeypatch.setattr(ops.OPT, "dmff_ksplit", 1)
    base, _ = run_block(blk, tok, B, N, torch.bfloat16, True, max_c=128, res32=False)
    monkeypatch.setattr(ops.OPT, "dmff_ksplit", ksplit)
    got, names_k = run_block(blk, tok, B, N, torch.bfloat16, True, max_c=128, res32=False)
    again, _ = run_block(blk, tok, B, N, torch.bfloat16, True, max_c=128, res32=False)
    assert ("dmff_proj_mlp_reduce" in names_k) == (ksplit > 1)
    assert torch.equal(got, again)
    scale = ref.abs().max().item()
    e_k, e_b = (got - ref).abs().max().item() / scale, (base - ref).abs().max().item() / scale
    print(f"C={C} N={N} ksplit={ksplit}: max err {e_k:.3e} (unsplit {e_b:.3e}), vs unsplit {(got - base).abs().max().item() / scale:.3e}")
    assert e_k <= 1.25 * e_b + 1e-4 and (got - base).abs().max().item() / scale <= 4 * 2.0 ** -8


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("name", ["dmff_c128_20x20_in40x40", "dmff_c256_16x16_in40x40_overlap",
                                  "dmff_c128_20x20_in64x80_rect_loops3", "dmff_c512_10x10_in10x10_identity"])
def test_fused_dmff_block_vs_reference_golden_16bit(name, dtype):
    """The whole TransformerFusionBlock in a 16-bit type with the fused block kernels vs the real reference's fp32 output: the error
    stays that of the per-layer 16-bit launches (the fp32 build is held to 1e-3 in test_gpu_model.py)."""
    g = load_golden(name)
    c, va, ha, batch, h, w, seed, loops = [int(v) for v in g["meta"]]
    blk = TransformerFusionBlock(c, va, ha, loops_num=loops)
    sd = {k: (v if k.endswith("num_batches_tracked") else synth_tensor("model.20." + k, v.shape, seed=seed)) for k, v in blk.state_dict().items()}
    blk.load_state_dict(sd)
    blk.conv1x1_out.bn.eps = 1e-3
    blk = blk.eval().to(DEV)
    blk.compute_dtype = dtype
    rg = np.random.default_rng([seed, 77, c, h, w])
    rgb = torch.from_numpy(rg.normal(0, 1, (batch, c, h, w)).astype(np.float32)).to(DEV)
    ir = torch.from_numpy(rg.normal(0, 1, (batch, c, h, w)).astype(np.float32)).to(DEV)
    errs = {}
    for fused in (True, False):
        blk.crosstransformer[0].fuse_block, blk.crosstransformer[0].fuse_max_c = fused, 512
        blk.invalidate()
        out = blk([rgb, ir]).float().cpu()
        got = out.reshape(-1)[torch.from_numpy(sample_idx(out.numel(), 200, 8192))].numpy()
        errs[fused] = np.abs(got - g["out"]).max() / max(1.0, np.abs(g["out"]).max())
    print(f"{name} {dtype}: fused {errs[True]:.3e}, per-layer {errs[False]:.3e}")
    assert errs[True] <= 1.5 * errs[False] + 1e-4


# ---- the fp32 INSTANTIATION of the fused kernels: a direct oracle bound ----------------------------------------
# The benchmarked P3 path runs dmff_ln_qkv_kernel / dmff_attn_mlp_kernel in bf16, where its parity could only be stated relative
# to the per-layer launches (same rounding points).  The same template is instantiated for fp32 (v_mfma_f32_32x32x2_f32, fp32 tiles,
# erff) wherever its LDS plan fits (C <= 128): indexing, masking, the online softmax, the LayerNorm-from-registers, the chunked MLP
# and the weight stream are then held to the reference itself at fp32 accuracy.
@pytest.mark.parametrize("shape", [(128, 8, 400, 3, 1), (64, 8, 400, 2, 1), (128, 8, 77, 2, 3), (128, 4, 130, 1, 1), (128, 8, 100, 2, 2)])
def test_fp32_fused_block_vs_oracle(shape):
    C, heads, N, B, loops = shape
    blk, sd = make_block(C, heads, loops, seed=C + N)
    blk = blk.to(DEV)
    g = np.random.default_rng(C * 1000 + N)
    tok = torch.from_numpy(g.normal(0.2, 0.8, (2, B * N, C)).astype(np.float32))
    rv, ri = oracle.cross_transformer(tok[0].reshape(B, N, C), tok[1].reshape(B, N, C), sd, "b", heads, loops)
    ref = torch.stack((rv.reshape(B * N, C), ri.reshape(B * N, C)))
    fused, names_f = run_block(blk, tok, B, N, torch.float32, True)
    plain, names_p = run_block(blk, tok, B, N, torch.float32, False)
    assert names_f == ["dmff_ln_qkv", "dmff_attn_mlp"] * loops and len(names_p) == 7 * loops
    scale = ref.abs().max().item()
    e_f, e_p = (fused - ref).abs().max().item(), (plain - ref).abs().max().item()
    print(f"fp32 C={C} N={N} B={B} loops={loops}: fused max abs {e_f:.3e} (per-layer {e_p:.3e}), |ref| max {scale:.3f}")
    assert e_f <= 1e-4 * max(1.0, scale)                 # north_star's fp32 1e-3, with a decade to spare


@pytest.mark.parametrize("name", ["dmff_c128_20x20_in40x40", "dmff_c128_20x20_in64x80_rect_loops3"])
def test_fp32_fused_dmff_block_vs_reference_golden(name):
    """Whole TransformerFusionBlock, fp32, block iterations through the FUSED kernels, vs the real reference's recorded output
    (the two C = 128 goldens: square input; rectangular input with three shared-weight iterations) at 1e-3."""
    g = load_golden(name)
    c, va, ha, batch, h, w, seed, loops = [int(v) for v in g["meta"]]
    blk = TransformerFusionBlock(c, va, ha, loops_num=loops)
    sd = {k: (v if k.endswith("num_batches_tracked") else synth_tensor("model.20." + k, v.shape, seed=seed)) for k, v in blk.state_dict().items()}
    blk.load_state_dict(sd)
    blk.conv1x1_out.bn.eps = 1e-3
    blk = blk.eval().to(DEV)
    rg = np.random.default_rng([seed, 77, c, h, w])
    rgb = torch.from_numpy(rg.normal(0, 1, (batch, c, h, w)).astype(np.float32)).to(DEV)
    ir = torch.from_numpy(rg.normal(0, 1, (batch, c, h, w)).astype(np.float32)).to(DEV)
    ct = blk.crosstransformer[0]
    ct.fuse_block, ct.fuse_fp32, ct.fuse_max_c = True, True, 128          # (the two-launch kernels: not the default at C = 128 since round 4)
    blk.invalidate()
    out = blk([rgb, ir]).float().cpu()
    names = [l.name for pl in blk.__dict__["_plans"].values() for l in pl.launches]
    assert names.count("dmff_attn_mlp") == loops and "cross_attention" not in names          # the fused fp32 kernels ran
    got = out.reshape(-1)[torch.from_numpy(sample_idx(out.numel(), 200, 8192))].numpy()
    err = np.abs(got - g["out"]).max()
    print(f"{name} fp32 fused: max abs error {err:.3e}, |out| max {np.abs(g['out']).max():.3f}")
    assert err <= 1e-3 * max(1.0, np.abs(g["out"]).max())
    assert err <= 2e-4                                    # measured ~1e-5: a wrong index or mask is orders of magnitude above this


# ---- round 5: the fp32 INSTANTIATION of the THREE-launch kernels (dmff_wide.hip at C = 128) -------------------------------------------
# Since round 4 every yolov5s level runs dmff_wide_ln_qkv + cross_attention + dmff_wide_proj_mlp by default; their parity with the
# reference was a 16-bit bound (storage roundings at 2 x the measured error).  The same templates are instantiated for fp32 at C = 128
# (WG<4, 2>: four wavefronts, 128-channel passes; v_mfma_f32_32x32x2_f32, fp32 tiles and fragment-major fp32 weights, erff), so the
# DEFAULT path's indexing, row masks, weight stream, LayerNorm-from-registers and hidden-chunk loop are held to the oracle and to the real
# reference's recorded outputs at fp32 accuracy.
@pytest.mark.parametrize("shape", [(128, 8, 400, 3, 1), (128, 8, 77, 2, 3), (128, 4, 130, 1, 1), (128, 8, 100, 2, 2), (128, 16, 64, 1, 1)])
def test_fp32_wide_block_three_launches_vs_oracle(shape):
    C, heads, N, B, loops = shape
    blk, sd = make_block(C, heads, loops, seed=C + N + 7)
    blk = blk.to(DEV)
    g = np.random.default_rng(C * 1000 + N + 7)
    tok = torch.from_numpy(g.normal(0.2, 0.8, (2, B * N, C)).astype(np.float32))
    rv, ri = oracle.cross_transformer(tok[0].reshape(B, N, C), tok[1].reshape(B, N, C), sd, "b", heads, loops)
    ref = torch.stack((rv.reshape(B * N, C), ri.reshape(B * N, C)))
    wide, names_w = run_block(blk, tok, B, N, torch.float32, True, max_c=64)          # fuse_max_c = 64 (the default): C = 128 takes the wide path
    plain, names_p = run_block(blk, tok, B, N, torch.float32, False)
    assert names_w == ["dmff_ln_qkv", "cross_attention", "dmff_proj_mlp"] * loops and len(names_p) == 7 * loops
    scale = ref.abs().max().item()
    e_w, e_p = (wide - ref).abs().max().item(), (plain - ref).abs().max().item()
    print(f"fp32 wide C={C} N={N} B={B} loops={loops}: three launches max abs {e_w:.3e} (per-layer {e_p:.3e}), |ref| max {scale:.3f}")
    assert e_w <= 1e-4 * max(1.0, scale)


@pytest.mark.parametrize("name", ["dmff_c128_20x20_in40x40", "dmff_c128_20x20_in64x80_rect_loops3"])
def test_fp32_wide_dmff_block_vs_reference_golden(name):
    """Whole TransformerFusionBlock, fp32, block iterations through the fp32 instantiation of the THREE-launch kernels (what the 16-bit
    plans run by default), vs the real reference's recorded output: the two C = 128 goldens at <= 2e-4 absolute."""
    g = load_golden(name)
    c, va, ha, batch, h, w, seed, loops = [int(v) for v in g["meta"]]
    blk = TransformerFusionBlock(c, va, ha, loops_num=loops)
    sd = {k: (v if k.endswith("num_batches_tracked") else synth_tensor("model.20." + k, v.shape, seed=seed)) for k, v in blk.state_dict().items()}
    blk.load_state_dict(sd)
    blk.conv1x1_out.bn.eps = 1e-3
    blk = blk.eval().to(DEV)
    rg = np.random.default_rng([seed, 77, c, h, w])
    rgb = torch.from_numpy(rg.normal(0, 1, (batch, c, h, w)).astype(np.float32)).to(DEV)
    ir = torch.from_numpy(rg.normal(0, 1, (batch, c, h, w)).astype(np.float32)).to(DEV)
    ct = blk.crosstransformer[0]
    ct.fuse_block, ct.fuse_fp32, ct.fuse_max_c, ct.fuse_wide = True, True, 64, True
    blk.invalidate()
    out = blk([rgb, ir]).float().cpu()
    names = [l.name for pl in blk.__dict__["_plans"].values() for l in pl.launches]
    assert names.count("dmff_proj_mlp") == loops and names.count("dmff_ln_qkv") == loops and names.count("cross_attention") == loops
    got = out.reshape(-1)[torch.from_numpy(sample_idx(out.numel(), 200, 8192))].numpy()
    err = np.abs(got - g["out"]).max()
    print(f"{name} fp32 three-launch: max abs error {err:.3e}, |out| max {np.abs(g['out']).max():.3f}")
    assert err <= 2e-4



# ---- round 5: the token stream between the iterations of a block in fp32 (loops > 1) --------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(128, 8, 400, 2, 3), (256, 8, 256, 2, 3), (512, 8, 100, 2, 3), (256, 4, 77, 3, 2)])
def test_wide_block_fp32_token_stream_between_iterations(shape, dtype):
    """loops > 1 (BASELINE config 4 runs three): icaf_dmff_wide_proj_mlp (and the split + reduce pair at P5) keep x -> x_att -> x' in fp32 from
    one iteration to the next (icaf_dmff_args.x32 / y32) instead of rounding the stream to 16 bits twice per iteration.  Against the fp32
    oracle the error must not grow beyond the 16-bit stream's and should shrink; against the float64 oracle with the SAME storage pattern
    (oracle.cross_transformer(res32=True)) it stays within the absolute bound of the kernels' own arithmetic."""
    C, heads, N, B, loops = shape
    blk, sd = make_block(C, heads, loops, seed=C + N + 11)
    blk = blk.to(DEV)
    g = np.random.default_rng(C * 1000 + N + 11)
    tok = torch.from_numpy(g.normal(0.2, 0.8, (2, B * N, C)).astype(np.float32))
    tq = tok.to(dtype).float()
    rv, ri = oracle.cross_transformer(tq[0].reshape(B, N, C), tq[1].reshape(B, N, C), sd, "b", heads, loops)
    ref = torch.stack((rv.reshape(B * N, C), ri.reshape(B * N, C)))
    ref_r = rounded_reference(tok, sd, heads, loops, dtype, B, N, C, res32=True)
    on, names = run_block(blk, tok, B, N, dtype, True, max_c=64, res32=True)
    off, _ = run_block(blk, tok, B, N, dtype, True, max_c=64, res32=False)
    assert names.count("dmff_proj_mlp") == loops
    scale = ref.abs().max().item()
    e_on, e_off = (on - ref).abs().max().item() / scale, (off - ref).abs().max().item() / scale
    m_on, m_off = (on - ref).abs().mean().item() / scale, (off - ref).abs().mean().item() / scale
    d = (on.double() - ref_r).abs()
    print(f"C={C} N={N} loops={loops} {dtype}: fp32 stream max {e_on:.3e} mean {m_on:.3e} | 16-bit stream max {e_off:.3e} mean {m_off:.3e} | vs same-storage float64 oracle max {d.max().item() / scale:.3e}")
    assert torch.isfinite(on).all()
    assert m_on <= 1.02 * m_off + 1e-6 and e_on <= 1.25 * e_off + 1e-4
    assert d.max().item() / scale <= ABS_BOUND[dtype][0] and d.mean().item() / scale <= ABS_BOUND[dtype][1]
