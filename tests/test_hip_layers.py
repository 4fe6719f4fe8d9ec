"""GPU: every HIP kernel, called through the C ABI, against (a) the golden outputs the reference's own
layer modules produced (tests/golden/f3_kats.*) and (b) the oracle on further seeded shapes.

Tolerances (relative to max|expected|, see hip_helpers.LAYER_TOL): fp32 2e-5 (exact-fp32 MFMA, only the
summation order differs from oneDNN), fp16 4e-3, bf16 3e-2 (8-bit mantissa operands, fp32 accumulate)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import i3d_oracle as oracle  # noqa: E402
import hip_helpers as hh  # noqa: E402
from af_mi355x import synth  # noqa: E402
from test_oracle import _kat_input, _kat_state, run_oracle_kat  # noqa: E402

pytestmark = pytest.mark.gpu
DTYPES = ["f32", "f16", "bf16"]


def _close(got, want, dtype, what, scale=1.0):
    want = want if torch.is_tensor(want) else torch.from_numpy(np.asarray(want))
    got = got.float().cpu().reshape(want.shape)
    ref = want.abs().max().item() + 1e-12
    err = (got - want).abs().max().item()
    assert err <= hh.LAYER_TOL[dtype] * scale * ref, "%s [%s]: max|d|=%.3e vs max|ref|=%.3e" % (what, dtype, err, ref)


def hip_block(x_ncdhw, sd, p, stride, dtype, fuse_shortcut=True):
    x = hh.to_ndhwc(x_ncdhw, dtype)
    wa = sd[p + ".branch2.a.weight"]
    tk = wa.shape[2]
    a = hh.conv_bn_act(x, wa, *hh.fold_bn(sd, p + ".branch2.a_bn"), (1, 1, 1), (tk // 2, 0, 0), True, dtype)
    b = hh.conv_bn_act(a, sd[p + ".branch2.b.weight"], *hh.fold_bn(sd, p + ".branch2.b_bn"), (1, stride, stride),
                       (0, 1, 1), True, dtype)
    if (p + ".branch1.weight") in sd and fuse_shortcut:       # what the engine does: one launch for c + branch1
        out = hh.conv_dual(b, sd[p + ".branch2.c.weight"], hh.fold_bn(sd, p + ".branch2.c_bn"), x,
                           sd[p + ".branch1.weight"], hh.fold_bn(sd, p + ".branch1_bn"), (1, stride, stride), dtype)
        return hh.to_ncdhw(out)
    if (p + ".branch1.weight") in sd:
        sc = hh.conv_bn_act(x, sd[p + ".branch1.weight"], *hh.fold_bn(sd, p + ".branch1_bn"), (1, stride, stride),
                            (0, 0, 0), False, dtype)
    else:
        sc = x
    out = hh.conv_bn_act(b, sd[p + ".branch2.c.weight"], *hh.fold_bn(sd, p + ".branch2.c_bn"), (1, 1, 1), (0, 0, 0),
                         True, dtype, residual=sc)
    return hh.to_ncdhw(out)


def hip_stem(x_ncdhw, sd, p, dtype, fused=None):
    xd = x_ncdhw.cuda()
    n, _, t, h, w = xd.shape
    sin = hh.pack_input_f32(xd, dtype)
    scale, shift = hh.fold_bn(sd, p + ".bn")
    if fused is None:
        fused = "rgb3" if dtype != "f32" else False      # what the engine does: one K-packed launch for conv + BN + ReLU + pool
    if fused == "rgb3":
        return hh.to_ncdhw(hh.stem3_conv_pool(xd, sd[p + ".conv.weight"], scale, shift, dtype))
    if fused:
        return hh.to_ncdhw(hh.stem_conv_pool(sin, (n, t, h, w), sd[p + ".conv.weight"], scale, shift, dtype))
    y = hh.stem_conv(sin, (n, t, h, w), sd[p + ".conv.weight"], scale, shift, dtype)
    y = hh.maxpool(y, (1, 3, 3), (1, 2, 2), (0, 1, 1), dtype)
    return hh.to_ncdhw(y)


def run_hip_kat(case, dtype):
    sd, x, name, kind = _kat_state(case), _kat_input(case), case["name"], case["kind"]
    if kind == "stem":
        return hip_stem(x, sd, name, dtype)
    if kind == "block":
        return hip_block(x, sd, name, case["stride"], dtype)
    if kind == "maxpool":
        return hh.to_ncdhw(hh.maxpool(hh.to_ndhwc(x, dtype), case["kernel"], case["stride"], case["pad"], dtype))
    if kind == "head":
        _, logits = hh.avgpool_fc(hh.to_ndhwc(x, dtype), case["pool"], sd[name + ".projection.weight"],
                                  sd[name + ".projection.bias"], dtype)
        return logits.cpu()
    if kind == "fuse":
        k, a = case["kernel"], case["alpha"]
        y = hh.conv_bn_act(hh.to_ndhwc(x, dtype), sd[name + ".conv_f2s.weight"], *hh.fold_bn(sd, name + ".bn"),
                           (a, 1, 1), (k // 2, 0, 0), True, dtype)
        return hh.to_ncdhw(y)
    raise KeyError(kind)


@pytest.mark.parametrize("dtype", DTYPES)
def test_golden_kats(golden_f3, dtype):
    cases, arrays = golden_f3
    for case in cases:
        want = torch.from_numpy(arrays[case["name"] + "_out"])
        if dtype != "f32" and case["kind"] == "maxpool":
            want = want.to(hh.TORCH_DT[dtype]).float()          # max-pool of rounded inputs is exact
        got = run_hip_kat(case, dtype)
        _close(got, want, dtype, case["name"], scale=3.0 if case["kind"] in ("block", "stem") else 1.0)


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_stem_unfused_path_and_odd_sizes(golden_f3, dtype):
    """16-bit stem as two launches (conv, pool) on the golden cases; and the fused launch on odd conv heights/widths
    (last pooled row/column sees a 2-wide window) against the oracle."""
    cases, arrays = golden_f3
    for case in cases:
        if case["kind"] == "stem":
            got = hip_stem(_kat_input(case), _kat_state(case), case["name"], dtype, fused=False)
            _close(got, torch.from_numpy(arrays[case["name"] + "_out"]), dtype, case["name"], scale=3.0)
    lay = [("s.conv.weight", (64, 3, 5, 7, 7), "float32"), ("s.bn.weight", (64,), "float32"), ("s.bn.bias", (64,), "float32"),
           ("s.bn.running_mean", (64,), "float32"), ("s.bn.running_var", (64,), "float32")]
    sd = synth.fill_layout(lay, 91)
    # conv out 17x19 and 9x35; then a frame count far below the CU count with a tall image: the fused stem cuts every
    # frame into 4 bands of pooled rows (5, 5, 5, 3 of 18), each started with an unstored row pair for the pool's upper row
    for shape in ((1, 3, 3, 33, 37), (2, 3, 2, 18, 70), (1, 3, 4, 70, 40)):
        x = synth.synthetic_tensor(shape, 92 + shape[3])
        want = oracle.stem(x, sd, "s")
        _close(hip_stem(x, sd, "s", dtype, fused=True), want, dtype, "stem %s" % (shape,), scale=3.0)
        _close(hip_stem(x, sd, "s", dtype, fused="rgb3"), want, dtype, "stem rgb3 %s" % (shape,), scale=3.0)
    # the K-packed stem with kt = 3 and kt = 1 (11 / 16 and 21 / 24 fragments in the last K-block), odd width, and its uint8 prologue
    x = synth.synthetic_tensor((1, 3, 2, 18, 230), 99)          # 8 column tiles: no free wave, pooling by the whole workgroup
    _close(hip_stem(x, sd, "s", dtype, fused="rgb3"), oracle.stem(x, sd, "s"), dtype, "stem rgb3 wide", scale=3.0)
    for kt, shape in ((3, (1, 3, 3, 20, 37)), (1, (1, 3, 2, 33, 18))):
        lay_k = [("s.conv.weight", (64, 3, kt, 7, 7), "float32")] + lay[1:]
        sdk = synth.fill_layout(lay_k, 93 + kt)
        x = synth.synthetic_tensor(shape, 95 + kt)
        _close(hip_stem(x, sdk, "s", dtype, fused="rgb3"), oracle.stem(x, sdk, "s"), dtype, "stem rgb3 kt=%d" % kt, scale=3.0)
    u8 = synth.synthetic_clips_u8(1, seed=7, kind="smooth", num_frames=3, size=40).cuda()
    mean, std = synth.pixel_mean_std_f32()
    got = hh.to_ncdhw(hh.stem3_conv_pool(None, sd["s.conv.weight"], *hh.fold_bn(sd, "s.bn"), dtype, u8=u8, mean=mean.tolist(), std=std.tolist()))
    _close(got, oracle.stem(synth.normalize_like_callers(u8.cpu()), sd, "s"), dtype, "stem rgb3 u8", scale=3.0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cout,kt,shape", [(8, 5, (2, 6, 50, 70)),      # Fast pathway: conv out 25x35 (ragged 8-row blocks, ragged column tile)
                                           (8, 5, (1, 3, 32, 32)),      # exactly two row blocks, one column tile
                                           (16, 1, (1, 2, 18, 40)),     # kt = 1 (Slow-pathway style), 16 channels
                                           (4, 3, (1, 4, 14, 14))])     # fewer rows than a block
def test_narrow_stem_vs_oracle(dtype, cout, kt, shape):
    """[kt,7,7] stride [1,2,2] stems with <= 16 output channels (SlowFast's 8-channel Fast-pathway stem): the 16-bit path is
    the row-sharing kernel (a wave owns 16 columns x 8 output rows), fp32 the generic stem kernel."""
    n, t, h, w = shape
    seed = 900 + cout + kt
    lay = [("conv.weight", (cout, 3, kt, 7, 7), "float32"), ("bn.weight", (cout,), "float32"), ("bn.bias", (cout,), "float32"),
           ("bn.running_mean", (cout,), "float32"), ("bn.running_var", (cout,), "float32")]
    sd = synth.fill_layout(lay, seed)
    x = synth.synthetic_tensor((n, 3, t, h, w), seed)
    if dtype != "f32":
        x = x.to(hh.TORCH_DT[dtype]).float()
        sd["conv.weight"] = sd["conv.weight"].to(hh.TORCH_DT[dtype]).float()
    want = oracle.conv_bn_act(x.double(), sd["conv.weight"].double(), {k: v.double() for k, v in sd.items()}, "bn", (1, 2, 2),
                              (kt // 2, 3, 3), True)
    sin = hh.pack_input_f32(x.cuda(), dtype)
    got = hh.to_ncdhw(hh.stem_conv(sin, (n, t, h, w), sd["conv.weight"], *hh.fold_bn(sd, "bn"), dtype)).double()
    assert got.shape == want.shape
    tol = {"f32": 5e-6, "f16": 1.5e-3, "bf16": 1.2e-2}[dtype]
    assert (got - want).abs().max().item() <= tol * (want.abs().max().item() + 1e-9)


@pytest.mark.parametrize("dtype", DTYPES)
def test_projection_block_unfused_path(golden_f3, dtype):
    """the same blocks with the shortcut as its own launch + residual add (generic path of af_conv3d_bn_act)"""
    cases, arrays = golden_f3
    for case in cases:
        if case["kind"] == "block" and "proj" in case["name"]:
            sd, x = _kat_state(case), _kat_input(case)
            got = hip_block(x, sd, case["name"], case["stride"], dtype, fuse_shortcut=False)
            _close(got, torch.from_numpy(arrays[case["name"] + "_out"]), dtype, case["name"], scale=3.0)


def test_fold_bn_matches_torch():
    sd = synth.fill_layout([("bn.weight", (256,), "float32"), ("bn.bias", (256,), "float32"),
                            ("bn.running_mean", (256,), "float32"), ("bn.running_var", (256,), "float32")], 77)
    scale, shift = hh.fold_bn(sd, "bn")
    x = synth.synthetic_tensor((3, 256, 2, 2, 2), 78)
    want = F.batch_norm(x, sd["bn.running_mean"], sd["bn.running_var"], sd["bn.weight"], sd["bn.bias"], False, 0.1, 1e-5)
    got = x * scale.cpu()[:256].view(1, -1, 1, 1, 1) + shift.cpu()[:256].view(1, -1, 1, 1, 1)
    assert (got - want).abs().max().item() <= 2e-6


CONV_CASES = [
    # name, cin, cout, kernel, stride, pad, in dims (n,t,h,w), relu, residual
    ("1x1x1", 64, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 3, 9, 7), False, False),
    ("1x1x1_res_relu", 256, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 2, 11, 13), True, True),
    ("1x1x1_s2", 256, 512, (1, 1, 1), (1, 2, 2), (0, 0, 0), (2, 2, 9, 11), False, False),
    ("3x1x1", 256, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 5, 6, 7), True, False),
    ("1x3x3", 64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (2, 2, 13, 12), True, False),
    ("1x3x3_s2", 128, 128, (1, 3, 3), (1, 2, 2), (0, 1, 1), (1, 3, 14, 15), True, False),
    # the weights-in-registers 64 -> 64 kernel (s2 `b` convs) at its own frame size and on a ragged 30 x 27 frame (7.5 strips
    # of 4 rows); narrower frames take the generic kernel (case "1x3x3" above)
    ("c64_1x3x3_56x56", 64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 5, 56, 56), True, False),
    ("c64_1x3x3_30x27", 64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (2, 3, 30, 27), True, False),
    # ... and with several strips per workgroup (round 4's skewed form: a three-slot patch ring that wraps, one barrier per strip,
    # the two channel halves half a strip apart): 7 strips per workgroup at 56 x 56, 4-5 at a ragged 54 x 50
    ("c64_1x3x3_56x56_many", 64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (4, 32, 56, 56), True, False),
    ("c64_1x3x3_54x50_many", 64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (3, 27, 54, 50), True, False),
    ("3x3x3_synthetic", 64, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 4, 9, 10), True, False),
    ("5x1x1_t_stride8", 64, 128, (5, 1, 1), (8, 1, 1), (2, 0, 0), (1, 32, 4, 5), True, False),
    ("big_m_tail", 64, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0), (3, 7, 17, 19), False, False),
    # a handful of rows and a long K (one clip in the deep stages): 8-way split-K, fp32 partial sums + finish kernel
    ("splitk_1x3x3_res", 512, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 2, 7, 7), True, True),
    ("splitk_3x1x1_wide", 1024, 512, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 4, 5, 5), False, False),
    # SlowFast channel counts: not multiples of the 64-wide K-step / channel tile (padding + masks)
    ("sf_slow_s2a_80", 80, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 2, 9, 9), True, False),
    ("sf_fast_8to8_3x1x1", 8, 8, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 6, 7, 7), True, False),
    ("sf_fast_8to32_res", 8, 32, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 3, 6, 5), True, True),
    ("sf_fast_16_1x3x3_s2", 16, 16, (1, 3, 3), (1, 2, 2), (0, 1, 1), (1, 4, 11, 12), True, False),
    ("sf_slow_320to128", 320, 128, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 2, 7, 8), True, False),
    ("sf_fuse_32to64_t8", 32, 64, (5, 1, 1), (8, 1, 1), (2, 0, 0), (1, 32, 3, 4), True, False),
    # large enough for the 256x256 tile (4 waves, accumulators in AGPRs, 2-slot ring): full and ragged last tile
    ("tile256_1x3x3_res", 64, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 16, 64, 64), True, True),
    ("tile256_ragged_m", 64, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 16, 56, 55), False, False),
    # ... and its 224-row form (64 x 112 per wave; positions that are multiples of 49: the deep stages' 14x14 / 7x7 frames):
    # 224 whole tiles, a ragged last tile (171.5 tiles), and the 3x1x1 / 1x1x1 long-K shapes of s4 at a reduced size
    ("tile224_1x3x3_res", 64, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 16, 56, 56), True, True),
    ("tile224_ragged_m", 64, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 16, 49, 49), False, False),
    ("tile224_3x1x1_1024to256", 1024, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0), (11, 16, 14, 14), True, False),
    ("tile224_1x1x1_1024to256_res", 1024, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), (11, 16, 14, 14), True, True),
    # ... and for the 128x512 tile of 128-channel layers (its 2-slot ring is the whole 160 KB of LDS)
    ("tile512_1x3x3_res", 128, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 8, 96, 96), True, True),
    ("tile512_ragged_m", 128, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 8, 96, 95), False, False),
    # time-tiled 3x1x1 -> 64 kernel (T = 32: 8 positions per tile, T = 16: 16), ragged last spatial chunk, several tiles
    # per workgroup only at bench sizes - here every workgroup gets one or two
    ("t311_64to64_T32", 64, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), (2, 32, 7, 9), True, False),
    ("t311_256to64_T16", 256, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 16, 10, 10), True, False),
    ("t311_256to64_T32_many", 256, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0), (3, 32, 28, 28), False, False),
    ("t311_256to128_T16", 256, 128, (3, 1, 1), (1, 1, 1), (1, 0, 0), (2, 16, 9, 9), True, False),   # 128-row tiles; fp32 -> generic
    # persistent 1x1x1 stream (weights in registers, residual prefetched a tile ahead): needs >= 4 tiles per CU;
    # full tiles, a ragged last tile, two channel columns (512 outputs), K = 64 and 128
    ("stream111_64to256_res", 64, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 8, 96, 96), True, True),
    ("stream111_64to256_ragged", 64, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 9, 121, 121), True, False),
    ("stream111_128to512_res_ragged", 128, 512, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 6, 107, 109), False, True),
    ("stream111_256to1024_res_ragged", 256, 1024, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 4, 91, 93), True, True),   # 32-channel wave columns
    ("stream111_64to768_three_columns", 64, 768, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 5, 95, 97), True, True),    # grid not a multiple of 8 x columns
    # K = 128 / 256 run 64-position tiles through a 4-slot ring (three tiles ahead, counted waits): without a residual the
    # per-tile operation counts differ; ragged last tiles
    ("stream111_128to256_nores_ragged", 128, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 13, 101, 103), True, False),
    ("stream111_256to512_nores_ragged", 256, 512, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 7, 99, 101), False, False),
    # frame / band resident 1x3x3 (the s3 / s4 `b` convs at bench size: >= 192 work units): whole 14x14 frames, 256 channels,
    # 1 and 3 K slabs (the third slab's patch is issued inside the loop); a 13-row frame; s3's 28x28 in two bands of 14
    # rows; 27x26 (bands of 14 + 13 rows, narrower pitch); no ReLU
    ("halo133_64to256_14x14", 64, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1), (2, 100, 14, 14), True, False),
    ("halo133_192to256_13x14", 192, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 192, 13, 14), True, False),
    ("halo133_128to128_28x28", 128, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1), (2, 48, 28, 28), True, False),
    ("halo133_256to128_27x26_norelu", 256, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 100, 27, 26), False, False),
    # the same kernel with kT = 3 (SYNTHETIC 3x3x3, BASELINE.json's literal metric; no such layer in the reference model):
    # patches from frames t-1, t, t+1 (zeros outside the clip: 2 clips, so clip boundaries are inside the batch), 64 / 128 /
    # 256 output channels, 1 and 2 channel slabs per frame
    ("halo333_64to64_56x56", 64, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), (2, 14, 56, 56), True, False),
    ("halo333_64to64_50x53_norelu", 64, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 36, 50, 53), False, False),
    ("halo333_64to128_28x28", 64, 128, (3, 3, 3), (1, 1, 1), (1, 1, 1), (2, 48, 28, 28), True, False),
    ("halo333_128to256_14x14", 128, 256, (3, 3, 3), (1, 1, 1), (1, 1, 1), (2, 100, 14, 14), True, False),
    # round 4: the same patch-resident kernel in its TEMPORAL mode (3x1x1 into 128 / 256 channels, >= 192 units of (clip, P pixels,
    # all T frames)): 256 channels at T = 16 (P = 14, three ring slots), a ragged last chunk, 128 channels at T = 16 (P = 28) and
    # at T = 32 (P = 14), no ReLU
    ("t311g_128to256_T16", 128, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0), (2, 16, 28, 49), True, False),
    ("t311g_192to256_T16_ragged", 192, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 16, 53, 51), True, False),
    ("t311g_64to128_T16_ragged", 64, 128, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 16, 74, 73), True, False),
    ("t311g_128to128_T32_norelu", 128, 128, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 32, 52, 52), False, False),
]
EXPECT_VARIANT = {"c64_1x3x3_56x56": {"f16": 4, "bf16": 4}, "c64_1x3x3_30x27": {"f16": 4, "bf16": 4},
                  "c64_1x3x3_56x56_many": {"f16": 4, "bf16": 4}, "c64_1x3x3_54x50_many": {"f16": 4, "bf16": 4}, "1x3x3": {"f16": 3, "bf16": 3},
                  "tile256_1x3x3_res": 6, "tile256_ragged_m": 6, "tile224_1x3x3_res": 12, "tile224_ragged_m": 12,
                  "tile224_3x1x1_1024to256": {"f16": 12, "bf16": 12}, "tile224_1x1x1_1024to256_res": {"f16": 12, "bf16": 12}, "tile512_1x3x3_res": 7, "tile512_ragged_m": 7,
                  "t311_64to64_T32": 8, "t311_256to64_T16": 8, "t311_256to64_T32_many": 8,
                  "t311_256to128_T16": {"f32": 3, "f16": 8, "bf16": 8},
                  "stream111_64to256_res": {"f32": 2, "f16": 10, "bf16": 10}, "stream111_64to256_ragged": {"f32": 2, "f16": 10, "bf16": 10},
                  "stream111_128to512_res_ragged": {"f32": 5, "f16": 10, "bf16": 10},
                  "stream111_256to1024_res_ragged": {"f32": 5, "f16": 10, "bf16": 10},
                  "stream111_64to768_three_columns": {"f32": 2, "f16": 10, "bf16": 10},
                  "stream111_128to256_nores_ragged": {"f16": 10, "bf16": 10}, "stream111_256to512_nores_ragged": {"f16": 10, "bf16": 10},
                  "halo133_64to256_14x14": {"f32": 12, "f16": 11, "bf16": 11}, "halo133_192to256_13x14": {"f32": None, "f16": 11, "bf16": 11},
                  "halo133_128to128_28x28": {"f32": 7, "f16": 11, "bf16": 11},
                  "halo133_256to128_27x26_norelu": {"f32": 7, "f16": 11, "bf16": 11},
                  "halo333_64to64_56x56": {"f16": 11, "bf16": 11}, "halo333_64to64_50x53_norelu": {"f16": 11, "bf16": 11},
                  "halo333_64to128_28x28": {"f16": 11, "bf16": 11}, "halo333_128to256_14x14": {"f16": 11, "bf16": 11},
                  "t311g_128to256_T16": {"f16": 13, "bf16": 13}, "t311g_192to256_T16_ragged": {"f16": 13, "bf16": 13},
                  "t311g_64to128_T16_ragged": {"f16": 13, "bf16": 13}, "t311g_128to128_T32_norelu": {"f16": 13, "bf16": 13}}


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_vs_oracle(case, dtype):
    name, cin, cout, k, s, p, dims, relu, use_res = case
    seed = 1000 + sum(map(ord, name))
    lay = [("w.weight", (cout, cin) + k, "float32"), ("bn.weight", (cout,), "float32"), ("bn.bias", (cout,), "float32"),
           ("bn.running_mean", (cout,), "float32"), ("bn.running_var", (cout,), "float32")]
    sd = synth.fill_layout(lay, seed)
    x = synth.synthetic_tensor((dims[0], cin) + dims[1:], seed)
    if dtype != "f32":                       # same rounded operands for both sides: isolates the kernel
        x = x.to(hh.TORCH_DT[dtype]).float()
        sd["w.weight"] = sd["w.weight"].to(hh.TORCH_DT[dtype]).float()
    want = oracle.conv_bn_act(x.double(), sd["w.weight"].double(), {kk: v.double() for kk, v in sd.items()}, "bn", s, p, False)
    res = None
    if use_res:
        res = synth.synthetic_tensor(tuple(want.shape), seed + 1)
        if dtype != "f32":
            res = res.to(hh.TORCH_DT[dtype]).float()
        want = want + res.double()
    if relu:
        want = F.relu(want)
    got = hh.conv_bn_act(hh.to_ndhwc(x, dtype), sd["w.weight"], *hh.fold_bn(sd, "bn"), s, p, relu, dtype,
                         residual=None if res is None else hh.to_ndhwc(res, dtype))
    if name in EXPECT_VARIANT:
        want_variant = EXPECT_VARIANT[name].get(dtype) if isinstance(EXPECT_VARIANT[name], dict) else EXPECT_VARIANT[name]
        assert want_variant is None or hh.conv_bn_act.last_variant == want_variant, hh.conv_bn_act.last_variant
    tol = {"f32": 2e-6, "f16": 1.5e-3, "bf16": 1.2e-2}[dtype]     # operands pre-rounded: only output rounding + fp32 accumulation remain
    if dtype == "f32":
        tol *= max(1.0, cin * k[0] * k[1] * k[2] / 1024.0)        # fp32 accumulation error grows with K (3072 here at most)
    got = hh.to_ncdhw(got).double()
    err = (got - want).abs().max().item()
    assert err <= tol * (want.abs().max().item() + 1e-9), "%s[%s] err %.3e" % (name, dtype, err)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cout,dims", [(64, 256, (2, 6, 9, 7)), (64, 64, (1, 4, 5, 5)),
                                           (64, 256, (2, 16, 63, 65))])     # persistent stream: 1024 tiles, ragged 64-pixel chunks
def test_conv_with_fused_temporal_maxpool(dtype, cin, cout, dims):
    """s2's last 1x1x1 (+ residual + ReLU) with pathway0_pool = MaxPool3d([2,1,1]) fused into its epilogue."""
    seed = 4242 + cout
    lay = [("w.weight", (cout, cin, 1, 1, 1), "float32"), ("bn.weight", (cout,), "float32"), ("bn.bias", (cout,), "float32"),
           ("bn.running_mean", (cout,), "float32"), ("bn.running_var", (cout,), "float32")]
    sd = synth.fill_layout(lay, seed)
    x = synth.synthetic_tensor((dims[0], cin) + dims[1:], seed)
    res = synth.synthetic_tensor((dims[0], cout) + dims[1:], seed + 1)
    if dtype != "f32":
        x, res = x.to(hh.TORCH_DT[dtype]).float(), res.to(hh.TORCH_DT[dtype]).float()
        sd["w.weight"] = sd["w.weight"].to(hh.TORCH_DT[dtype]).float()
    y = oracle.conv_bn_act(x.double(), sd["w.weight"].double(), {k: v.double() for k, v in sd.items()}, "bn", (1, 1, 1), (0, 0, 0), False)
    want = F.max_pool3d(F.relu(y + res.double()), (2, 1, 1), (2, 1, 1))
    got = hh.conv_bn_act(hh.to_ndhwc(x, dtype), sd["w.weight"], *hh.fold_bn(sd, "bn"), (1, 1, 1), (0, 0, 0), True, dtype,
                         residual=hh.to_ndhwc(res, dtype), tpool=True)
    if dims[2] * dims[3] > 4000 and dtype != "f32":
        assert hh.conv_bn_act.last_variant == 10, hh.conv_bn_act.last_variant
    got = hh.to_ncdhw(got).double()
    assert got.shape == want.shape
    tol = {"f32": 2e-6, "f16": 1.5e-3, "bf16": 1.2e-2}[dtype]
    assert (got - want).abs().max().item() <= tol * want.abs().max().item()


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_projection_shortcut_stream_at_size(dtype):
    """relu(bn(c(x)) + bn1(branch1(x2))) of s2's first block at a size that takes the persistent 1x1x1 stream (two 64-channel
    K slabs from two inputs, no residual), ragged last tile; against the oracle on pre-rounded operands."""
    dims, cin, cout, seed = (1, 10, 115, 115), 64, 256, 909          # 132250 positions = 1033 tiles + 26 rows
    lay = []
    for nm in ("c", "b1"):
        lay += [(nm + ".weight", (cout, cin, 1, 1, 1), "float32"), (nm + "_bn.weight", (cout,), "float32"), (nm + "_bn.bias", (cout,), "float32"),
                (nm + "_bn.running_mean", (cout,), "float32"), (nm + "_bn.running_var", (cout,), "float32")]
    sd = synth.fill_layout(lay, seed)
    x = synth.synthetic_tensor((dims[0], cin) + dims[1:], seed).to(hh.TORCH_DT[dtype]).float()
    x2 = synth.synthetic_tensor((dims[0], cin) + dims[1:], seed + 1).to(hh.TORCH_DT[dtype]).float()
    sdd = {k: v.double() for k, v in sd.items()}
    want = F.relu(oracle.conv_bn_act(x.double(), sdd["c.weight"], sdd, "c_bn", (1, 1, 1), (0, 0, 0), False) +
                  oracle.conv_bn_act(x2.double(), sdd["b1.weight"], sdd, "b1_bn", (1, 1, 1), (0, 0, 0), False))
    got = hh.conv_dual(hh.to_ndhwc(x, dtype), sd["c.weight"], hh.fold_bn(sd, "c_bn"), hh.to_ndhwc(x2, dtype), sd["b1.weight"],
                       hh.fold_bn(sd, "b1_bn"), (1, 1, 1), dtype)
    assert hh.conv_dual.last_variant == 10, hh.conv_dual.last_variant
    got = hh.to_ncdhw(got).double()
    # the BN scales are folded into the packed weights here (one more rounding of each weight than the single-input cases)
    tol = {"f16": 3e-3, "bf16": 2.4e-2}[dtype]
    assert (got - want).abs().max().item() <= tol * want.abs().max().item()


@pytest.mark.parametrize("rows", [256, 224])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_projection_shortcut_256_tile(dtype, rows, monkeypatch):
    """s4's first block: c (256 -> 1024) + strided shortcut (512 -> 1024, stride (1,2,2)) as one launch on the 256x256 tile
    (12 K-steps over two inputs, second one gathered at stride 2) and on its 224-row form (what 14x14 frames get)."""
    monkeypatch.setenv("AF_IGEMM_224", "0" if rows == 256 else "1")
    n, t, hw, cin, cin2, cout, seed = 4, 16, 14, 256, 512, 1024, 1212
    lay = []
    for nm, ci in (("c", cin), ("b1", cin2)):
        lay += [(nm + ".weight", (cout, ci, 1, 1, 1), "float32"), (nm + "_bn.weight", (cout,), "float32"), (nm + "_bn.bias", (cout,), "float32"),
                (nm + "_bn.running_mean", (cout,), "float32"), (nm + "_bn.running_var", (cout,), "float32")]
    sd = synth.fill_layout(lay, seed)
    x = synth.synthetic_tensor((n, cin, t, hw, hw), seed)
    x2 = synth.synthetic_tensor((n, cin2, t, 2 * hw, 2 * hw), seed + 1)
    if dtype != "f32":
        x, x2 = x.to(hh.TORCH_DT[dtype]).float(), x2.to(hh.TORCH_DT[dtype]).float()
    sdd = {k: v.double() for k, v in sd.items()}
    want = F.relu(oracle.conv_bn_act(x.double(), sdd["c.weight"], sdd, "c_bn", (1, 1, 1), (0, 0, 0), False) +
                  oracle.conv_bn_act(x2.double(), sdd["b1.weight"], sdd, "b1_bn", (1, 2, 2), (0, 0, 0), False))
    got = hh.conv_dual(hh.to_ndhwc(x, dtype), sd["c.weight"], hh.fold_bn(sd, "c_bn"), hh.to_ndhwc(x2, dtype), sd["b1.weight"],
                       hh.fold_bn(sd, "b1_bn"), (1, 2, 2), dtype)
    assert hh.conv_dual.last_variant == (6 if rows == 256 else 12), hh.conv_dual.last_variant
    got = hh.to_ncdhw(got).double()
    tol = {"f32": 5e-6, "bf16": 2.4e-2}[dtype]          # bf16: BN scales folded into the packed weights (one more rounding)
    assert (got - want).abs().max().item() <= tol * want.abs().max().item()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cout,dims,relu", [(128, 128, (2, 3, 8, 10), True),       # pooled `b` conv (64x128 tile: 4 pooled rows per patch)
                                                (256, 512, (1, 4, 12, 6), False),      # pooled projection shortcut
                                                (64, 256, (1, 16, 56, 56), True)])     # 256x256 tile
def test_conv_with_fused_2x2_maxpool(dtype, cin, cout, dims, relu):
    """FTCN-TT: conv -> BN -> MaxPool3d((1,2,2)) [-> ReLU] (i3d_temporal_var_fix_dropout_tt_cfg.py:207-288) in one launch."""
    seed = 777 + cout
    lay = [("w.weight", (cout, cin, 1, 1, 1), "float32"), ("bn.weight", (cout,), "float32"), ("bn.bias", (cout,), "float32"),
           ("bn.running_mean", (cout,), "float32"), ("bn.running_var", (cout,), "float32")]
    sd = synth.fill_layout(lay, seed)
    x = synth.synthetic_tensor((dims[0], cin) + dims[1:], seed)
    if dtype != "f32":
        x = x.to(hh.TORCH_DT[dtype]).float()
        sd["w.weight"] = sd["w.weight"].to(hh.TORCH_DT[dtype]).float()
    y = oracle.conv_bn_act(x.double(), sd["w.weight"].double(), {k: v.double() for k, v in sd.items()}, "bn", (1, 1, 1), (0, 0, 0), False)
    want = F.max_pool3d(y, (1, 2, 2))
    if relu:
        want = F.relu(want)
    got = hh.conv_bn_act(hh.to_ndhwc(x, dtype), sd["w.weight"], *hh.fold_bn(sd, "bn"), (1, 1, 1), (0, 0, 0), relu, dtype, tpool=2)
    got = hh.to_ncdhw(got).double()
    assert got.shape == want.shape
    tol = {"f32": 2e-6, "f16": 1.5e-3, "bf16": 1.2e-2}[dtype]
    assert (got - want).abs().max().item() <= tol * (want.abs().max().item() + 1e-9)


def test_conv_out_ld_writes_into_concat_buffer():
    """FuseFastToSlow concatenates by channel: the conv writes at a channel offset of a wider tensor."""
    dtype = "f32"
    lay = [("w.weight", (64, 64, 1, 1, 1), "float32"), ("bn.weight", (64,), "float32"), ("bn.bias", (64,), "float32"),
           ("bn.running_mean", (64,), "float32"), ("bn.running_var", (64,), "float32")]
    sd = synth.fill_layout(lay, 5)
    x = synth.synthetic_tensor((1, 64, 2, 5, 5), 6)
    wide = torch.full((1, 2, 5, 5, 192), 7.0, device="cuda")
    view = wide.view(-1, 192)[:, 128:]
    assert view.data_ptr() % 16 == 0
    hh.conv_bn_act(hh.to_ndhwc(x, dtype), sd["w.weight"], *hh.fold_bn(sd, "bn"), (1, 1, 1), (0, 0, 0), True, dtype,
                   out=view, out_ld=192)
    want = oracle.conv_bn_act(x, sd["w.weight"], sd, "bn", (1, 1, 1), (0, 0, 0), True).permute(0, 2, 3, 4, 1)
    assert torch.all(wide[..., :128] == 7.0)
    assert (wide[..., 128:].cpu() - want).abs().max().item() <= 1e-5


def test_stream_kernel_out_ld_and_temporal_pool_into_concat_buffer():
    """the persistent 1x1x1 stream (plain and with the fused frame-pair max) writing at a channel offset of a wider tensor:
    only its 256 columns change, row pitch = out_ld."""
    dtype = "bf16"
    lay = [("w.weight", (256, 64, 1, 1, 1), "float32"), ("bn.weight", (256,), "float32"), ("bn.bias", (256,), "float32"),
           ("bn.running_mean", (256,), "float32"), ("bn.running_var", (256,), "float32")]
    sd = synth.fill_layout(lay, 15)
    sd["w.weight"] = sd["w.weight"].to(torch.bfloat16).float()
    dims = (2, 8, 96, 96)
    x = synth.synthetic_tensor((dims[0], 64) + dims[1:], 16).to(torch.bfloat16).float()
    y = oracle.conv_bn_act(x.double(), sd["w.weight"].double(), {k: v.double() for k, v in sd.items()}, "bn", (1, 1, 1), (0, 0, 0), True)
    for tpool in (False, True):
        t_out = dims[1] // 2 if tpool else dims[1]
        want = (F.max_pool3d(y, (2, 1, 1), (2, 1, 1)) if tpool else y).permute(0, 2, 3, 4, 1)
        wide = torch.full((dims[0], t_out, 96, 96, 320), 7.0, dtype=torch.bfloat16, device="cuda")
        view = wide.view(-1, 320)[:, 32:288]
        assert view.data_ptr() % 16 == 0
        hh.conv_bn_act(hh.to_ndhwc(x, dtype), sd["w.weight"], *hh.fold_bn(sd, "bn"), (1, 1, 1), (0, 0, 0), True, dtype,
                       out=view, out_ld=320, tpool=tpool)
        assert hh.conv_bn_act.last_variant == 10
        assert torch.all(wide[..., :32] == 7.0) and torch.all(wide[..., 288:] == 7.0)
        got = wide[..., 32:288].float().cpu().double()
        assert (got - want).abs().max().item() <= 1.2e-2 * want.abs().max().item()


def test_input_prologue_u8_matches_callers():
    u8 = synth.synthetic_clips_u8(2, seed=3, kind="uniform", num_frames=3, size=10)
    x = synth.normalize_like_callers(u8)                       # (B,3,T,H,W) view, channels-last strides
    mean, std = synth.pixel_mean_std_f32()
    a = hh.pack_input_u8(u8.cuda(), mean.tolist(), std.tolist(), "f32")
    b = hh.pack_input_f32(x.cuda(), "f32")
    c = hh.pack_input_f32(x.contiguous().cuda(), "f32")        # NCDHW-contiguous source (feature.py:123)
    assert torch.equal(a, b) and torch.equal(b, c)
    vol = b.view(2, 3 + 4, 10 + 6, 10 + 8, 4)
    assert torch.equal(vol[:, 2:5, 3:13, 3:13, :3].cpu(), x.permute(0, 2, 3, 4, 1))
    assert vol[..., 3].abs().max().item() == 0 and vol[:, :2].abs().max().item() == 0


def test_argument_errors_are_reported():
    import ctypes as C
    L = hh.lib()
    d = L.ConvDesc()
    rc = L.lib.af_conv3d_bn_act(C.byref(d), None, None, None, None, None, None, 0, None, 0, None)
    assert rc == -1 and b"null" in L.lib.af_last_error()
    x = torch.zeros((1, 1, 4, 4, 42), device="cuda")
    with pytest.raises(L.AfError, match="multiple of"):
        hh.conv_bn_act(x, torch.zeros(64, 42, 1, 1, 1), torch.ones(64, device="cuda"), torch.zeros(64, device="cuda"),
                       (1, 1, 1), (0, 0, 0), False, "f32")


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("name,cin,cmid,cout,dims,use_res", [
    ("s4_like", 256, 256, 1024, (1, 192, 14, 14), True),          # whole 14x14 frames, 4 K slabs, 4 output passes
    ("s3_like", 128, 128, 512, (2, 48, 28, 28), True),            # two bands of 14 rows per frame
    ("ragged_no_res", 64, 128, 256, (1, 100, 27, 26), False),     # bands of 14 + 13 rows, no residual
])
def test_fused_b_c_vs_oracle(name, cin, cmid, cout, dims, use_res, dtype):
    """af_conv3d_bc_bn_act: relu(bn_c(c(relu(bn_b(b(x))))) + residual) in one launch (the b output stays in LDS) against the
    oracle's two conv_bn_act calls in fp64; the intermediate is rounded to the storage type on both sides."""
    seed = 4000 + sum(map(ord, name))
    lay = [("b.weight", (cmid, cin, 1, 3, 3), "float32"), ("c.weight", (cout, cmid, 1, 1, 1), "float32")]
    for p_, ch in (("b_bn", cmid), ("c_bn", cout)):
        lay += [(p_ + s_, (ch,), "float32") for s_ in (".weight", ".bias", ".running_mean", ".running_var")]
    sd = synth.fill_layout(lay, seed)
    tdt = hh.TORCH_DT[dtype]
    x = synth.synthetic_tensor((dims[0], cin) + dims[1:], seed).to(tdt).float()
    for k in ("b.weight", "c.weight"):
        sd[k] = sd[k].to(tdt).float()
    sd64 = {k: v.double() for k, v in sd.items()}
    mid = oracle.conv_bn_act(x.double(), sd64["b.weight"], sd64, "b_bn", (1, 1, 1), (0, 1, 1), True)
    mid = mid.to(tdt).double()                                  # the one rounding of the b output (LDS tile in the storage type)
    want = oracle.conv_bn_act(mid, sd64["c.weight"], sd64, "c_bn", (1, 1, 1), (0, 0, 0), False)
    res = None
    if use_res:
        res = synth.synthetic_tensor(tuple(want.shape), seed + 1).to(tdt).float()
        want = want + res.double()
    want = F.relu(want)
    got = hh.conv_bc(hh.to_ndhwc(x, dtype), sd["b.weight"], hh.fold_bn(sd, "b_bn"), sd["c.weight"], hh.fold_bn(sd, "c_bn"),
                     None if res is None else hh.to_ndhwc(res, dtype), dtype)
    assert got is not None, "the library should fuse this pair"
    got = hh.to_ncdhw(got).double()
    tol = {"f16": 2e-3, "bf16": 1.6e-2}[dtype]                   # a b value on a rounding boundary may round the other way
    err = (got - want).abs().max().item()
    assert err <= tol * (want.abs().max().item() + 1e-9), "%s[%s] err %.3e" % (name, dtype, err)


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("name,inner,kta,dims,out_ld", [
    ("fast_s2_like", 8, 3, (2, 12, 56, 56), 0),          # 14 x 14 patches, 16 per frame, 2 time segments of 6 frames... per clip
    ("fast_s2_ragged", 8, 3, (1, 9, 37, 45), 0),         # ragged patch rows / columns (37 = 2 x 14 + 9, 45 = 3 x 14 + 3), odd T
    ("fast_s3_like", 16, 3, (2, 16, 28, 28), 0),         # inner 16: 7-row patches, two K-blocks per tap in a
    ("fast_s3_kt1_wide_rows", 16, 1, (1, 8, 20, 30), 80),   # a without temporal taps; output rows wider than the trunk (concat buffer)
    ("one_frame", 8, 3, (1, 1, 14, 14), 0),              # T = 1: both temporal taps fall outside the clip
    ("fast_s4_like", 32, 3, (2, 16, 14, 14), 0),         # inner 32: two channel tiles in a / b, 3-row patches, 4 K-blocks per tap
    ("fast_s4_ragged_kt1", 32, 1, (1, 5, 10, 17), 0),
])
def test_fused_block_a_b_c_vs_oracle(name, inner, kta, dims, out_ld, dtype, monkeypatch):
    """af_block_abc_bn_act: a whole identity-shortcut bottleneck of SlowFast's Fast pathway (kT x 1 x 1 -> 1x3x3 -> 1x1x1, + x,
    ReLU) in one launch against three oracle conv_bn_act calls in fp64; a and b are rounded to the storage type on both sides
    (they cross LDS in it).  Columns beyond the block's channels of a wider output row must stay untouched."""
    if inner == 32:
        monkeypatch.setenv("AF_ABC_INNER32", "1")                # correct but not faster than three launches: off by default
    seed = 6000 + sum(map(ord, name))
    C4 = 4 * inner
    lay = [("a.weight", (inner, C4, kta, 1, 1), "float32"), ("b.weight", (inner, inner, 1, 3, 3), "float32"),
           ("c.weight", (C4, inner, 1, 1, 1), "float32")]
    for p_, ch in (("a_bn", inner), ("b_bn", inner), ("c_bn", C4)):
        lay += [(p_ + s_, (ch,), "float32") for s_ in (".weight", ".bias", ".running_mean", ".running_var")]
    sd = synth.fill_layout(lay, seed)
    tdt = hh.TORCH_DT[dtype]
    x = synth.synthetic_tensor((dims[0], C4) + dims[1:], seed).to(tdt).float()
    for k in ("a.weight", "b.weight", "c.weight"):
        sd[k] = sd[k].to(tdt).float()
    sd64 = {k: v.double() for k, v in sd.items()}
    ya = oracle.conv_bn_act(x.double(), sd64["a.weight"], sd64, "a_bn", (1, 1, 1), (kta // 2, 0, 0), True).to(tdt).double()
    yb = oracle.conv_bn_act(ya, sd64["b.weight"], sd64, "b_bn", (1, 1, 1), (0, 1, 1), True).to(tdt).double()
    want = F.relu(oracle.conv_bn_act(yb, sd64["c.weight"], sd64, "c_bn", (1, 1, 1), (0, 0, 0), False) + x.double())
    got = hh.block_abc(hh.to_ndhwc(x, dtype), sd["a.weight"], hh.fold_bn(sd, "a_bn"), sd["b.weight"], hh.fold_bn(sd, "b_bn"),
                       sd["c.weight"], hh.fold_bn(sd, "c_bn"), dtype, out_ld=out_ld)
    assert got is not None, "the library should fuse this block"
    if out_ld:
        assert torch.all(got[..., C4:] == 7.0), "columns beyond the block's channels were written"
        got = got[..., :C4]
    got = hh.to_ncdhw(got.contiguous()).double()
    tol = {"f16": 3e-3, "bf16": 2.4e-2}[dtype]                   # two intermediates on rounding boundaries may round the other way
    err = (got - want).abs().max().item()
    assert err <= tol * (want.abs().max().item() + 1e-9), "%s[%s] err %.3e" % (name, dtype, err)


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("dims", [(2, 12, 56, 56), (1, 7, 33, 30)])
def test_fused_projection_block_vs_oracle(dims, dtype):
    """af_block_abc_bn_act, projection form (block 0 of the Fast pathway's s2): x has 8 channels, a: 3x1x1 8 -> 8 (all three
    taps in one K-block), b: 1x3x3, c: 1x1x1 8 -> 32 and the shortcut 1x1x1 8 -> 32 share one accumulator (BN scales folded into
    both packed weights, shifts summed); against four oracle conv_bn_act calls in fp64."""
    seed = 6500 + dims[2]
    lay = [("a.weight", (8, 8, 3, 1, 1), "float32"), ("b.weight", (8, 8, 1, 3, 3), "float32"), ("c.weight", (32, 8, 1, 1, 1), "float32"),
           ("s.weight", (32, 8, 1, 1, 1), "float32")]
    for p_, ch in (("a_bn", 8), ("b_bn", 8), ("c_bn", 32), ("s_bn", 32)):
        lay += [(p_ + s_, (ch,), "float32") for s_ in (".weight", ".bias", ".running_mean", ".running_var")]
    sd = synth.fill_layout(lay, seed)
    tdt = hh.TORCH_DT[dtype]
    x = synth.synthetic_tensor((dims[0], 8) + dims[1:], seed).to(tdt).float()
    for k in ("a.weight", "b.weight"):
        sd[k] = sd[k].to(tdt).float()
    sd64 = {k: v.double() for k, v in sd.items()}
    ya = oracle.conv_bn_act(x.double(), sd64["a.weight"], sd64, "a_bn", (1, 1, 1), (1, 0, 0), True).to(tdt).double()
    yb = oracle.conv_bn_act(ya, sd64["b.weight"], sd64, "b_bn", (1, 1, 1), (0, 1, 1), True).to(tdt).double()
    want = F.relu(oracle.conv_bn_act(yb, sd64["c.weight"], sd64, "c_bn", (1, 1, 1), (0, 0, 0), False) +
                  oracle.conv_bn_act(x.double(), sd64["s.weight"], sd64, "s_bn", (1, 1, 1), (0, 0, 0), False))
    got = hh.block_abc(hh.to_ndhwc(x, dtype), sd["a.weight"], hh.fold_bn(sd, "a_bn"), sd["b.weight"], hh.fold_bn(sd, "b_bn"),
                       sd["c.weight"], hh.fold_bn(sd, "c_bn"), dtype, w1_oidhw=sd["s.weight"], bn_1=hh.fold_bn(sd, "s_bn"))
    assert got is not None, "the library should fuse this block"
    got = hh.to_ncdhw(got).double()
    tol = {"f16": 4e-3, "bf16": 3e-2}[dtype]       # folded weights: one more rounding of each c / shortcut weight than the oracle's
    err = (got - want).abs().max().item()
    assert err <= tol * (want.abs().max().item() + 1e-9), "proj %s[%s] err %.3e" % (dims, dtype, err)


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("name,ctrunk,dims", [
    ("s2_like_T32", 256, (2, 32, 64, 66)),        # T = 32: 8 pixels per tile, 4 K slabs; 1 056 tiles >= 4 per CU
    ("T16_two_slabs", 128, (3, 16, 76, 76)),      # T = 16: 16 pixels per tile, 2 K slabs, 1 083 tiles
])
def test_fused_c_a_vs_oracle(name, ctrunk, dims, dtype):
    """af_conv3d_ca_bn_act: x = relu(bn_c(c(b)) + res) and a_out = relu(bn_a(a3x1x1(x))) in one launch (the trunk slab is
    produced in LDS and multiplied by the temporal taps there) against the oracle's two conv_bn_act calls in fp64; the trunk
    is rounded to the storage type on both sides before the temporal conv."""
    seed = 5000 + sum(map(ord, name))
    n, t, h, w = dims
    lay = [("c.weight", (ctrunk, 64, 1, 1, 1), "float32"), ("a.weight", (64, ctrunk, 3, 1, 1), "float32")]
    for p_, ch in (("c_bn", ctrunk), ("a_bn", 64)):
        lay += [(p_ + s_, (ch,), "float32") for s_ in (".weight", ".bias", ".running_mean", ".running_var")]
    sd = synth.fill_layout(lay, seed)
    tdt = hh.TORCH_DT[dtype]
    b = synth.synthetic_tensor((n, 64, t, h, w), seed).to(tdt).float()
    res = synth.synthetic_tensor((n, ctrunk, t, h, w), seed + 1).to(tdt).float()
    for k in ("c.weight", "a.weight"):
        sd[k] = sd[k].to(tdt).float()
    sd64 = {k: v.double() for k, v in sd.items()}
    x = F.relu(oracle.conv_bn_act(b.double(), sd64["c.weight"], sd64, "c_bn", (1, 1, 1), (0, 0, 0), False) + res.double())
    xr = x.to(tdt).double()                                     # the one rounding of the trunk
    want_a = oracle.conv_bn_act(xr, sd64["a.weight"], sd64, "a_bn", (1, 1, 1), (1, 0, 0), True)
    out = hh.conv_ca(hh.to_ndhwc(b, dtype), sd["c.weight"], hh.fold_bn(sd, "c_bn"), hh.to_ndhwc(res, dtype), sd["a.weight"],
                     hh.fold_bn(sd, "a_bn"), dtype)
    assert out is not None, "the library should fuse this pair"
    if t == 16:      # a frame whose pixel count is not a multiple of the tile's 16 pixels is not fused (the two launches run instead)
        assert hh.conv_ca(hh.to_ndhwc(b[..., :65].contiguous(), dtype), sd["c.weight"], hh.fold_bn(sd, "c_bn"),
                          hh.to_ndhwc(res[..., :65].contiguous(), dtype), sd["a.weight"], hh.fold_bn(sd, "a_bn"), dtype) is None
    got_x, got_a = hh.to_ncdhw(out[0]).double(), hh.to_ncdhw(out[1]).double()
    tol = {"f16": 1.5e-3, "bf16": 1.2e-2}[dtype]
    ex = (got_x - x).abs().max().item()
    assert ex <= tol * (x.abs().max().item() + 1e-9), "%s[%s] trunk err %.3e" % (name, dtype, ex)
    ea = (got_a - want_a).abs().max().item()
    assert ea <= 1.5 * tol * (want_a.abs().max().item() + 1e-9), "%s[%s] a err %.3e" % (name, dtype, ea)   # a trunk value on a rounding boundary may round the other way


@pytest.mark.parametrize("env", [("AF_CA_CWL", "0"), ("AF_C111_WC64", "0")])
def test_ab_forms_behind_environment_switches_stay_correct(env, monkeypatch):
    """The forms round 4 replaced stay in the library for A/B runs (conv_ca with per-wave fragment loads of the c weights: AF_CA_CWL=0;
    the K = 256 stream on 32-channel wave columns: AF_C111_WC64=0): the same layers, bit-identical results to the shipped forms
    (the arithmetic and its order do not change, only how operands and results travel)."""
    dtype = "bf16"
    tdt = hh.TORCH_DT[dtype]
    if env[0] == "AF_CA_CWL":
        seed = 777
        lay = [("c.weight", (256, 64, 1, 1, 1), "float32"), ("a.weight", (64, 256, 3, 1, 1), "float32")]
        for p_, ch in (("c_bn", 256), ("a_bn", 64)):
            lay += [(p_ + s_, (ch,), "float32") for s_ in (".weight", ".bias", ".running_mean", ".running_var")]
        sd = synth.fill_layout(lay, seed)
        b = synth.synthetic_tensor((2, 64, 32, 64, 66), seed).to(tdt).float()
        res = synth.synthetic_tensor((2, 256, 32, 64, 66), seed + 1).to(tdt).float()
        run = lambda: hh.conv_ca(hh.to_ndhwc(b, dtype), sd["c.weight"], hh.fold_bn(sd, "c_bn"), hh.to_ndhwc(res, dtype), sd["a.weight"],
                                 hh.fold_bn(sd, "a_bn"), dtype)
        ref = [t_.clone() for t_ in run()]
        monkeypatch.setenv(*env)
        got = run()
        assert all(torch.equal(g, r) for g, r in zip(got, ref))
    else:
        seed = 778
        lay = [("w.weight", (1024, 256, 1, 1, 1), "float32"), ("bn.weight", (1024,), "float32"), ("bn.bias", (1024,), "float32"),
               ("bn.running_mean", (1024,), "float32"), ("bn.running_var", (1024,), "float32")]
        sd = synth.fill_layout(lay, seed)
        x = synth.synthetic_tensor((1, 256, 4, 91, 93), seed).to(tdt).float()
        res = synth.synthetic_tensor((1, 1024, 4, 91, 93), seed + 1).to(tdt).float()
        run = lambda: hh.conv_bn_act(hh.to_ndhwc(x, dtype), sd["w.weight"], *hh.fold_bn(sd, "bn"), (1, 1, 1), (0, 0, 0), True, dtype,
                                     residual=hh.to_ndhwc(res, dtype))
        ref = run().clone()
        assert hh.conv_bn_act.last_variant == 10
        monkeypatch.setenv(*env)
        assert torch.equal(run(), ref)


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("x_sub", [2, 1])
@pytest.mark.parametrize("name,ctrunk,dims", [
    ("s2_to_s3", 256, (3, 32, 56, 56)),           # 4 K slabs; 28 x 14 tiles per clip (2 rows x 4 columns each): 1 176 >= 4 per CU
    ("two_slabs_odd_patch_count", 128, (3, 32, 52, 60)),
])
def test_fused_c_pool_a_vs_oracle(name, ctrunk, dims, x_sub, dtype):
    """af_conv3d_cpa_bn_act - the s2 -> s3 boundary in one launch: x = relu(bn_c(c(b)) + res), pathway0_pool (max over frame
    pairs), a_out = relu(bn_a(a3x1x1(pooled))) - against the oracle's conv_bn_act / max_pool3d in fp64; the pooled trunk is
    rounded to the storage type on both sides before the temporal conv (rounding is monotonic: pooling before or after it is
    the same).  x_sub = 2: only the even (h, w) positions of the pooled trunk are stored, packed - what the next stage's
    stride-(1,2,2) projection shortcut reads."""
    seed = 5200 + sum(map(ord, name))
    n, t, h, w = dims
    lay = [("c.weight", (ctrunk, 64, 1, 1, 1), "float32"), ("a.weight", (128, ctrunk, 3, 1, 1), "float32")]
    for p_, ch in (("c_bn", ctrunk), ("a_bn", 128)):
        lay += [(p_ + s_, (ch,), "float32") for s_ in (".weight", ".bias", ".running_mean", ".running_var")]
    sd = synth.fill_layout(lay, seed)
    tdt = hh.TORCH_DT[dtype]
    b = synth.synthetic_tensor((n, 64, t, h, w), seed).to(tdt).float()
    res = synth.synthetic_tensor((n, ctrunk, t, h, w), seed + 1).to(tdt).float()
    for k in ("c.weight", "a.weight"):
        sd[k] = sd[k].to(tdt).float()
    sd64 = {k: v.double() for k, v in sd.items()}
    x = F.relu(oracle.conv_bn_act(b.double(), sd64["c.weight"], sd64, "c_bn", (1, 1, 1), (0, 0, 0), False) + res.double())
    xp = F.max_pool3d(x, (2, 1, 1), (2, 1, 1))
    xr = xp.to(tdt).double()                                    # the one rounding of the trunk
    want_a = oracle.conv_bn_act(xr, sd64["a.weight"], sd64, "a_bn", (1, 1, 1), (1, 0, 0), True)
    out = hh.conv_cpa(hh.to_ndhwc(b, dtype), sd["c.weight"], hh.fold_bn(sd, "c_bn"), hh.to_ndhwc(res, dtype), sd["a.weight"],
                      hh.fold_bn(sd, "a_bn"), dtype, x_sub)
    assert out is not None, "the library should fuse this pair"
    # frames with an odd row count or a width that is not a multiple of 4 keep their two launches
    assert hh.conv_cpa(hh.to_ndhwc(b[..., :w - 2].contiguous(), dtype), sd["c.weight"], hh.fold_bn(sd, "c_bn"),
                       hh.to_ndhwc(res[..., :w - 2].contiguous(), dtype), sd["a.weight"], hh.fold_bn(sd, "a_bn"), dtype, x_sub) is None
    got_x, got_a = hh.to_ncdhw(out[0]).double(), hh.to_ncdhw(out[1]).double()
    want_x = xp[..., ::2, ::2] if x_sub == 2 else xp
    tol = {"f16": 1.5e-3, "bf16": 1.2e-2}[dtype]
    assert got_x.shape == want_x.shape
    ex = (got_x - want_x).abs().max().item()
    assert ex <= tol * (xp.abs().max().item() + 1e-9), "%s[%s] trunk err %.3e" % (name, dtype, ex)
    ea = (got_a - want_a).abs().max().item()
    assert ea <= 1.5 * tol * (want_a.abs().max().item() + 1e-9), "%s[%s] a err %.3e" % (name, dtype, ea)   # a trunk value on a rounding boundary may round the other way


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_fused_c_shortcut_a_vs_oracle(dtype):
    """the projection-block form of af_conv3d_ca_bn_act: x = relu(bn_c(c(b)) + bn_1(branch1(x0))), a_out = relu(bn_a(a3x1x1(x))) -
    block 0 of s2 and the first conv of block 1 in one launch.  The BN scales are folded into the 16-bit weights there (two convs
    share an accumulator), so the oracle folds them the same way before rounding."""
    name, ctrunk, (n, t, h, w) = "s2_res0_T32", 256, (2, 32, 64, 66)
    seed = 6000 + sum(map(ord, name))
    lay = [("c.weight", (ctrunk, 64, 1, 1, 1), "float32"), ("b1.weight", (ctrunk, 64, 1, 1, 1), "float32"),
           ("a.weight", (64, ctrunk, 3, 1, 1), "float32")]
    for p_, ch in (("c_bn", ctrunk), ("b1_bn", ctrunk), ("a_bn", 64)):
        lay += [(p_ + s_, (ch,), "float32") for s_ in (".weight", ".bias", ".running_mean", ".running_var")]
    sd = synth.fill_layout(lay, seed)
    tdt = hh.TORCH_DT[dtype]
    b = synth.synthetic_tensor((n, 64, t, h, w), seed).to(tdt).float()
    x0 = synth.synthetic_tensor((n, 64, t, h, w), seed + 1).to(tdt).float()
    sd["a.weight"] = sd["a.weight"].to(tdt).float()
    bn_c, bn_1 = hh.fold_bn(sd, "c_bn"), hh.fold_bn(sd, "b1_bn")
    fold = lambda wk, bn: (sd[wk] * bn[0].cpu()[:ctrunk].view(-1, 1, 1, 1, 1)).to(tdt).double()      # scale folded in fp32, rounded once
    x = F.conv3d(b.double(), fold("c.weight", bn_c)) + F.conv3d(x0.double(), fold("b1.weight", bn_1))
    x = F.relu(x + (bn_c[1] + bn_1[1]).cpu()[:ctrunk].double().view(1, -1, 1, 1, 1))
    sd64 = {k: v.double() for k, v in sd.items()}
    want_a = oracle.conv_bn_act(x.to(tdt).double(), sd64["a.weight"], sd64, "a_bn", (1, 1, 1), (1, 0, 0), True)
    out = hh.conv_ca(hh.to_ndhwc(b, dtype), sd["c.weight"], bn_c, None, sd["a.weight"], hh.fold_bn(sd, "a_bn"), dtype,
                     x0_ndhwc=hh.to_ndhwc(x0, dtype), w1_oidhw=sd["b1.weight"], bn_1=bn_1)
    assert out is not None, "the library should fuse this pair"
    got_x, got_a = hh.to_ncdhw(out[0]).double(), hh.to_ncdhw(out[1]).double()
    tol = {"f16": 1.5e-3, "bf16": 1.2e-2}[dtype]
    assert (got_x - x).abs().max().item() <= tol * (x.abs().max().item() + 1e-9)
    assert (got_a - want_a).abs().max().item() <= 1.5 * tol * (want_a.abs().max().item() + 1e-9)
