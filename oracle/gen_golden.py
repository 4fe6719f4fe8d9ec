"""TEST INFRASTRUCTURE — generates tests/golden/* from the imported reference.

Run in the BUILD CONTAINER ONLY (needs /root/reference):

    python oracle/gen_golden.py

Every number written here is produced by the reference's own, unmodified code
(``PluginLoader.get_classifier('i3d_ori')`` and the layer modules under
altfreezing/slowfast/models), fed with seeded synthetic weights/inputs from the
product's ``synth`` recipe.  Only data (inputs' seeds + hashes, expected outputs)
is committed; no reference source travels.

Fixtures
  layout.json       the reference's ``network.state_dict()`` keys / shapes / dtypes
  f1_logits.json    full-size logits, fp32 and fp64, W(seed=0), three seeded clips
  f1b_logits.json   B=16 logits of BASELINE config[1]'s batch (W(0)) + 8 clips on the "hot" checkpoint W(3, hot), fp32/fp64
  f2_stages.npz     per-stage statistics + sampled activations for clip 0
  f3_kats.npz/.json per-layer-class known-answer tests on small tensors
  f9_dualrgb.*      the reference's tri-modal DualEncoderRGB (dualrun/model/dual_rgb.py): logits fp32 / fp64, masked / unmasked /
                    broadcast V
  f4_load.json      behaviour table of ``ModelBase.load`` on crafted checkpoints
"""
import json
from collections import OrderedDict
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import af_mi355x  # noqa: E402  (alias of the product package; only its synth/arch data recipes are used)
from af_mi355x import arch, synth  # noqa: E402
import ref_import  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
WEIGHT_SEED = 0
CLIP_SEED = 2026


def _layout_of(module):
    return [(k, list(v.shape), str(v.dtype).replace("torch.", "")) for k, v in module.state_dict().items()]


def gen_layout(clf):
    lay = _layout_of(clf.network)
    with open(os.path.join(GOLD, "layout.json"), "w") as f:
        json.dump({"source": "reference PluginLoader.get_classifier('i3d_ori')().network.state_dict()",
                   "num_keys": len(lay),
                   "num_params": int(sum(p.numel() for p in clf.network.parameters())),
                   "entries": lay}, f)
    mine = [(k, list(s), d) for k, s, d in arch.state_dict_layout(arch.i3d_r50_spec())]
    assert mine == lay, "product architecture table disagrees with the reference state_dict layout"
    print("layout: %d keys, matches arch.state_dict_layout" % len(lay))


def _sample_idx(numel, n=64, seed=7):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, numel, (n,), generator=g)


def gen_f1_f2(clf):
    sd = synth.synthetic_state_dict(seed=WEIGHT_SEED)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "w0.pth")
        torch.save(sd, path)
        ok, epoch = clf.load(path)            # the reference's own loader
        assert ok and epoch == -1
    got = clf.network.state_dict()
    assert all(torch.equal(got[k], sd[k]) for k in sd)

    clips = [("uniform", CLIP_SEED, 0), ("uniform", CLIP_SEED, 1), ("smooth", CLIP_SEED, 0)]
    entries = []
    stage_out = {}
    hooks = []
    res = clf.network.resnet
    for name, mod in (("s1", res.s1), ("s2", res.s2), ("pool", res.pathway0_pool), ("s3", res.s3),
                      ("s4", res.s4), ("s5", res.s5), ("avgpool", res.head.pathway0_avgpool)):
        def mk(n):
            def hook(m, inp, out):
                o = out[0] if isinstance(out, (list, tuple)) else out
                stage_out[n] = o.detach().clone()
            return hook
        hooks.append(mod.register_forward_hook(mk(name)))

    f2 = {}
    for ci, (kind, seed, index) in enumerate(clips):
        u8 = synth.synthetic_clips_u8(index + 1, seed=seed, kind=kind)[index:index + 1]
        x = synth.normalize_like_callers(u8)
        assert x.is_contiguous(memory_format=torch.channels_last_3d)
        with torch.no_grad():
            y32 = clf(x)["final_output"]
        if ci == 0:
            for n, t in stage_out.items():
                flat = t.flatten()
                idx = _sample_idx(flat.numel())
                f2[n + "_shape"] = np.array(t.shape, dtype=np.int64)
                f2[n + "_stats"] = np.array([flat.double().mean().item(), flat.double().abs().mean().item(),
                                             flat.max().item(), flat.min().item()], dtype=np.float64)
                f2[n + "_idx"] = idx.numpy()
                f2[n + "_val"] = flat[idx].numpy()
        clf64 = clf.double()
        with torch.no_grad():
            y64 = clf64(x.double())["final_output"]
        clf.float()
        entries.append({"kind": kind, "seed": seed, "index": index, "clip_sha256": synth.tensor_sha256(u8),
                        "logit_f32": float(y32[0, 0]), "logit_f32_hex": y32[0, 0].item().hex(),
                        "logit_f64": float(y64[0, 0])})
        print("F1 clip", kind, index, "logit f32 %.9g f64 %.12g" % (y32[0, 0].item(), y64[0, 0].item()))
    # batch invariance of the reference (B=2 equals two B=1 runs)
    u8 = synth.synthetic_clips_u8(2, seed=CLIP_SEED, kind="uniform")
    with torch.no_grad():
        yb = clf(synth.normalize_like_callers(u8))["final_output"]
    batch2 = [float(yb[0, 0]), float(yb[1, 0])]
    for h in hooks:
        h.remove()
    with open(os.path.join(GOLD, "f1_logits.json"), "w") as f:
        json.dump({"source": "reference i3d_ori forward, PyTorch CPU, weights W(seed) loaded via ModelBase.load",
                   "torch": torch.__version__, "weights_seed": WEIGHT_SEED,
                   "weights_sha256": synth.state_dict_sha256(sd), "clips": entries,
                   "batch2_uniform_logits_f32": batch2}, f, indent=1)
    np.savez_compressed(os.path.join(GOLD, "f2_stages.npz"), **f2)


def gen_f1b(clf):
    """F1b: (a) BASELINE config[1]'s own batch - W(0), B=16 uniform clips seed 2026 (exactly bench.py's rank-0 input) in ONE
    reference forward; (b) the "hot" checkpoint W(3, hot) whose logits are O(10..40) and move between clips, 8 clips
    (4 uniform + 4 smooth), fp32 and fp64.  Written by the reference's own forward, as F1."""
    out = {"source": "reference i3d_ori forward, PyTorch CPU, weights loaded via ModelBase.load", "torch": torch.__version__}

    def load(sd):
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "w.pth")
            torch.save({"state_dict": sd}, path)
            ok, epoch = clf.load(path)
            assert ok and epoch == -1

    sd0 = synth.synthetic_state_dict(seed=WEIGHT_SEED)
    load(sd0)
    u8 = synth.synthetic_clips_u8(16, seed=CLIP_SEED, kind="uniform")
    with torch.no_grad():
        y = clf(synth.normalize_like_callers(u8))["final_output"]
    out["batch16"] = {"weights_seed": WEIGHT_SEED, "recipe": "mild", "weights_sha256": synth.state_dict_sha256(sd0),
                      "kind": "uniform", "seed": CLIP_SEED, "clips_sha256": synth.tensor_sha256(u8),
                      "logits_f32": [float(v) for v in y.flatten()]}
    print("F1b batch16", out["batch16"]["logits_f32"])
    sd3 = synth.synthetic_state_dict(seed=3, recipe="hot")
    load(sd3)
    u8 = torch.cat([synth.synthetic_clips_u8(4, seed=11, kind="uniform"), synth.synthetic_clips_u8(4, seed=12, kind="smooth")])
    x = synth.normalize_like_callers(u8)
    with torch.no_grad():
        y32 = clf(x)["final_output"]
        clf64 = clf.double()
        y64 = torch.cat([clf64(x[i:i + 1].double())["final_output"] for i in range(x.shape[0])])
        clf.float()
    out["hot"] = {"weights_seed": 3, "recipe": "hot", "weights_sha256": synth.state_dict_sha256(sd3),
                  "clips": [["uniform", 11, 4], ["smooth", 12, 4]], "clips_sha256": synth.tensor_sha256(u8),
                  "logits_f32": [float(v) for v in y32.flatten()], "logits_f64": [float(v) for v in y64.flatten()]}
    print("F1b hot f32", out["hot"]["logits_f32"])
    print("F1b hot f64", out["hot"]["logits_f64"])
    with open(os.path.join(GOLD, "f1b_logits.json"), "w") as f:
        json.dump(out, f, indent=1)


def _fill(module, seed, prefix, final_bn=(), linear=()):
    lay = [(prefix + k, s, d) for k, s, d in _layout_of(module)]
    sd = synth.fill_layout(lay, seed, final_bn=[prefix + b for b in final_bn], linear=[prefix + l for l in linear])
    module.load_state_dict({k[len(prefix):]: v for k, v in sd.items()})
    return lay


def gen_f3():
    ref = ref_import.reference_modules()
    RB = ref.resnet_helper.ResBlock
    BT = ref.resnet_helper.BottleneckTransform
    cases = []
    arrays = {}

    def run(name, module, in_shape, seed, meta, final_bn=(), linear=(), scale=1.0):
        module.eval()
        lay = _fill(module, seed, name + ".", final_bn, linear)
        x = synth.synthetic_tensor(in_shape, seed, scale)
        with torch.no_grad():
            y = module(x.clone())
        arrays[name + "_out"] = y.numpy()
        meta = dict(meta)
        meta.update({"name": name, "seed": seed, "in_shape": list(in_shape), "in_scale": scale,
                     "in_sha256": synth.tensor_sha256(x), "layout": lay,
                     "final_bn": list(final_bn), "linear": list(linear), "out_shape": list(y.shape)})
        cases.append(meta)
        print("F3", name, tuple(in_shape), "->", tuple(y.shape))

    run("stem", ref.stem_helper.ResNetBasicStem(3, 64, [5, 7, 7], [1, 2, 2], [2, 3, 3]),
        (1, 3, 6, 40, 40), 11, {"kind": "stem"})
    run("stem_b2", ref.stem_helper.ResNetBasicStem(3, 64, [5, 7, 7], [1, 2, 2], [2, 3, 3]),
        (2, 3, 5, 34, 38), 12, {"kind": "stem"})
    run("block_proj_s1", RB(64, 256, 3, 1, BT, 64), (1, 64, 4, 10, 10), 21,
        {"kind": "block", "stride": 1}, final_bn=["branch2.c_bn"])
    run("block_proj_s2", RB(256, 512, 3, 2, BT, 128), (1, 256, 3, 8, 8), 22,
        {"kind": "block", "stride": 2}, final_bn=["branch2.c_bn"])
    run("block_proj_s2_odd", RB(256, 512, 1, 2, BT, 128), (2, 256, 2, 7, 9), 23,
        {"kind": "block", "stride": 2}, final_bn=["branch2.c_bn"])
    run("block_id_t1", RB(256, 256, 1, 1, BT, 64), (1, 256, 3, 6, 6), 24,
        {"kind": "block", "stride": 1}, final_bn=["branch2.c_bn"])
    run("block_id_t3", RB(256, 256, 3, 1, BT, 64), (2, 256, 5, 5, 7), 25,
        {"kind": "block", "stride": 1}, final_bn=["branch2.c_bn"])
    run("pool_t2", torch.nn.MaxPool3d(kernel_size=[2, 1, 1], stride=[2, 1, 1], padding=[0, 0, 0]),
        (2, 64, 6, 5, 5), 31, {"kind": "maxpool", "kernel": [2, 1, 1], "stride": [2, 1, 1], "pad": [0, 0, 0]})
    run("pool_t2_odd", torch.nn.MaxPool3d(kernel_size=[2, 1, 1], stride=[2, 1, 1], padding=[0, 0, 0]),
        (1, 64, 7, 3, 4), 32, {"kind": "maxpool", "kernel": [2, 1, 1], "stride": [2, 1, 1], "pad": [0, 0, 0]})
    run("pool_s133", torch.nn.MaxPool3d(kernel_size=[1, 3, 3], stride=[1, 2, 2], padding=[0, 1, 1]),
        (1, 64, 2, 9, 12), 33, {"kind": "maxpool", "kernel": [1, 3, 3], "stride": [1, 2, 2], "pad": [0, 1, 1]})

    class _Head(torch.nn.Module):                       # ResNetBasicHead takes a 1-list
        def __init__(self, h):
            super().__init__()
            self.h = h

        def forward(self, x):
            return self.h([x])

        def state_dict(self, *a, **k):
            return self.h.state_dict(*a, **k)

        def load_state_dict(self, sd, *a, **k):
            return self.h.load_state_dict(sd, *a, **k)

    run("head", _Head(ref.head_helper.ResNetBasicHead([128], 1, [[2, 3, 3]], dropout_rate=0.5)),
        (2, 128, 2, 3, 3), 41, {"kind": "head", "pool": [2, 3, 3]}, linear=["projection"])
    run("head_multi", _Head(ref.head_helper.ResNetBasicHead([128], 1, [[2, 3, 3]], dropout_rate=0.5)),
        (1, 128, 3, 3, 4), 42, {"kind": "head", "pool": [2, 3, 3]}, linear=["projection"])

    class _Fuse(torch.nn.Module):                       # FuseFastToSlow takes [slow, fast]
        def __init__(self, f):
            super().__init__()
            self.f = f

        def forward(self, x):
            xs, xf = x[:, :16, ::4], x                   # any slow tensor; fusion only reads fast
            return self.f([xs, xf])[0][:, 16:]

        def state_dict(self, *a, **k):
            return self.f.state_dict(*a, **k)

        def load_state_dict(self, sd, *a, **k):
            return self.f.load_state_dict(sd, *a, **k)

    run("fuse_f2s", _Fuse(ref.video_model_builder.FuseFastToSlow(64, 2, 5, 4)),
        (1, 64, 16, 6, 6), 51, {"kind": "fuse", "ratio": 2, "kernel": 5, "alpha": 4})

    np.savez_compressed(os.path.join(GOLD, "f3_kats.npz"), **arrays)
    with open(os.path.join(GOLD, "f3_kats.json"), "w") as f:
        json.dump({"source": "reference layer modules (stem_helper/resnet_helper/head_helper/"
                             "video_model_builder.FuseFastToSlow), eval mode, PyTorch CPU fp32",
                   "cases": cases}, f)


def gen_f4(clf):
    """ModelBase.load behaviour table (altfreezing/model/_base.py:39-104)."""
    base = synth.synthetic_state_dict(seed=3)
    probe = "resnet.head.projection.bias"
    other = "resnet.s1.pathway0_stem.bn.bias"
    rows = []
    with tempfile.TemporaryDirectory() as td:
        def case(name, obj=None, raw_bytes=None, path=None):
            clf.load_state_dict({k: torch.zeros_like(v) for k, v in clf.state_dict().items()})
            p = path or os.path.join(td, name + ".pth")
            if obj is not None:
                torch.save(obj, p)
            elif raw_bytes is not None:
                with open(p, "wb") as f:
                    f.write(raw_bytes)
            try:
                ret = clf.load(p)
                err = None
            except Exception as e:                       # noqa: BLE001
                ret, err = None, type(e).__name__
            cur = clf.network.state_dict()
            rows.append({"case": name, "ret": list(ret) if ret is not None else None, "raises": err,
                         "probe_loaded": bool(torch.equal(cur[probe], base[probe])),
                         "other_loaded": bool(torch.equal(cur[other], base[other]))})
            print("F4", rows[-1])

        case("raw", obj=base)
        case("wrap_state_dict", obj={"state_dict": base, "epoch": 7})
        case("wrap_classifier_state_dict", obj={"classifier_state_dict": base})
        case("wrap_model_state_dict", obj={"model_state_dict": base})
        for pfx in ("module.", "network.", "_warped_network."):
            case("prefix_" + pfx.strip("._"), obj={pfx + k: v for k, v in base.items()})
        case("prefix_double", obj={"module.network." + k: v for k, v in base.items()})
        case("extra_key", obj=dict(base, **{"resnet.extra.weight": torch.ones(3)}))
        missing = {k: v for k, v in base.items() if k != probe}
        case("missing_probe", obj=missing)
        bad = dict(base)
        bad[probe] = torch.ones(5)
        case("shape_mismatch_probe", obj=bad)
        case("missing_file", path=os.path.join(td, "does_not_exist.pth"))
        case("epoch_arg_passthrough", obj=base)
        clf.load_state_dict({k: torch.zeros_like(v) for k, v in clf.state_dict().items()})
        p = os.path.join(td, "e.pth")
        torch.save(base, p)
        rows.append({"case": "epoch_kw", "ret": list(clf.load(p, epoch=12))})
    with open(os.path.join(GOLD, "f4_load.json"), "w") as f:
        json.dump({"source": "reference ModelBase.load on crafted checkpoints (weights W(seed=3)); "
                             "probe = resnet.head.projection.bias, other = resnet.s1.pathway0_stem.bn.bias",
                   "rows": rows}, f, indent=1)


def gen_slowfast():
    """F5: the reference's two-pathway SlowFast-R50 (video_model_builder.py:146-387).  No shipped plugin builds it;
    it is instantiated here straight from the reference's own config defaults."""
    ref_import.import_reference()
    from slowfast.config.defaults import get_cfg
    from slowfast.models.video_model_builder import SlowFast
    cfg = get_cfg()
    cfg.MODEL.ARCH = "slowfast"
    cfg.MODEL.NUM_CLASSES = 1
    cfg.DATA.NUM_FRAMES = 32
    cfg.RESNET.NUM_BLOCK_TEMP_KERNEL = [[3, 3], [4, 4], [6, 6], [3, 3]]
    cfg.RESNET.SPATIAL_STRIDES = [[1, 1], [2, 2], [2, 2], [2, 2]]
    cfg.RESNET.SPATIAL_DILATIONS = [[1, 1]] * 4
    cfg.NONLOCAL.LOCATION = [[[], []]] * 4
    cfg.NONLOCAL.GROUP = [[1, 1]] * 4
    net = SlowFast(cfg).eval()
    spec = arch.slowfast_r50_spec()
    lay = [("resnet." + k, list(v.shape), str(v.dtype).replace("torch.", "")) for k, v in net.state_dict().items()]
    mine = [(k, list(sh), d) for k, sh, d in arch.state_dict_layout(spec)]
    assert mine == lay, "product SlowFast table disagrees with the reference state_dict layout"
    sd = synth.synthetic_state_dict(spec, seed=WEIGHT_SEED)
    net.load_state_dict({k[len("resnet."):]: v for k, v in sd.items()})
    alpha = cfg.SLOWFAST.ALPHA

    stage_out, hooks = {}, []
    for name in ("s1_fuse", "s2_fuse", "s3_fuse", "s4_fuse", "s5"):
        def mk(n):
            def hook(m, inp, out):
                stage_out[n] = [o.detach().clone() for o in out]
            return hook
        hooks.append(getattr(net, name).register_forward_hook(mk(name)))
    entries, f5 = [], {}
    for ci, (kind, seed, index) in enumerate([("uniform", CLIP_SEED, 0), ("smooth", CLIP_SEED, 0)]):
        u8 = synth.synthetic_clips_u8(index + 1, seed=seed, kind=kind)[index:index + 1]
        x = synth.normalize_like_callers(u8)
        with torch.no_grad():
            y32 = net([x[:, :, ::alpha], x])
        if ci == 0:
            for n, (ts, tf) in stage_out.items():
                for tag, t in (("slow", ts), ("fast", tf)):
                    flat = t.flatten()
                    idx = _sample_idx(flat.numel())
                    f5["%s_%s_shape" % (n, tag)] = np.array(t.shape, dtype=np.int64)
                    f5["%s_%s_absmean" % (n, tag)] = np.array([flat.double().abs().mean().item()])
                    f5["%s_%s_idx" % (n, tag)] = idx.numpy()
                    f5["%s_%s_val" % (n, tag)] = flat[idx].numpy()
        net.double()
        with torch.no_grad():
            y64 = net([x[:, :, ::alpha].double(), x.double()])
        net.float()
        entries.append({"kind": kind, "seed": seed, "index": index, "clip_sha256": synth.tensor_sha256(u8),
                        "logit_f32": float(y32[0, 0]), "logit_f64": float(y64[0, 0])})
        print("F5 slowfast clip", kind, "logit f32 %.9g f64 %.12g" % (y32[0, 0].item(), y64[0, 0].item()))
    for h in hooks:
        h.remove()
    with open(os.path.join(GOLD, "f5_slowfast.json"), "w") as f:
        json.dump({"source": "reference slowfast.models.video_model_builder.SlowFast (R50, alpha 8, beta_inv 8), "
                             "input [x[:,:,::8], x], PyTorch CPU, weights W(seed) in its state_dict layout",
                   "num_keys": len(lay), "num_params": int(sum(p.numel() for p in net.parameters())),
                   "alpha": alpha, "weights_seed": WEIGHT_SEED, "weights_sha256": synth.state_dict_sha256(sd),
                   "clips": entries}, f, indent=1)
    np.savez_compressed(os.path.join(GOLD, "f5_slowfast_stages.npz"), **f5)


def gen_ftcn():
    """F6: the reference's FTCN-TT plugin (`classifier_type: i3d_temporal_var_fix_dropout_tt_cfg`, setting/ftcn_tt.yaml).
    Run in its OWN process (`python -m oracle.gen_golden --ftcn`): the reference config is a process-wide singleton.
    Two harness patches, neither touching arithmetic: `timm.models.layers.trunc_normal_` (absent package; only seeds the
    random init that the fixture overwrites) and the plugin's module-level list of nn.Conv3d constructor parameters,
    from which this torch's `device` / `dtype` entries are dropped (they are not attributes of a built module, so the
    unmodified plugin raises AttributeError on torch >= 1.9)."""
    import importlib
    ref_import._install_shims()
    for n in ("timm", "timm.models"):
        ref_import._mod(n)
    ref_import._mod("timm.models.layers").trunc_normal_ = torch.nn.init.trunc_normal_
    cfg, _ = ref_import.import_reference("ftcn_tt.yaml")
    mod = importlib.import_module("model.classifier." + cfg.classifier_type)
    mod.parameters = [p for p in mod.parameters if p not in ("device", "dtype")]
    clf = mod.Classifier().eval()
    net = clf.network
    spec = arch.ftcn_tt_spec()
    lay = [(k, list(v.shape), str(v.dtype).replace("torch.", "")) for k, v in net.state_dict().items()]
    mine = [(k, list(sh), d) for k, sh, d in arch.state_dict_layout(spec)]
    assert mine == lay, "product FTCN-TT table disagrees with the reference state_dict layout"
    sd = synth.synthetic_state_dict(spec, seed=WEIGHT_SEED)
    net.load_state_dict(sd)

    stage_out, hooks = {}, []
    for name in ("s1", "s2", "s3", "s4"):
        def mk(n):
            def hook(m, inp, out):
                stage_out[n] = out[0].detach().clone()
            return hook
        hooks.append(getattr(net.resnet, name).register_forward_hook(mk(name)))
    hooks.append(net.resnet.head.time_T.register_forward_hook(
        lambda m, inp, out: stage_out.__setitem__("tokens", inp[0].detach().clone())))
    entries, f6 = [], {}
    for ci, (kind, seed, index) in enumerate([("uniform", CLIP_SEED, 0), ("smooth", CLIP_SEED, 0)]):
        u8 = synth.synthetic_clips_u8(index + 1, seed=seed, kind=kind)[index:index + 1]
        x = synth.normalize_like_callers(u8)
        with torch.no_grad():
            y32 = clf(x)["final_output"]
        if ci == 0:
            for n, t in stage_out.items():
                flat = t.flatten()
                idx = _sample_idx(flat.numel())
                f6["%s_shape" % n] = np.array(t.shape, dtype=np.int64)
                f6["%s_absmean" % n] = np.array([flat.double().abs().mean().item()])
                f6["%s_idx" % n] = idx.numpy()
                f6["%s_val" % n] = flat[idx].numpy()
            # known-answer vector for the transformer head alone: its real input and output
            f6["head_tokens"] = stage_out["tokens"].numpy()
            f6["head_logit"] = y32.numpy()
        net.double()
        with torch.no_grad():
            y64 = net(x.double())["final_output"]
        net.float()
        entries.append({"kind": kind, "seed": seed, "index": index, "clip_sha256": synth.tensor_sha256(u8),
                        "logit_f32": float(y32[0, 0]), "logit_f64": float(y64[0, 0])})
        print("F6 ftcn_tt clip", kind, "logit f32 %.9g f64 %.12g" % (y32[0, 0].item(), y64[0, 0].item()))
    for h in hooks:
        h.remove()
    with open(os.path.join(GOLD, "f6_ftcn.json"), "w") as f:
        json.dump({"source": "reference model/classifier/i3d_temporal_var_fix_dropout_tt_cfg.py Classifier() with "
                             "setting/ftcn_tt.yaml (stop_point 5, depth 1, patch_type time), PyTorch CPU, weights W(seed) "
                             "in its state_dict layout",
                   "num_keys": len(lay), "num_params": int(sum(p.numel() for p in net.parameters())),
                   "weights_seed": WEIGHT_SEED, "weights_sha256": synth.state_dict_sha256(sd), "clips": entries}, f, indent=1)
    np.savez_compressed(os.path.join(GOLD, "f6_ftcn_stages.npz"), **f6)


def gen_dualrun():
    """F7: the reference's dualrun ``DualEncoderAU_LMK`` (dualrun/model/dual_encoder.py), built like dualrun/cli/run.py:175-187
    with checkpoints/test7/args.json (d_model 256, 4 layers, 4 heads, ff_dim 768, T 8).  The module is pure torch and is
    imported straight from its file (its package name ``model`` would collide with altfreezing's)."""
    import importlib.util
    from af_mi355x import dualrun
    path = os.path.join(ref_import.REFERENCE_ROOT, "dualrun", "model", "dual_encoder.py")
    spec_ = importlib.util.spec_from_file_location("ref_dual_encoder", path)
    mod = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(mod)
    sp = dualrun.DualSpec()
    net = mod.DualEncoderAU_LMK(au_dim=sp.au_dim, lmk_dim=sp.lmk_dim, d_model=sp.d_model, depth=sp.depth, heads=sp.heads,
                                mlp_ratio=float(sp.ff) / sp.d_model, dropout=0.15, pool_tau=sp.pool_tau).eval()
    lay = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    assert lay == [(k, tuple(sh)) for k, sh in dualrun.dual_state_dict_layout(sp)], "product dualrun table disagrees with the reference"
    sd = dualrun.dual_synthetic_state_dict(sp, seed=WEIGHT_SEED)
    net.load_state_dict(sd)
    out = {}
    for tag, batch, frames, ragged in (("b6_t8", 6, 8, True), ("b3_t8_full", 3, 8, False), ("b2_t5", 2, 5, True)):
        A, L, lengths = dualrun.synthetic_dual_inputs(batch, sp, frames=frames, seed=CLIP_SEED)
        if not ragged:
            lengths = None
        if tag == "b2_t5":
            lengths = torch.tensor([5, 0], dtype=torch.int32)          # a clip with no valid frame (keeps frame 0)
        with torch.no_grad():
            o32 = net(A, L, lengths, return_z=True)
        net.double()
        with torch.no_grad():
            o64 = net(A.double(), L.double(), lengths, return_z=True)
        net.float()
        out[tag + "_lengths"] = np.array([-1]) if lengths is None else lengths.numpy().astype(np.int64)
        out[tag + "_logits_f32"] = o32["bin_logits"].numpy()
        out[tag + "_logits_f64"] = o64["bin_logits"].numpy()
        out[tag + "_z_f32"] = o32["z"].numpy()
        print("F7 dualrun", tag, "logits", o32["bin_logits"].numpy().round(5).tolist())
    # GatedMoE (dualrun/rgb/engine_rgb.py:369-384): fuses the RGB logit with the dual logit; seeded parameters and logits
    sys.path.insert(0, os.path.join(ref_import.REFERENCE_ROOT, "dualrun"))
    spec_e = importlib.util.spec_from_file_location("ref_engine_rgb", os.path.join(ref_import.REFERENCE_ROOT, "dualrun", "rgb", "engine_rgb.py"))
    eng = importlib.util.module_from_spec(spec_e)
    spec_e.loader.exec_module(eng)
    moe = eng.GatedMoE().eval()
    gm = torch.Generator().manual_seed(4711)
    msd = OrderedDict((k, (torch.randn(v.shape, generator=gm) * 0.8 + (1.5 if k.startswith("t_") else 0.0)))
                      for k, v in moe.state_dict().items())
    moe.load_state_dict(msd)
    zr, zd = torch.randn(9, 1, generator=gm) * 3, torch.randn(9, 1, generator=gm) * 3
    with torch.no_grad():
        zf, gate = moe(zr, zd)
    for k, v in msd.items():
        out["moe_w_" + k] = v.numpy()
    out["moe_z_rgb"], out["moe_z_dual"], out["moe_z"], out["moe_gate"] = zr.numpy(), zd.numpy(), zf.numpy(), gate.numpy()
    with open(os.path.join(GOLD, "f7_dualrun.json"), "w") as f:
        json.dump({"source": "reference dualrun/model/dual_encoder.py DualEncoderAU_LMK (au 36, lmk 132, d_model 256, depth 4, "
                             "heads 4, mlp_ratio 3.0, pool_tau 1.0), eval, PyTorch CPU, weights dual_synthetic_state_dict(seed), "
                             "inputs synthetic_dual_inputs(seed)",
                   "num_keys": len(lay), "num_params": int(sum(p.numel() for p in net.parameters())),
                   "weights_seed": WEIGHT_SEED, "inputs_seed": CLIP_SEED,
                   "weights_sha256": synth.state_dict_sha256(sd)}, f, indent=1)
    np.savez_compressed(os.path.join(GOLD, "f7_dualrun.npz"), **out)


def synthetic_aligner_case(rng, frames, size, mirrored=False):
    """per-frame (ldm5, ldm68, big box): a face whose 5 points are the aligner's standard points under a random similarity
    (+ per-frame jitter), boxes of slightly different origin / size per frame, as the trackers upstream produce"""
    std = np.array([[85.82991, 115.7792], [169.0532, 114.3381], [127.574, 167.0006], [90.6964, 204.7014], [167.3069, 203.3733]])
    ang, sc = rng.uniform(-0.5, 0.5), rng.uniform(0.6, 1.6)
    rot = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]]) * sc
    infos = []
    for _ in range(frames):
        p5 = std @ rot.T + rng.uniform(40, 60, size=(1, 2)) + rng.normal(0, 1.5, size=(5, 2))
        if mirrored:
            p5[:, 0] = 400 - p5[:, 0]
        p68 = p5.mean(0, keepdims=True) + rng.normal(0, 40 * sc, size=(68, 2))
        x0, y0 = rng.integers(100, 140, size=2)
        bw, bh = rng.integers(380, 460, size=2)
        infos.append((None, p5, p68, np.array([x0, y0, x0 + bw, y0 + bh], dtype=np.int64)))
    return infos


def gen_dualrun_rgb():
    """F9: the reference's tri-modal ``DualEncoderRGB`` (dualrun/model/dual_rgb.py:47-122) with ``rgb_from_features=True``
    (V = AltFreezing features, vis_dim 2048 = the i3d head's pooled vector), d_model 256, depth 4, heads 4 and
    ``ff_dim = 3.0`` (the constructor forwards ff_dim into BranchEncoder's mlp_ratio slot: 3.0 gives the 768-wide layers of
    checkpoints/test7/args.json).  Pure torch; imported as a two-file package so that its ``from .dual_encoder import``
    resolves (the directory's own package name ``model`` would collide with altfreezing's)."""
    import importlib.util
    import types
    from af_mi355x import dualrun
    mdir = os.path.join(ref_import.REFERENCE_ROOT, "dualrun", "model")
    pkg = types.ModuleType("ref_dualrun_model")
    pkg.__path__ = [mdir]                                   # the package's own __init__ is not run
    sys.modules["ref_dualrun_model"] = pkg
    mod = importlib.import_module("ref_dualrun_model.dual_rgb")
    vis = 2048
    net = mod.DualEncoderRGB(au_dim=36, lmk_dim=132, vis_dim=vis, d_model=256, depth=4, heads=4, ff_dim=3.0, dropout=0.1,
                             rgb_backbone=None, rgb_from_features=True).eval()
    sp = dualrun.DualSpec(36, 132, 256, 4, 4, 768, 0.7, 128)
    lay = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    assert lay == [(k, tuple(sh)) for k, sh in dualrun.dual_rgb_state_dict_layout(sp, vis)], "product DualEncoderRGB table disagrees with the reference"
    sd = dualrun.dual_rgb_synthetic_state_dict(sp, vis, seed=WEIGHT_SEED)
    net.load_state_dict(sd)
    out = {}
    for tag, batch, frames, tv, ragged in (("b6_t8", 6, 8, 8, True), ("b3_t8_nomask", 3, 8, 8, False), ("b4_t8_v1", 4, 8, 1, True)):
        A, L, lengths = dualrun.synthetic_dual_inputs(batch, sp, frames=frames, seed=CLIP_SEED + 1)
        gv = torch.Generator().manual_seed(CLIP_SEED + 77)
        V = torch.rand((batch, tv, vis), generator=gv) * 2.0          # post-ReLU average-pooled features are non-negative
        mask = net.lengths_to_mask(lengths, frames, torch.device("cpu")) if ragged else None
        with torch.no_grad():
            y32 = net(A, L, V, key_padding_mask=mask)
        net.double()
        with torch.no_grad():
            y64 = net(A.double(), L.double(), V.double(), key_padding_mask=mask)
        net.float()
        out[tag + "_lengths"] = lengths.numpy().astype(np.int64) if ragged else np.array([-1])
        out[tag + "_logits_f32"], out[tag + "_logits_f64"] = y32.numpy(), y64.numpy()
        print("F9 dualrun rgb", tag, "logits", y32.numpy().round(5).tolist())
    with open(os.path.join(GOLD, "f9_dualrgb.json"), "w") as f:
        json.dump({"source": "reference dualrun/model/dual_rgb.py DualEncoderRGB(au 36, lmk 132, vis 2048, d_model 256, depth 4, heads 4, "
                             "ff_dim 3.0, rgb_from_features True), eval, PyTorch CPU; weights dual_rgb_synthetic_state_dict(seed), "
                             "inputs synthetic_dual_inputs(seed + 1), V = 2 * rand (generator seed + 77)",
                   "num_keys": len(lay), "num_params": int(sum(p.numel() for p in net.parameters())),
                   "weights_seed": WEIGHT_SEED, "inputs_seed": CLIP_SEED + 1,
                   "weights_sha256": synth.state_dict_sha256(sd)}, f, indent=1)
    np.savez_compressed(os.path.join(GOLD, "f9_dualrgb.npz"), **out)


def gen_aligner():
    """F8: the similarity fit / landmark transform of the clip aligner (test_tools/warp_for_xray.py,
    test_tools/faster_crop_align_xray.py with images=None) - pure numpy in the reference.  Both files `import cv2` at the top;
    OpenCV is absent here, so an EMPTY stand-in module satisfies the import and is never called (cv2.warpAffine, the only
    use, is not on this path and stays unpinned: oracle/aligner_oracle.py header)."""
    import importlib
    import types
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    pkg = types.ModuleType("ref_test_tools")
    pkg.__path__ = [os.path.join(ref_import.ALTFREEZING_DIR, "test_tools")]      # the package's own __init__ is not run
    sys.modules["ref_test_tools"] = pkg
    fca = importlib.import_module("ref_test_tools.faster_crop_align_xray")
    wfx = importlib.import_module("ref_test_tools.warp_for_xray")
    rng = np.random.default_rng(20260104)
    out = {"std_points_256": wfx.std_points_256}
    for tag, frames, size, mirrored in (("t32_224", 32, 224, False), ("t1_256", 1, 256, False), ("t8_mirrored", 8, 224, True)):
        infos = synthetic_aligner_case(rng, frames, size, mirrored)
        out[tag + "_ldm5"] = np.array([i[1] for i in infos])
        out[tag + "_ldm68"] = np.array([i[2] for i in infos])
        out[tag + "_boxes"] = np.array([i[3] for i in infos])
        al = fca.FasterCropAlignXRay(size, return_ldm5=True)
        t5, t68 = al([(a, b.copy(), c.copy(), d.copy()) for a, b, c, d in infos], images=None)
        boxes = out[tag + "_boxes"]
        diff = boxes[:, :2] - boxes[:, :2].min(0)[None]
        tfm, trans = wfx.estimiate_batch_transform(out[tag + "_ldm5"] + diff[:, None, :], tgt_pts=al.std_points)
        out[tag + "_t5"], out[tag + "_t68"], out[tag + "_tfm"], out[tag + "_trans"] = t5, t68, tfm, trans
        print("F8 aligner", tag, "tfm", np.round(tfm, 4).tolist())
    np.savez_compressed(os.path.join(GOLD, "f8_aligner.npz"), **out)


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    clf = ref_import.build_reference_classifier()
    gen_layout(clf)
    gen_f3()
    gen_f4(clf)
    gen_f1_f2(clf)
    gen_f1b(clf)
    gen_slowfast()
    gen_dualrun()
    gen_dualrun_rgb()
    gen_aligner()


if __name__ == "__main__":
    if "--aligner" in sys.argv:
        os.makedirs(GOLD, exist_ok=True)
        gen_aligner()
    elif "--dualrun" in sys.argv:
        os.makedirs(GOLD, exist_ok=True)
        torch.set_num_threads(8)
        gen_dualrun()
    elif "--dualrun-rgb" in sys.argv:
        os.makedirs(GOLD, exist_ok=True)
        torch.set_num_threads(8)
        import importlib
        gen_dualrun_rgb()
    elif "--f1b" in sys.argv:
        os.makedirs(GOLD, exist_ok=True)
        torch.manual_seed(0)
        torch.set_num_threads(8)
        gen_f1b(ref_import.build_reference_classifier())
    elif "--ftcn" in sys.argv:
        os.makedirs(GOLD, exist_ok=True)
        torch.manual_seed(0)
        torch.set_num_threads(8)
        gen_ftcn()
    else:
        main()
