"""TEST INFRASTRUCTURE — CPU oracle for the AltFreezing ``i3d_ori`` forward path.

A plain PyTorch-CPU fp32 (or fp64) restatement of the reference's forward, driven
directly by a checkpoint ``state_dict`` (the 320-tensor layout of SURVEY.md 8a).
It is deliberately independent of the product package: the network structure is
recovered from the state_dict keys and weight shapes plus the handful of fixed
rules cited below, so that a mistake in the product's architecture table cannot
also hide in the checker.

Pinned: ``oracle/gen_golden.py`` runs the *imported reference model* in the build
container and commits its outputs under ``tests/golden/``; ``tests/test_oracle.py``
checks this restatement against those vectors (bit-exact for the per-layer
known-answer fixtures, <=1e-5 for full-size logits).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this file — it is the checker, never the product.

Reference lines each function follows:
  stem        altfreezing/slowfast/models/stem_helper.py:156-178
  bottleneck  altfreezing/slowfast/models/resnet_helper.py:255-326
  res block   altfreezing/slowfast/models/resnet_helper.py:411-444
  stages      altfreezing/slowfast/models/resnet_helper.py:616-647,
              altfreezing/slowfast/models/video_model_builder.py:423-578
  head        altfreezing/slowfast/models/head_helper.py:50-95
  callers     test/af_realtime.py:75-96 (normalise, sigmoid)
"""
import re
from collections import OrderedDict

import torch
import torch.nn.functional as F

BN_EPS = 1e-5


def _bn(x, sd, prefix):
    # nn.BatchNorm3d in eval mode: (x - mean) / sqrt(var + eps) * gamma + beta
    return F.batch_norm(x, sd[prefix + ".running_mean"].to(x.dtype), sd[prefix + ".running_var"].to(x.dtype),
                        sd[prefix + ".weight"].to(x.dtype), sd[prefix + ".bias"].to(x.dtype),
                        training=False, momentum=0.1, eps=BN_EPS)


def conv_bn_act(x, w, sd, bn_prefix, stride, pad, relu):
    y = F.conv3d(x, w.to(x.dtype), None, stride=stride, padding=pad)
    y = _bn(y, sd, bn_prefix)
    return F.relu(y) if relu else y


def stem(x, sd, p="resnet.s1.pathway0_stem"):
    w = sd[p + ".conv.weight"]
    kt, kh, kw = w.shape[2:]
    x = conv_bn_act(x, w, sd, p + ".bn", (1, 2, 2), (kt // 2, kh // 2, kw // 2), relu=True)
    return F.max_pool3d(x, kernel_size=(1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1))


def res_block(x, sd, p, stride):
    """relu( shortcut(x) + c_bn(c(relu(b_bn(b(relu(a_bn(a(x)))))))) ); the spatial stride
    sits on the 1x3x3 (STRIDE_1X1=False, i3d_ori.py:24) and on the projection shortcut."""
    wa = sd[p + ".branch2.a.weight"]
    tk = wa.shape[2]
    y = conv_bn_act(x, wa, sd, p + ".branch2.a_bn", (1, 1, 1), (tk // 2, 0, 0), relu=True)
    y = conv_bn_act(y, sd[p + ".branch2.b.weight"], sd, p + ".branch2.b_bn", (1, stride, stride), (0, 1, 1), relu=True)
    y = conv_bn_act(y, sd[p + ".branch2.c.weight"], sd, p + ".branch2.c_bn", (1, 1, 1), (0, 0, 0), relu=False)
    if (p + ".branch1.weight") in sd:
        sc = conv_bn_act(x, sd[p + ".branch1.weight"], sd, p + ".branch1_bn", (1, stride, stride), (0, 0, 0), relu=False)
    else:
        sc = x
    return F.relu(sc + y)


def _num_blocks(sd, stage):
    n = 0
    while ("resnet.s%d.pathway0_res%d.branch2.a.weight" % (stage, n)) in sd:
        n += 1
    return n


def res_stage(x, sd, stage):
    stage_stride = 1 if stage == 2 else 2          # RESNET.SPATIAL_STRIDES [[1],[2],[2],[2]], defaults.py:164
    for i in range(_num_blocks(sd, stage)):
        x = res_block(x, sd, "resnet.s%d.pathway0_res%d" % (stage, i), stage_stride if i == 0 else 1)
    return x


def head(x, sd, pool_size, p="resnet.head"):
    x = F.avg_pool3d(x, kernel_size=pool_size, stride=1)
    x = x.permute(0, 2, 3, 4, 1)                    # N,T,H,W,C ; dropout is identity in eval
    x = F.linear(x, sd[p + ".projection.weight"].to(x.dtype), sd[p + ".projection.bias"].to(x.dtype))
    return x.reshape(x.shape[0], -1)


def forward(sd, x, num_frames=32, crop=224, dtype=torch.float32, return_stages=False):
    """x: (B,3,T,H,W) normalised clip.  Returns (B,1) logits [and an OrderedDict of stage outputs]."""
    sd = {k: v for k, v in sd.items()}
    x = x.to(dtype)
    stages = OrderedDict()
    with torch.no_grad():
        x = stem(x, sd); stages["s1"] = x
        x = res_stage(x, sd, 2); stages["s2"] = x
        x = F.max_pool3d(x, kernel_size=(2, 1, 1), stride=(2, 1, 1)); stages["pool"] = x
        x = res_stage(x, sd, 3); stages["s3"] = x
        x = res_stage(x, sd, 4); stages["s4"] = x
        x = res_stage(x, sd, 5); stages["s5"] = x
        pool = (num_frames // 2, crop // 32, crop // 32)   # video_model_builder.py:548-556
        stages["avgpool"] = F.avg_pool3d(x, kernel_size=pool, stride=1)
        logits = head(x, sd, pool)
    if return_stages:
        return logits, stages
    return logits


# ---- FTCN-TT (altfreezing/model/classifier/i3d_temporal_var_fix_dropout_tt_cfg.py, time_transformer.py) ----------
def _conv_bn_pool_act(x, w, sd, bn_prefix, pad, pool2, relu):
    """conv (stride 1) -> BN -> [MaxPool3d((1,2,2)), where `temporal_only_conv` (:207-288) removed a stride 2 and wrapped
    the BN into nn.Sequential(bn, pool) - hence the `.0` in the key] -> [ReLU]."""
    y = F.conv3d(x, w.to(x.dtype), None, stride=1, padding=pad)
    y = _bn(y, sd, bn_prefix + (".0" if pool2 else ""))
    if pool2:
        y = F.max_pool3d(y, kernel_size=(1, 2, 2))
    return F.relu(y) if relu else y


def ftcn_stem(x, sd, p="resnet.s1.pathway0_stem"):
    w = sd[p + ".conv.weight"]                      # (64, 3, 5, 1, 1)
    x = _conv_bn_pool_act(x, w, sd, p + ".bn", (w.shape[2] // 2, 0, 0), True, True)
    return F.max_pool3d(x, kernel_size=(1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1))   # stem_helper.py:171,178


def ftcn_block(x, sd, p):
    pool2 = (p + ".branch2.b_bn.0.weight") in sd    # block 0 of s3 / s4
    wa = sd[p + ".branch2.a.weight"]
    y = _conv_bn_pool_act(x, wa, sd, p + ".branch2.a_bn", (wa.shape[2] // 2, 0, 0), False, True)
    y = _conv_bn_pool_act(y, sd[p + ".branch2.b.weight"], sd, p + ".branch2.b_bn", (0, 0, 0), pool2, True)
    y = _conv_bn_pool_act(y, sd[p + ".branch2.c.weight"], sd, p + ".branch2.c_bn", (0, 0, 0), False, False)
    if (p + ".branch1.weight") in sd:
        sc = _conv_bn_pool_act(x, sd[p + ".branch1.weight"], sd, p + ".branch1_bn", (0, 0, 0), pool2, False)
    else:
        sc = x
    return F.relu(sc + y)


def ftcn_stage(x, sd, stage):
    for i in range(_num_blocks(sd, stage)):
        x = ftcn_block(x, sd, "resnet.s%d.pathway0_res%d" % (stage, i))
    return x


def layer_norm(x, sd, p):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"].to(x.dtype), sd[p + ".bias"].to(x.dtype), 1e-5)


def time_transformer(tokens, sd, p="resnet.head.time_T", heads=16):
    """TimeTransformer.forward (time_transformer.py:268-281), depth 1: class token + position embedding, pre-norm
    attention (:36-71, scale dim_head^-0.5, no qkv bias) and pre-norm GELU MLP (:23-34), both residual (:8-13),
    then LayerNorm + Linear on the class token.  tokens: (B, N, D)."""
    b, n, d = tokens.shape
    x = torch.cat([sd[p + ".cls_token"].to(tokens.dtype).expand(b, 1, d), tokens], dim=1)
    x = x + sd[p + ".pos_embedding"].to(tokens.dtype)[:, :n + 1]
    l0, l1 = p + ".transformer.layers.0.0.fn", p + ".transformer.layers.0.1.fn"
    h = layer_norm(x, sd, l0 + ".norm")
    qkv = F.linear(h, sd[l0 + ".fn.to_qkv.weight"].to(h.dtype))
    q, k, v = [t.reshape(b, n + 1, heads, -1).transpose(1, 2) for t in qkv.chunk(3, dim=-1)]
    att = (q @ k.transpose(-1, -2) * (q.shape[-1] ** -0.5)).softmax(dim=-1)
    o = (att @ v).transpose(1, 2).reshape(b, n + 1, -1)
    x = x + F.linear(o, sd[l0 + ".fn.to_out.0.weight"].to(h.dtype), sd[l0 + ".fn.to_out.0.bias"].to(h.dtype))
    h = layer_norm(x, sd, l1 + ".norm")
    h = F.gelu(F.linear(h, sd[l1 + ".fn.net.0.weight"].to(h.dtype), sd[l1 + ".fn.net.0.bias"].to(h.dtype)))
    x = x + F.linear(h, sd[l1 + ".fn.net.3.weight"].to(h.dtype), sd[l1 + ".fn.net.3.bias"].to(h.dtype))
    c = layer_norm(x[:, 0], sd, p + ".mlp_head.0")
    return F.linear(c, sd[p + ".mlp_head.1.weight"].to(c.dtype), sd[p + ".mlp_head.1.bias"].to(c.dtype))


def ftcn_forward(sd, x, dtype=torch.float32, return_stages=False):
    """x: (B,3,T,H,W) normalised clip -> (B,1) logits of the FTCN-TT plugin (stop_point 5, patch_type 'time')."""
    x = x.to(dtype)
    stages = OrderedDict()
    with torch.no_grad():
        x = ftcn_stem(x, sd); stages["s1"] = x
        x = ftcn_stage(x, sd, 2); stages["s2"] = x
        x = F.max_pool3d(x, kernel_size=(2, 1, 1), stride=(2, 1, 1))
        x = ftcn_stage(x, sd, 3); stages["s3"] = x
        x = ftcn_stage(x, sd, 4); stages["s4"] = x
        tok = F.avg_pool3d(x, kernel_size=(1, x.shape[3], x.shape[4]))          # TransformerHead 'time' (:133-135)
        tok = tok.reshape(x.shape[0], x.shape[1], x.shape[2]).permute(0, 2, 1)  # (B, T/2, C)   (:186-188)
        stages["tokens"] = tok
        logits = time_transformer(tok, sd)
    if return_stages:
        return logits, stages
    return logits


# ---- two-pathway SlowFast (altfreezing/slowfast/models/video_model_builder.py:86-143, 146-387) ----------------

def _stage_pathway(x, sd, stage, pathway):
    stage_stride = 1 if stage == 2 else 2
    i = 0
    while ("resnet.s%d.pathway%d_res%d.branch2.a.weight" % (stage, pathway, i)) in sd:
        x = res_block(x, sd, "resnet.s%d.pathway%d_res%d" % (stage, pathway, i), stage_stride if i == 0 else 1)
        i += 1
    return x


def fuse_fast_to_slow(xs, xf, sd, p, alpha):
    """FuseFastToSlow.forward: conv_f2s [k,1,1] stride [alpha,1,1] pad [k//2,0,0] + BN + ReLU on the Fast tensor,
    concatenated to the Slow one by channel."""
    w = sd[p + ".conv_f2s.weight"]
    k = w.shape[2]
    f = conv_bn_act(xf, w, sd, p + ".bn", (alpha, 1, 1), (k // 2, 0, 0), relu=True)
    return torch.cat([xs, f], 1)


def slowfast_forward(sd, x_slow, x_fast, alpha=8, dtype=torch.float32, return_stages=False):
    """SlowFast.forward([x_slow, x_fast]) -> (B, num_classes * positions) logits."""
    sd = {k: v for k, v in sd.items()}
    xs, xf = x_slow.to(dtype), x_fast.to(dtype)
    stages = OrderedDict()
    with torch.no_grad():
        xs = stem(xs, sd, "resnet.s1.pathway0_stem")
        xf = stem(xf, sd, "resnet.s1.pathway1_stem")
        xs = fuse_fast_to_slow(xs, xf, sd, "resnet.s1_fuse", alpha)
        stages["s1"] = (xs, xf)
        for st in (2, 3, 4, 5):
            xs = _stage_pathway(xs, sd, st, 0)
            xf = _stage_pathway(xf, sd, st, 1)
            if st < 5:
                xs = fuse_fast_to_slow(xs, xf, sd, "resnet.s%d_fuse" % st, alpha)
            stages["s%d" % st] = (xs, xf)                          # pathway pools are [1,1,1]: identity
        # head pools: [T/alpha, crop/32, crop/32] and [T, crop/32, crop/32] (video_model_builder.py:349-365)
        t_fast = x_fast.shape[2]
        crop32 = x_fast.shape[3] // 32
        ps = F.avg_pool3d(xs, kernel_size=(t_fast // alpha, crop32, crop32), stride=1)
        pf = F.avg_pool3d(xf, kernel_size=(t_fast, crop32, crop32), stride=1)
        z = torch.cat([ps, pf], 1).permute(0, 2, 3, 4, 1)
        z = F.linear(z, sd["resnet.head.projection.weight"].to(z.dtype), sd["resnet.head.projection.bias"].to(z.dtype))
        logits = z.reshape(z.shape[0], -1)
    if return_stages:
        return logits, stages
    return logits


def normalize(clips_bthwc):
    """callers' pre-processing (af_realtime.py:77-83): (B,T,H,W,C) 0..255 -> (B,C,T,H,W) fp32."""
    x = torch.as_tensor(clips_bthwc, dtype=torch.float32).permute(0, 4, 1, 2, 3)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1, 1) * 255
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1, 1) * 255
    return x.sub(mean).div(std)


def scores(logits):
    """callers' post-processing (af_realtime.py:92-95)."""
    if logits.ndim == 1:
        logits = logits.unsqueeze(1)
    if logits.size(1) == 1:
        return torch.sigmoid(logits).squeeze(1).float()
    return torch.softmax(logits, dim=1)[:, 1].float()


def strip_checkpoint(saved):
    """the unwrap + one-prefix strip of ModelBase.load (altfreezing/model/_base.py:58-73)."""
    if isinstance(saved, dict):
        for k in ("state_dict", "classifier_state_dict", "model_state_dict"):
            if k in saved:
                saved = saved[k]
                break
    out = OrderedDict()
    for k, v in saved.items():
        for p in ("module.", "network.", "_warped_network."):
            if k.startswith(p):
                k = k[len(p):]
                break
        out[k] = v
    return out
