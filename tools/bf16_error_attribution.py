#!/usr/bin/env python3
"""TEST / ANALYSIS TOOL (CPU, uses the oracle): where does the bf16 logit error come from?

Emulates the 16-bit engine's roundings inside the fp32 oracle forward, one source at a time:
  W   conv weights rounded to the 16-bit type (fp32 accumulate, activations exact)
  A   activations rounded at every tensor the engine stores (conv+BN(+ReLU) outputs, block outputs, the input), weights exact
  A(stage) only the activations of one stage rounded
  WA  both (what the engine does)
and prints max |dlogit| vs the exact fp32 forward on the golden clips.  Result recorded in DESIGN.md section 5."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import af_mi355x  # noqa: E402,F401
from af_mi355x import synth  # noqa: E402
import i3d_oracle as oracle  # noqa: E402


def run(sd, x, dt, round_w, round_a_stages):
    rd = (lambda t: t.to(dt).float())
    sdw = {k: (rd(v) if (round_w and v.dim() == 5) else v) for k, v in sd.items()}
    state = {"stage": "s1"}
    orig = oracle.conv_bn_act

    def cba(xx, w, sd_, bn, stride, pad, relu):
        y = orig(xx, w, sd_, bn, stride, pad, relu)
        st = bn.split(".")[1]
        return rd(y) if st in round_a_stages else y

    orig_block = oracle.res_block

    def block(xx, sd_, p, stride):
        y = orig_block(xx, sd_, p, stride)
        return rd(y) if p.split(".")[1] in round_a_stages else y

    oracle.conv_bn_act, oracle.res_block = cba, block
    try:
        xin = rd(x) if "s1" in round_a_stages else x
        return oracle.forward(sdw, xin)
    finally:
        oracle.conv_bn_act, oracle.res_block = orig, orig_block


def main():
    torch.set_num_threads(8)
    recipe = sys.argv[1] if len(sys.argv) > 1 else "mild"
    sd = synth.synthetic_state_dict(seed=0 if recipe == "mild" else 3, recipe=recipe)
    u8 = torch.cat([synth.synthetic_clips_u8(2, seed=2026, kind="uniform"), synth.synthetic_clips_u8(1, seed=2026, kind="smooth")])
    x = synth.normalize_like_callers(u8)
    exact = oracle.forward(sd, x)
    print("exact logits", exact.flatten().tolist())
    allst = ("s1", "s2", "s3", "s4", "s5")
    for name, dt in (("bf16", torch.bfloat16), ("f16", torch.float16)):
        rows = [("W", True, ()), ("A", False, allst), ("WA", True, allst)] + \
               ([("A(%s)" % s, False, (s,)) for s in allst] if name == "bf16" else [])
        for tag, rw, ra in rows:
            y = run(sd, x, dt, rw, ra)
            print("%-5s %-6s max|d| %.3e   d = %s" % (name, tag, (y - exact).abs().max().item(),
                                                      ["%.2e" % v for v in (y - exact).flatten().tolist()]))
    # the weight rounding alone, one group of conv weights at a time (bf16)
    rd = lambda t: t.to(torch.bfloat16).float()
    groups = [("W(%s)" % st, (lambda k, st=st: k.split(".")[1] == st)) for st in allst] + \
             [("W(%s)" % cls, (lambda k, cls=cls: k.endswith(cls) and ".s1." not in k))
              for cls in (".a.weight", ".b.weight", ".c.weight", "branch1.weight")]
    for tag, pred in groups:
        sdw = {k: (rd(v) if (v.dim() == 5 and pred(k)) else v) for k, v in sd.items()}
        y = oracle.forward(sdw, x)
        n = sum(1 for k, v in sd.items() if v.dim() == 5 and pred(k))
        print("bf16  %-18s %2d layers  d = %s" % (tag, n, ["%+.2e" % v for v in (y - exact).flatten().tolist()]))


if __name__ == "__main__":
    main()
