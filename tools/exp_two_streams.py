#!/usr/bin/env python3
"""EXPERIMENT (GPU): does running two half-batches on two streams beat one full batch?  MFMA-bound and HBM-bound launches
alternate in the forward; kernels of two independent half-batches could fill each other's idle resource / tile tails.
    python tools/exp_two_streams.py [--batch 16] [--dtype bf16]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import af_mi355x  # noqa
from af_mi355x import synth
from af_mi355x.classifier import Classifier

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--streams", type=int, default=2)
args = ap.parse_args()
dev = torch.device("cuda", 0)
sd = synth.synthetic_state_dict(seed=0)
def mk():
    c = Classifier(precision=args.dtype); c.network.load_state_dict(sd); return c.to(dev).eval()
B = args.batch
u8 = synth.synthetic_clips_u8(B, seed=2026, kind="uniform").to(dev)
full = mk()
NS = args.streams
halves = [mk() for _ in range(NS)]
streams = [torch.cuda.Stream(dev) for _ in range(NS)]
parts = [p.contiguous() for p in u8.chunk(NS)]

def run_full():
    return full.network.forward_clips_u8(u8)["final_output"]

def run_split():
    outs = []
    cur = torch.cuda.current_stream(dev)
    for c, s, p in zip(halves, streams, parts):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            outs.append(c.network.forward_clips_u8(p)["final_output"])
    for s in streams:
        cur.wait_stream(s)
    return torch.cat(outs)

with torch.inference_mode():
    for fn, name in ((run_full, "one stream, B=%d" % B), (run_split, "%d streams x B=%d" % (NS, B // NS)), (run_full, "one stream again")):
        for _ in range(5):
            y = fn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            y = fn()
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / args.steps
        print("%-28s %.3f ms/step  %.0f clips/s" % (name, 1e3 * dt, B / dt), flush=True)
    a, b = run_full(), run_split()
    torch.cuda.synchronize(dev)
    print("max |d| full vs split", float((a - b).abs().max()))
