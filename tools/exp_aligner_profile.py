"""Run ON THE GPU BOX: cProfile of FasterCropAlignXRay.__call__(device_output=True), 200 calls back to back."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from af_mi355x import aligner
dev = torch.device("cuda:0")
al = aligner.FasterCropAlignXRay(224, device=dev)
clips = [aligner.synthetic_clip(32, seed=2026 + i) for i in range(4)]
for i in range(8): al(*clips[i % 4], device_output=True)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(200): al(*clips[i % 4], device_output=True)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
