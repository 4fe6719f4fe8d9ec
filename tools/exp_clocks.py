"""Run ON THE GPU BOX: sclk / power (rocm-smi) while the 256x256 tile loops on the full chip and on half of it."""
import os, sys, subprocess, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from exp_conv111 import layer   # noqa

def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=20).stdout
        return out.strip().replace("\n", " | ")[:600]
    except Exception as e:
        return "rocm-smi failed: %r" % (e,)

print("idle:", smi(), flush=True)
for t, label in ((32, "half chip (256 WGs of 128x256 = 128 CUs busy? no: 256 WGs)"), (16, "128 WGs"), (64, "full chip 256x256")):
    run, name = layer(1, t, 32, 32, 4608, 256, res=False)
    stop = [False]
    def spin():
        while not stop[0]:
            for _ in range(200): run()
            torch.cuda.synchronize()
    th = threading.Thread(target=spin); th.start()
    time.sleep(1.5)
    print(label, name, ":", smi(), flush=True)
    stop[0] = True; th.join()
