#!/usr/bin/env python3
"""One-line-per-basic-block picture of how hipcc scheduled a kernel's instruction stream (no GPU needed):

    python tools/isa_schedule.py <package>/csrc/af_stem3.hip [substring of the mangled kernel name]

M = MFMA, r / w = LDS read / write, G = global or buffer load, D = LDS-DMA (buffer_load ... lds), S = store, # = scratch,
[V(n)] / [L(n)] = s_waitcnt vmcnt(n) / lgkmcnt(n), |B| = s_barrier, v / s = other vector / scalar instructions.
What to look for: `r[L(0)]M` chains (an LDS read issued right in front of the MFMA that needs it - the wave stalls for the
LDS latency every time; reads belong a whole MFMA group earlier), `[V(0)]` inside a loop that also prefetches (the wait drains
the prefetch), `#` in a loop with counted vmcnt waits.  Round 3's stem / conv311 / conv_ca / stem_rows rewrites started here."""
import re
import subprocess
import sys

src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
out = "/tmp/" + src.split("/")[-1] + ".s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--offload-device-only", "-S", "-o", out, src],
               stderr=subprocess.DEVNULL, check=True)
asm = open(out).read()
for m in re.finditer(r"^(_ZN2af\w+):\s*;?.*?$(.*?)\.Lfunc_end", asm, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if pat and pat not in name:
        continue
    seq = []
    for line in body.split("\n"):
        line = line.strip()
        if not line or line.startswith(";") or line.startswith("."):
            if line.startswith(".LBB"):
                seq.append("\n" + line.split()[0])
            continue
        op = line.split()[0]
        if op.startswith("v_mfma"): seq.append("M")
        elif op.startswith("ds_read"): seq.append("r")
        elif op.startswith("ds_write"): seq.append("w")
        elif op.startswith("global_load") or op.startswith("buffer_load"): seq.append("D" if " lds" in line else "G")
        elif op.startswith("global_store") or op.startswith("buffer_store"): seq.append("S")
        elif op.startswith("scratch"): seq.append("#")
        elif op.startswith("s_waitcnt"): seq.append("[" + line.split(None, 1)[1].replace(" ", "").replace("lgkmcnt", "L").replace("vmcnt", "V") + "]")
        elif op.startswith("s_barrier"): seq.append("|B|")
        elif op.startswith("s_cbranch") or op.startswith("s_branch"): seq.append("<br>")
        elif op.startswith("v_"): seq.append("v")
        elif op.startswith("s_"): seq.append("s")
        else: seq.append("?")
    print("=====", name)
    print("".join(seq))
for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size", asm, re.S):
    blk = m.group(0)
    nm = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if pat and pat not in nm:
        continue
    print("%-100s vgpr %s spilled %s sgpr %s" % (nm[:100], re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1),
                                              re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1), re.search(r"\.sgpr_count:\s+(\d+)", blk).group(1)))
