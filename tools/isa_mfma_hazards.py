#!/usr/bin/env python3
"""Lint for asm-statement MFMAs (no GPU needed):  python tools/isa_mfma_hazards.py <package>/csrc/af_conv.hip [name substring] [--raw]

hipcc pads the hazards of its OWN MFMAs; an MFMA inside an asm statement gets none (cdna_hip_programming.md 5.7).  This walks the
generated ISA and reports, per kernel, every v_mfma whose operand registers are written by a vector-ALU instruction within the
WINDOW instructions in front of it (read-after-write without wait states) or whose operand / destination registers are written by
one within WINDOW instructions behind it (write-after-read / write-after-write while the MFMA is in flight).  LDS reads and
waits are not vector-ALU writes and are governed by the kernels' own s_waitcnt.  Exit code 1 if anything is found."""
import re
import subprocess
import sys

WINDOW = 2
args = [a for a in sys.argv[1:] if not a.startswith("--")]
src = args[0]
pat = args[1] if len(args) > 1 else ""
out = "/tmp/lint_" + src.split("/")[-1] + ".s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--offload-device-only", "-S", "-o", out, src],
               stderr=subprocess.DEVNULL, check=True)


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


bad = 0
for m in re.finditer(r"^(_ZN2af\w+):\s*;?.*?$(.*?)\.Lfunc_end", open(out).read(), re.S | re.M):
    name, body = m.group(1), m.group(2)
    if pat and pat not in name:
        continue
    ins = []
    for line in body.split("\n"):
        line = line.split(";")[0].strip()
        if not line or line.startswith(".") or line.endswith(":"):
            continue
        parts = line.replace(",", " ").split()
        ins.append((parts[0], parts[1:], line))
    found = []
    for i, (op, ops, line) in enumerate(ins):
        if not op.startswith("v_mfma"):
            continue
        dst, srcs = regs(ops[0]), set().union(*[regs(t) for t in ops[1:4]])
        for k in range(max(0, i - WINDOW), i):
            o2, p2, l2 = ins[k]
            if o2.startswith("v_") and not o2.startswith("v_mfma") and not o2.startswith("v_cmp") and p2 and regs(p2[0]) & srcs:
                found.append("RAW  %-60s -> %s" % (l2[:60], line[:70]))
        for k in range(i + 1, min(len(ins), i + 1 + WINDOW)):
            o2, p2, l2 = ins[k]
            if o2.startswith("v_") and not o2.startswith("v_mfma") and not o2.startswith("v_cmp") and p2 and regs(p2[0]) & (srcs | dst):
                found.append("WAR  %-60s <- %s" % (line[:60], l2[:70]))
    n = sum(1 for op, _, _ in ins if op.startswith("v_mfma"))
    if n:
        found = [x for x in found if x.startswith("RAW")] if "--raw" in sys.argv else found
        print("%-100s %4d MFMAs, %d findings" % (name[:100], n, len(found)))
        for f in [x for x in found if x.startswith("RAW")][:6]:
            print("    " + f)
        bad += len(found)
sys.exit(1 if bad else 0)
