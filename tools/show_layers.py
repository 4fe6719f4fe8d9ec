"""print a --layers-json table sorted by the gap to the per-launch speed of light:  python tools/show_layers.py file.json [n]"""
import json, sys
rows = json.load(open(sys.argv[1])); n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
tot = sum(r["ms"] for r in rows)
print("total %.3f ms, %d launches, speed of light %.3f ms" % (tot, len(rows), sum(r.get("sol_ms", 0) for r in rows)))
for r in sorted(rows, key=lambda r: -(r["ms"] - r.get("sol_ms", 0)))[:n]:
    print("%3d %-44s %-30s ms %.4f sol %.4f  TF %6.0f GB/s %5.0f  M=%s N=%s K=%s" % (
        r["i"], r["name"][-44:], r.get("kernel", "")[:30], r["ms"], r.get("sol_ms", 0), r.get("tflops", 0) or 0, r.get("alg_GBs", 0) or 0,
        r.get("M"), r.get("N"), r.get("K")))
