#!/usr/bin/env python3
"""Markdown table of the bench lines kept under profiles/ (one JSON line per file): python tools/measured_table.py r02_h_"""
import glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prefix = sys.argv[1] if len(sys.argv) > 1 else "r02_h_"
rows = []
for f in sorted(glob.glob(os.path.join(ROOT, "profiles", prefix + "*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception:
        continue
    r = d.get("roofline") or {}
    cb = d.get("cpu_baseline") or {}
    extra = []
    if "phases_ms" in d:
        extra.append(" / ".join("%s %.1f" % (k.replace("_ms", ""), v) for k, v in d["phases_ms"].items()))
    if "msm_ms" in d:
        extra.append("MSM %.2f ms (%.0f Mop/s), NTT %.2f ms" % (d["msm_ms"], d.get("msm_mops", 0), d.get("ntt_ms", 0)))
    rows.append((os.path.basename(f), d["ms_per_step"], d["value"], d["unit"], r.get("kernel", ""), r.get("frac"), r.get("avg_launch_us"),
                 cb.get("value"), "; ".join(extra)))
print("| file | ms/step | value | dominant kernel (avg launch, HBM-roofline frac) | CPU baseline | detail |")
print("|---|---|---|---|---|---|")
for name, ms, val, unit, kern, frac, us, cpu, extra in rows:
    k = "%s (%.0f us, %.4f)" % (kern.replace("_kernel", ""), us or 0, frac or 0) if kern else ""
    print("| `%s` | %.2f | %.3g %s | %s | %s | %s |" % (name, ms, val, unit, k, ("%.3g" % cpu) if cpu else "", extra))
