#!/usr/bin/env python3
"""Rows of DESIGN.md section 3.1 from profiles/<tag>_layers_f32.txt and profiles/pmc.json: per launch label the launches of a step, their
summed time, executed MFMA FLOPs / time / the peak of the instruction issued, the MFMA-busy counter, HBM bytes per launch.
usage: tools/design_table.py [tag, default r06]"""
import collections, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc.json")))["f32"]["kernels"]
t = collections.OrderedDict()
for ln in open(os.path.join(ROOT, "profiles", f"{tag}_layers_f32.txt")):
    m = re.match(r"\s*\d+\s+(\S+)\s+([0-9.]+)\s", ln)
    if m:
        e = t.setdefault(m.group(1), [0, 0.0])
        e[0] += 1
        e[1] += float(m.group(2))
tot = sum(v[1] for v in t.values())
wsum = 0.0
for name, (n, ms) in sorted(t.items(), key=lambda kv: -kv[1][1]):
    k = pmc.get(name)
    if not k:
        print(f"| `{name}` | {n} | {ms:.3f} | - | - |")
        continue
    fl = k["mfma_flops"] * n   # (the extract holds per-launch averages)
    peak = 157.3e12 if k["mfma_flops_f32"] > 0 else 2.5e15
    frac = fl / (ms * 1e-3) / peak if fl else 0
    wsum += frac * ms
    print(f"| `{name}` | {n} | {ms:.3f} | {frac:.2f} of {'157 TF f32' if peak < 1e15 else '2.5 PF bf16'}, MFMA busy {k['mfma_busy']:.2f} | {k['fetch_bytes'] / 1e6:.0f} + {k['write_bytes'] / 1e6:.0f} = {k['hbm_bytes'] / 1e6:.0f} MB |")
print(f"| sum | {sum(v[0] for v in t.values())} | {tot:.2f} | time-weighted {wsum / tot:.2f} | |")
