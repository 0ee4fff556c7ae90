"""ms per step, per-kernel table and the pipelined-detect sweep of bench.py lines.   python3 tools/sweep_summary.py <line.json> [...]"""
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable:", e)
        continue
    print("==", f, d.get("det_options"), "ms_per_step", d["ms_per_step"], "images/s", d["value"], "| bf16", (d.get("bf16") or {}).get("ms_per_step"), (d.get("bf16") or {}).get("images_per_s"))
    for k, v in (d.get("roofline") or {}).get("all_kernels", {}).items():
        print(f'   {v["ms_per_step"]:8.4f} {v["launches_per_step"]:3d} {str(v.get("frac")):8s} {k}')
    sw = d.get("post_threads_sweep") or {}
    for k in ("1", "2", "4", "16", "16_device_chain"):
        if k in sw:
            r = sw[k]
            print("   sweep", k, {kk: vv["detect_postprocess_pipelined_images_per_s"] for kk, vv in r.items() if isinstance(vv, dict) and "detect_postprocess_pipelined_images_per_s" in vv},
                  "chain images", (r.get("where") or {}).get("images_device_chain"))
    print("   ", {k: d.get(k) for k in ("detect_postprocess_pipelined_images_per_s", "e2e_pages_per_s", "postprocess_images_per_s", "postprocess_dense_images_per_s")},
          "bf16 e2e", (d.get("bf16") or {}).get("e2e_pages_per_s"))
