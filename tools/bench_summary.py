"""Print the headline fields of a bench.py JSON line (file argument)."""
import json, sys
d = json.load(open(sys.argv[1]))
print("value", d["value"], "mrays_isolated", d.get("mrays_isolated"), "kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"])
f = d.get("four_k")
if f:
    print({k: f.get(k) for k in ("ms_per_frame", "frames_per_s", "effective_tflops", "psnr_vs_oracle_db", "max_abs_vs_oracle", "rank_share_8gpu")},
          "sr frac", f["sr_roofline"]["frac"])
    print({k: f.get(k) for k in ("ms_per_frame_median", "ms_per_frame_p90", "frames_timed")})
    for n, v in (f.get("rank_share_projection") or {}).items():
        if isinstance(v, dict):
            print("  N =", n, "best", v["best"], "| all:", [(c["tile_size"], c["share_ms_median"]) for c in v["candidates"]])
    for k in ("four_k_f16x3", "four_k_bf16x6", "four_k_bf16x3", "four_k_fp32mfma"):
        if d.get(k):
            print(" ", k, {q: d[k].get(q) for q in ("ms_per_frame", "ms_per_frame_median") + tuple(x for x in d[k] if x.startswith("psnr_vs"))})
    if "cpu_baseline" in f:
        print("4k cpu", f["cpu_baseline"]["value"], f["cpu_baseline"]["unit"])
print(d.get("parity_vs_oracle"))
if "cpu_baseline" in d:
    print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"].get("threads_used"))
print("json chars", len(json.dumps(d)))
j = d.get("joint_train_step") or {}
print("joint", {k: j.get(k) for k in ("ms_per_iteration", "ms_per_iteration_after_tv_before", "iterations_per_s", "error")})
