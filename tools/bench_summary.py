"""Print the headline fields of a bench.py JSON line (file argument)."""
import json, sys
d = json.load(open(sys.argv[1]))
print("value", d["value"], "mrays_isolated", d.get("mrays_isolated"), "kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"])
f = d.get("four_k")
if f:
    print({k: f.get(k) for k in ("ms_per_frame", "frames_per_s", "effective_tflops", "psnr_vs_oracle_db", "max_abs_vs_oracle", "rank_share_8gpu")},
          "sr frac", f["sr_roofline"]["frac"])
    if "cpu_baseline" in f:
        print("4k cpu", f["cpu_baseline"]["value"], f["cpu_baseline"]["unit"])
print(d.get("parity_vs_oracle"))
if "cpu_baseline" in d:
    print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"].get("threads_used"))
print("json chars", len(json.dumps(d)))
