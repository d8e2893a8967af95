"""Prints the headline numbers of a bench.py JSON line (argument: a log file that contains the line)."""
import json, sys
line = [x for x in open(sys.argv[1]) if x.startswith("{")][-1]
d = json.loads(line)
r = d["roofline"]
print(f"{d['value']:.0f} {d['unit']}  {d['ms_per_step']:.2f} ms/step  model {d.get('model_tflops_per_gpu')} TF/s  "
      f"roofline {r['achieved']} {r['unit']} frac {r['frac']} avg launch {r.get('avg_launch_ms')} ms")
if "decode" in d:
    print("decode", d["decode"]["ms_per_step"], "ms/step hbm_frac", d["decode"]["hbm_frac"],
          "| long context", d["decode_long_context"]["ms_per_step"], d["decode_long_context"]["hbm_frac"])
