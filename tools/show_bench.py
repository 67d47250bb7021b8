"""Short view of a bench.py JSON line: python tools/show_bench.py <file>"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.1f %s  ms_per_step %.4f  n_gpus %d" % (d["value"], d["unit"], d["ms_per_step"], d["n_gpus"]))
r = d["roofline"]
print("roofline: %s  %.1f GB/s  frac %.3f  avg_launch_ms %.5f  traffic %s" % (r["kernel"], r["achieved"], r["frac"], r["avg_launch_ms"], r["traffic"]))
for p in d.get("passes", []):
    print("   pass %-50s %.5f ms  frac %.3f" % (p["kernel"][:50], p["avg_launch_ms"], p["frac"]))
print("phases", d.get("phases_ms"), "id_match", d["id_match"]["equal"], "pipelined", (d.get("pipelined") or {}).get("value"))
if "cpu_baseline" in d: print("cpu_baseline", d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"])
if "distributed" in d: print("distributed", d["distributed"])
