"""Prints the fields of a bench.py JSON line that a GPU session is usually run for."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.1f tiles/s  h2h %s  ms/step %.2f  instances %s (%s)" % (d["value"], d.get("value_host_to_host"), d["ms_per_step"],
                                                                      d["config"].get("instances_last_step"), d["config"].get("instances_from")))
print("checkpoint", d["config"].get("checkpoint"))
print("stage_ms", d["config"].get("stage_ms"))
r = d.get("roofline")
if r:
    print("roofline frac %.4f conv_ms %.2f achieved %.1f traffic %s | %s" % (r["frac"], r["conv_ms_per_step"], r["achieved"], r.get("traffic_hbm_bytes_per_step"),
                                                                            (r.get("traffic_unit") or "")[:90]))
for k, v in d.get("variants", {}).items():
    print("variant %-24s %.1f tiles/s  ms/step %s" % (k, v["value"], v.get("ms_per_step")))
c = d.get("variants", {}).get("cfg3_fast_b64_bf16")
if c:
    print("cfg3: network_ms %.2f step/network %.3f roofline frac %.4f conv_ms %.2f instances %s net-output %s | %s" % (
        c.get("network_ms", 0), c.get("step_over_network", 0), c["roofline"]["frac"], c["roofline"]["conv_ms_per_step"], c["instances_last_step"],
        c.get("instances_in_network_output"), c.get("checkpoint")))
print("cpu_baseline", d.get("cpu_baseline"))
if "per_rank" in d["config"]:
    print("per_rank", d["config"]["per_rank"])
