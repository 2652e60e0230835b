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
    if "value" in v:
        print("variant %-24s %.1f tiles/s  ms/step %s" % (k, v["value"], v.get("ms_per_step")))
t = d.get("variants", {}).get("train_step")
if t:
    for ph in ("phase0", "phase1"):
        r = t[ph]
        print("train %s: %.2f ms/step (fwd %.2f, loss+bwd %.2f, opt %.2f) batch %d, timed CONV launches %.2f ms, slab %.0f MB" % (
            ph, r["ms_per_step"], r["forward_ms"], r["loss_backward_ms"], r["optimizer_ms"], r["batch"], r["timed_conv_launch_ms"], r["gradient_slab_mb"]))
w = d.get("variants", {}).get("wsi_8k")
if w:
    print("wsi %s: stage 1 %.2f s = %.0f patches/s, stage 2 %.2f s, %d instances" % (w["slide"], w["stage1_s"], w["patches_per_s"], w["stage2_s"], w["instances"]))
if "flood_whole_tile_replays" in d:
    print("flood_whole_tile_replays", d["flood_whole_tile_replays"], d["config"].get("flood"))
c = d.get("variants", {}).get("cfg3_fast_b64_bf16")
if c:
    print("cfg3: network_ms %.2f step/network %.3f roofline frac %.4f conv_ms %.2f instances %s net-output %s | %s" % (
        c.get("network_ms", 0), c.get("step_over_network", 0), c["roofline"]["frac"], c["roofline"]["conv_ms_per_step"], c["instances_last_step"],
        c.get("instances_in_network_output"), c.get("checkpoint")))
print("cpu_baseline", d.get("cpu_baseline"))
if "per_rank" in d["config"]:
    print("per_rank", d["config"]["per_rank"])
