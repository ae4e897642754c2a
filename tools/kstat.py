#!/usr/bin/env python3
"""prints the headline numbers of a bench.py JSON line read from stdin (helper for experiments)"""
import json
import sys
d = json.loads(sys.stdin.read())
k = d["kernel_avg_launch_us"]
tag = sys.argv[1] if len(sys.argv) > 1 else ""
print(tag, "value", d["value"], "per_frame", d["per_frame_api"]["value"], "match1", k.get("k_match<16>:pass1"), "match2",
      k.get("k_match<16>:pass2"), "refine", k.get("k_refine"), "nms", k.get("k_nms"), "emit", k.get("k_emit"), "seq", d["sequence_timings_us"])
