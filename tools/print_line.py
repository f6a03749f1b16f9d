#!/usr/bin/env python3
"""One line per bench.py output file: value, ms per step, roofline.frac, host_to_host_lanes.   python tools/print_line.py <file.json>"""
import json,sys
d=json.loads(open(sys.argv[1]).readline()); r=d["roofline"]
print(sys.argv[1].split('/')[-1], d["value"], d["ms_per_step"], r["frac"], d["config"]["host_to_host_lanes"]["faces_per_s"] if d["config"].get("host_to_host_lanes") else None)
