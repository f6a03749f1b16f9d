#!/usr/bin/env python3
"""Sustained rate and host memory of the bench over long timed regions (one child process each; RSS = the child's peak):
    python tools/soak.py [steps ...]        FRP_NO_GRAPH=1 python tools/soak.py ..."""
import json
import os
import subprocess
import sys

steps = [int(a) for a in sys.argv[1:]] or [300, 3000]
for n in steps:
    code = ("import resource, runpy, sys; sys.argv = ['bench.py', '--steps', '%d', '--cpu-frames', '0', '--pcie-steps', '0', '--threshold-steps', '0'];"
            "\ntry:\n    runpy.run_path('bench.py', run_name='__main__')\nexcept SystemExit:\n    pass\n"
            "print('RSS_MIB', resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024, file=sys.stderr)") % n
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    d = json.loads([l for l in r.stdout.strip().split("\n") if l.startswith("{")][-1])
    rss = [l for l in r.stderr.split("\n") if l.startswith("RSS_MIB")]
    print(f"FRP_NO_GRAPH={os.environ.get('FRP_NO_GRAPH', '-')} {n} steps: {d['value']} faces/s, {d['ms_per_step']} ms/step, frac {d['roofline']['frac']}; peak RSS {rss[-1].split()[1] if rss else '?'} MiB", flush=True)
