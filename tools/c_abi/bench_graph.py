#!/usr/bin/env python3
"""Frame time of render_c (the C program on the C ABI) with its seven launches issued directly and replayed as a
captured HIP graph, for three frame sizes.  Usage (GPU box, from the repository root): python3 tools/c_abi/bench_graph.py"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from nerf_loc_amd import synth
    from test_gpu_c_abi import BINARY, _write_input
    sds = [synth.make_state_dict(0, "B"), synth.make_state_dict(1, "B")]
    with tempfile.TemporaryDirectory() as tmp:
        for res, reps in ((48, 200), (208, 50), (800, 5)):
            H, W, K, c2w, near, far = synth.lego_camera(res, theta=30.0)
            inp, out = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
            for precision in (1, 0):
                if precision == 0 and res == 800:
                    continue            # 1.4 s per frame: nothing to learn about launch overhead
                _write_input(inp, H, W, K, c2w, near, far, precision, sds)
                p = subprocess.run([BINARY, inp, out, "graph", str(reps)], capture_output=True, text=True, timeout=300)
                if p.returncode != 0:
                    raise SystemExit("render_c failed: " + p.stderr[-500:])
                line = json.loads(p.stdout.strip().splitlines()[-1])
                line["res"] = res
                line["graph_saves"] = 1.0 - line["ms_graph"] / line["ms_direct"]
                print(json.dumps(line))
                sys.stdout.flush()


if __name__ == "__main__":
    main()
