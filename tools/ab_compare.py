#!/usr/bin/env python3
"""Two output lines of bench/pcl_baseline.cpp side by side:  python tools/ab_compare.py a.json b.json [tolerance]
Exit code 0 if |T_a - T_b|_F <= tolerance (default 1e-4, BASELINE.json's bound on the final 4x4)."""
import json
import sys

import numpy as np


def main():
    a, b = (json.loads(open(p).read().strip().splitlines()[-1]) for p in sys.argv[1:3])
    tol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-4
    d = float(np.linalg.norm(np.asarray(a["T"], np.float64) - np.asarray(b["T"], np.float64)))
    for r in (a, b):
        print(f"{r['impl']:>14}  mode {r['mode']}  {r['n_source']} x {r['n_target']} points  {r['iterations']} iterations  "
              f"{r['ms_per_iteration']:.4f} ms/iteration  fitness {r['fitness']:.6g}")
    print(f"|T_a - T_b|_F = {d:.3e}  (tolerance {tol:g})   speed ratio b/a = {a['ms_per_iteration'] / b['ms_per_iteration']:.1f}x")
    sys.exit(0 if d <= tol else 1)


if __name__ == "__main__":
    main()
