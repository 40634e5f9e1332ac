#!/usr/bin/env python
"""Run golden command lines many times in fresh processes and count the distinct outputs (a race shows as more than one).
    python tools/flaky_probe.py N case [case ...]    (environment is passed through, e.g. PG_GPU_TOKENIZER=0)"""
import hashlib
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from cases import CASES  # noqa: E402

n = int(sys.argv[1])
gold = os.path.join(ROOT, "tests", "golden")
for name in sys.argv[2:]:
    case = [c for c in CASES if c["name"] == name][0]
    seen = {}
    want = open(os.path.join(gold, name + ".out")).read()
    for k in range(n):
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "o")
            argv = [a.format(geno=os.path.join(gold, case["fixture"] + ".geno.gz"), dir=gold, out=out) for a in case["argv"]] + ["-o", out]
            r = subprocess.run([sys.executable, os.path.join(ROOT, case["tool"])] + argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            txt = open(out).read() if os.path.exists(out) else "<no output> rc=%d %s" % (r.returncode, r.stderr.decode()[-300:])
            h = hashlib.md5(txt.encode()).hexdigest()
            if h not in seen:
                seen[h] = [0, txt]
            seen[h][0] += 1
    print(name, "distinct outputs:", len(seen), [v[0] for v in seen.values()], "equal to golden text:", [v[1] == want for v in seen.values()], flush=True)
    if len(seen) > 1:
        vals = list(seen.values())
        a, b = vals[0][1].splitlines(), vals[1][1].splitlines()
        for x, y in zip(a, b):
            if x != y:
                print("   ", x[:200]); print("   ", y[:200]); break
