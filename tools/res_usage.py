#!/usr/bin/env python3
"""Condense `hipcc -Rpass-analysis=kernel-resource-usage` remarks (stderr text on stdin or a file) into one line per kernel."""
import re, sys, subprocess
txt = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
cur = None
rows = []
for line in txt.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    body = m.group(1).strip()
    if body.startswith("Function Name:"):
        cur = {"name": body.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in body:
        k, v = body.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", r["name"]], capture_output=True, text=True).stdout.strip()
    except Exception:
        name = r["name"]
    name = re.sub(r"\(.*", "", name)
    print(f"{name:70s} vgpr {r.get('VGPRs','?'):>4} agpr {r.get('AGPRs','?'):>3} sgpr {r.get('TotalSGPRs', r.get('SGPRs','?')):>4} scratch {r.get('ScratchSize [bytes/lane]','?'):>5} occ {r.get('Occupancy [waves/SIMD]','?'):>2} lds {r.get('LDS Size [bytes/block]','?')}")
