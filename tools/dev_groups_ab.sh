#!/bin/bash
out=gpurun_out/r4; mkdir -p $out
for aa in 0 1 2; do
  D2D_GROUPS_AA=$aa D2D_GROUPS_DIAG=1 timeout -k 10 300 python tools/dev_groups_aa.py $out/aa_$aa.npz 8192 1e-6 150 2>&1 | grep -E "AA=|groups\]" | tail -2
done
D2D_GROUPS_AA=0 timeout -k 10 300 python tools/dev_groups_aa.py $out/aa_ref.npz 8192 1e-9 400 2>&1 | grep "AA="
python - <<'PY'
import numpy as np
ref = np.load('gpurun_out/r4/aa_ref.npz')
for aa in (0, 1, 2):
    d = np.load(f'gpurun_out/r4/aa_{aa}.npz')
    dq = np.abs(d['q'] - ref['q']).max(1) / (1 + np.abs(ref['q']).max(1))
    dc = np.abs(d['cost'] - ref['cost']) / np.abs(ref['cost'])
    print(f'AA={aa} vs plain sweeps to 1e-9: max rel q diff {dq.max():.2e} (p99 {np.percentile(dq, 99):.2e}), max rel cost diff {dc.max():.2e}')
PY
