#!/usr/bin/env python3
"""Compile fit_kernels.hip to gfx950 assembly and print register / scratch / instruction census per kernel.
  python tools/kstat.py [regex]      (default: fit_lm|fit_eval|fit_step)"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, 'drone-sim-python_amd', 'csrc', 'fit_kernels.hip')
out = '/tmp/fit_kernels.s'
subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-S', '--cuda-device-only', '-w'] + os.environ.get('KSTAT_FLAGS', '').split() + [
                       '-I', os.path.join(ROOT, 'include'), '-o', out, src])
pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else 'fit_lm|fit_eval|fit_step')
s = open(out).read()
for m in re.finditer(r'^(\S+):\s*; @\1\n(.*?)\.amdhsa_kernel \1(.*?)\.end_amdhsa_kernel', s, re.S | re.M):
    name, body, meta = m.group(1), m.group(2), m.group(3)
    if not pat.search(name):
        continue
    g = lambda k: re.search(r'\.amdhsa_' + k + r'\s+(\S+)', meta).group(1)     # noqa: E731
    cnt = collections.Counter()
    for line in body.split('\n'):
        t = line.strip().split(' ')[0]
        for key in ('scratch_load', 'scratch_store', 'v_mfma', 'ds_read', 'ds_write', 'v_readlane', 'v_writelane', 'global_load',
                    'global_store', 's_cbranch', 'v_rsq', 'flat_', 'v_pk_fma', 's_waitcnt', 'v_fma_f64', 'v_fma_f32', 'v_fmac_f32'):
            if t.startswith(key):
                cnt[key] += 1
    print(name[:60], '| vgpr', g('next_free_vgpr'), 'sgpr', g('next_free_sgpr'), 'scratch', g('private_segment_fixed_size'),
          '| lines', body.count('\n'))
    print('   ', ' '.join(f'{k}={v}' for k, v in sorted(cnt.items())))
