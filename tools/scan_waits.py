#!/usr/bin/env python3
"""(CPU) Where does a kernel wait for memory with little in flight?  Compiles a translation unit for gfx950 with line tables and lists,
per kernel (regex on the mangled name), every `s_waitcnt vmcnt(n)` (--lds: lgkmcnt) together with the loads issued since the wait
before it and the source lines they come from.  A run of "1 load, then vmcnt(0)" entries from neighbouring source lines is a chain of
serial memory round trips -- the pattern of a load the source placed inside a branch (the compiler does not speculate loads) or of a
copy loop it did not pipeline.  Found the collocation kernel's forty round trips per Newton step.
  python tools/scan_waits.py drone-sim-python_amd/csrc/nlp_kernels.hip nlp_solve_kernel [--lds] [--max-loads 3]"""
import argparse, collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('source'); ap.add_argument('kernel')
    ap.add_argument('--lds', action='store_true', help='LDS reads and lgkmcnt waits instead of global / scratch loads and vmcnt')
    ap.add_argument('--max-loads', type=int, default=3, help='list waits with at most this many loads in flight')
    ap.add_argument('--flags', default='-mllvm -disable-machine-licm')
    a = ap.parse_args()
    out = os.path.join(tempfile.mkdtemp(prefix='scan_waits_'), 'k.s')
    cmd = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-Wno-unused-result', '-Wno-unused-value', '-gline-tables-only',
           '-I', os.path.join(ROOT, 'include'), '-I', os.path.dirname(os.path.abspath(a.source)), '-S', '--cuda-device-only', a.source, '-o', out] + a.flags.split()
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    s = open(out).read().split('\n')
    files = {}
    for l in s:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]+)"(?:\s+"([^"]+)")?', l)
        if m:
            files[int(m.group(1))] = os.path.basename(m.group(3) or m.group(2))
    load_re = re.compile(r'^(ds_read|ds_bpermute)' if a.lds else r'^(global_load|scratch_load|flat_load|global_atomic)')
    wait_re = re.compile(r'lgkmcnt\((\d+)\)' if a.lds else r'vmcnt\((\d+)\)')
    for st in [n for n, l in enumerate(s) if re.match(r'^_Z\w+:', l) and re.search(a.kernel, l)]:
        en = next(n for n in range(st + 1, len(s)) if s[n].startswith('.Lfunc_end'))
        cur, pending, groups = None, [], []
        for n in range(st, en):
            t = s[n].strip()
            m = re.match(r'\.loc\s+(\d+)\s+(\d+)', t)
            if m:
                cur = (files.get(int(m.group(1)), '?'), int(m.group(2)))
                continue
            if load_re.match(t):
                pending.append(cur)
            else:
                w = wait_re.search(t) if t.startswith('s_waitcnt') else None
                if w and pending:
                    groups.append((len(pending), int(w.group(1)), sorted(set(pending))))
                    pending = []
        print(f'== {s[st].split(":")[0][:100]}: {len(groups)} waits with loads in flight')
        c = collections.Counter()
        for nl, left, src in groups:
            if nl <= a.max_loads and left == 0:
                c[(src[0], nl)] += 1
        for (src, nl), k in sorted(c.items()):
            print(f'   {src[0]}:{src[1]}: {nl} load(s), then a full wait  x{k}')


if __name__ == '__main__':
    main()
