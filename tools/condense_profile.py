#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile_bench.sh into the files committed under profiles/:
  <prefix>_kernel_stats.csv        the kernel-trace --stats summary (as rocprofv3 wrote it)
  <prefix>_pmc_summary.txt         per-kernel mean of every collected counter
  <prefix>_traffic.json            per-launch HBM traffic of the hot kernels from the FETCH_SIZE / WRITE_SIZE passes, as
                                   MI355X_MICROARCH.md prescribes (separate passes; the counters are in KiB; FETCH_SIZE of wide
                                   coalesced streaming reads reports half the bytes on gfx950: doubled for the kernel whose
                                   reads are 16-B-per-lane streams, left as is -- and flagged uncalibrated -- elsewhere)
  python tools/condense_profile.py gpurun_out/r2_prof profiles/r02/06_bench_final"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, prefix = sys.argv[1], sys.argv[2]
os.makedirs(os.path.dirname(prefix), exist_ok=True)
shutil.copy(os.path.join(src, 'trace', 'bench_kernel_stats.csv'), prefix + '_kernel_stats.csv')
if os.path.exists(os.path.join(src, 'bench.json')):
    shutil.copy(os.path.join(src, 'bench.json'), prefix + '_under_rocprof.json')
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(src, 'pmc_*', '*counter_collection.csv'))):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
with open(prefix + '_pmc_summary.txt', 'w') as out:
    for k in sorted(acc):
        if not any(t in k for t in ('fit_', 'gvf_', 'track_', 'nlp_', 'gradient_')):
            continue
        out.write(k[:110] + '\n')
        for c, v in sorted(acc[k].items()):
            out.write(f'    {c:34s} launches={len(v):4d} mean={sum(v) / len(v):.6g} min={min(v):.6g} max={max(v):.6g}\n')
traffic = {}
for k, cs in acc.items():
    if 'FETCH_SIZE' in cs and 'WRITE_SIZE' in cs:
        name = next((n for n in ('fit_lm_knot_kernel', 'fit_lm_kernel', 'fit_jtj_kernel', 'gvf_run_kernel', 'track_run_kernel', 'nlp_solve_kernel') if n in k), None)
        if name is None and 'gvf_run_quad' in k:
            name = 'gvf_run_kernel'         # (round 6: the DPP-quad instantiations of the same loop; one of them serves a launch)
        if name is None:
            continue
        fetch_kib = sum(cs['FETCH_SIZE']) / len(cs['FETCH_SIZE'])
        write_kib = sum(cs['WRITE_SIZE']) / len(cs['WRITE_SIZE'])
        wide = name == 'fit_jtj_kernel'                 # 16-B-per-lane streaming reads: FETCH_SIZE reports half of them
        traffic.setdefault(name, []).append({
            'kernel': k[:80], 'launches': len(cs['FETCH_SIZE']), 'fetch_bytes_per_launch': fetch_kib * 1024 * (2 if wide else 1),
            'write_bytes_per_launch': write_kib * 1024, 'fetch_doubled': wide,
            'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KiB), separate passes, mean over the launches of the pass'
                    + ('' if wide else '; 8-B-per-lane loads: FETCH_SIZE uncalibrated for this width (MI355X_MICROARCH.md)')})
json.dump(traffic, open(prefix + '_traffic.json', 'w'), indent=1)
# executed fp64 flop of the simulation kernels (they are bound by the fp64 vector pipe, not by HBM): one pass of
# SQ_INSTS_VALU_{FMA,MUL,ADD,TRANS}_F64 (wave-instructions): flop = (2 FMA + MUL + ADD) x 64 lanes, per launch
valu = {}
for k, cs in acc.items():
    name = next((n for n in ('gvf_run_kernel', 'track_run_kernel', 'nlp_solve_kernel', 'nlp_groups_kernel') if n in k), None)
    if name is None and 'gvf_run_quad_wide' in k:
        name = 'gvf_run_kernel'             # (the one-wave-per-SIMD instantiation: BASELINE configs[4])
    if name is None or 'SQ_INSTS_VALU_FMA_F64' not in cs:
        continue
    m = lambda c: sum(cs[c]) / len(cs[c]) if c in cs else 0.0        # noqa: E731
    valu[name] = {'kernel': k[:80], 'launches': len(cs['SQ_INSTS_VALU_FMA_F64']),
                  'fma_f64': m('SQ_INSTS_VALU_FMA_F64'), 'mul_f64': m('SQ_INSTS_VALU_MUL_F64'), 'add_f64': m('SQ_INSTS_VALU_ADD_F64'),
                  'trans_f64': m('SQ_INSTS_VALU_TRANS_F64'), 'valu_insts': m('SQ_INSTS_VALU'),
                  'fp64_flop_per_launch': (2 * m('SQ_INSTS_VALU_FMA_F64') + m('SQ_INSTS_VALU_MUL_F64') + m('SQ_INSTS_VALU_ADD_F64')) * 64,
                  'note': 'rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 '
                          '(wave-instructions, mean over the launches of the pass); executed flop = (2 FMA + MUL + ADD) x 64'}
json.dump(valu, open(prefix + '_valu.json', 'w'), indent=1)
# what the SIMDs did during a launch of the fused solver kernels (bench.py pmc_issue -> the parsed roofline object): per-launch means of the
# SQ_* passes.  SQ_BUSY_CYCLES counts per shader engine (32 on the chip): / 32 = cycles of the launch; x 1024 SIMDs = SIMD-cycles;
# SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES are in quad-cycles per SIMD (MI355X_MICROARCH.md).  The evaluation units of the same launches come
# from the bench line the MFMA pass printed (roofline.units_per_launch_avg).
issue = {}
units = None
for tag in ('SQ_INSTS_VALU_MFMA_MOPS_F32', 'SQ_WAVE_CYCLES'):
    try:
        line = [l for l in open(os.path.join(src, f'pmc_{tag}.json')).read().splitlines() if l.startswith('{')][-1]
        units = json.loads(line)['roofline']['units_per_launch_avg']
        break
    except (OSError, KeyError, IndexError, ValueError, TypeError):
        continue
for k, cs in acc.items():
    name = next((n for n in ('fit_lm_knot_kernel', 'fit_lm_kernel') if n in k), None)
    if name is None or 'SQ_BUSY_CYCLES' not in cs or 'SQ_INSTS_VALU_MFMA_MOPS_F32' not in cs:
        continue
    m = lambda c: sum(cs[c]) / len(cs[c]) if c in cs else 0.0        # noqa: E731
    cycles = m('SQ_BUSY_CYCLES') / 32.0
    simd_cycles = cycles * 1024.0
    mfma = m('SQ_INSTS_VALU_MFMA_MOPS_F32') / 4.0
    issue[name] = {'kernel': k[:80], 'launches': len(cs['SQ_BUSY_CYCLES']), 'cycles_per_launch': cycles, 'mfma_insts_per_launch': mfma,
                   'units_per_launch': units, 'mfma_insts_per_unit': (mfma / units) if units else None,
                   'mfma_busy_frac': m('SQ_VALU_MFMA_BUSY_CYCLES') / simd_cycles, 'issue_slot_frac': m('SQ_ACTIVE_INST_ANY') / (simd_cycles / 4.0),
                   'wave_slots_occupied': m('SQ_WAVE_CYCLES') / (simd_cycles / 4.0),
                   'valu_insts': m('SQ_INSTS_VALU'), 'lds_insts': m('SQ_INSTS_LDS'), 'salu_insts': m('SQ_INSTS_SALU'),
                   'note': 'rocprofv3 --pmc passes of tools/profile_bench.sh, mean over the launches of a pass; cycles = SQ_BUSY_CYCLES / 32, SIMD-cycles = x 1024, '
                           'MFMA instructions = SQ_INSTS_VALU_MFMA_MOPS_F32 / 4 (v_mfma_f32_16x16x4_f32 = 2048 flop), issue slots / wave slots in quad-cycles'}
json.dump(issue, open(prefix + '_issue.json', 'w'), indent=1)
print(open(prefix + '_pmc_summary.txt').read()[:3000])
print(json.dumps(traffic, indent=1)[:2000])
