"""The line the driver parses (the LAST stdout line of bench.py) must stay small: round 4's 20.9 kB line was not parsed."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config')


def fake_record(pad):
    line = {'metric': 'trajectory-optimisations/sec (6-seg poly, 50 wpts)', 'value': 1.2e6, 'unit': 'trajectory-optimisations/s', 'n_gpus': 1, 'steps': 20, 'warmup': 5,
            'ms_per_step': 3.4, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64 residual/gradient + f32 MFMA J^T J', 'data': 'synthetic',
            'config': {'workload': 'w' * pad, 'solver': 's' * pad, 'handout': 'h' * pad, 'parallelism': 'trajectory-sharded x1', 'max_iter': 150},
            'rank_kernel_ms_per_step_min_max': [3.3, 3.5]}
    line.update({f'hoisted_scalar_number_{i}': 0.123456789012345 for i in range(30)})
    roof = {'bound': 'issue', 'priced_against': 'mfma', 'executed_mfma_flop_per_unit': 188000.0, 'frac_of_executed_flop': 0.061, 'mfma_busy_frac': 0.069,
            'issue_slot_frac': 0.70, 'wave_slots_occupied': 1.4, 'pmc_source': 'profiles/r06/01_bench_issue.json', 'kernel': 'k' * pad, 'achieved': 18.0, 'peak': 157.3, 'unit': 'TFLOP/s', 'frac': 0.117, 'traffic': 1.0e7, 'traffic_source': {'file': 'f' * pad},
            'alg_flop_per_unit': 470400, 'units_per_launch_avg': 134347.5, 'avg_launch_us': 3440.0, 'launches': 20, 'note': 'n' * 10 * pad}
    cpu = {'value': 720.0, 'unit': 'trajectory-optimisations/s', 'cores': 16, 'kind': 'port', 'sample': 'x' * pad, 'host_cpu_count': 256,
           'oracle_lm': {'sample': 'y' * 10 * pad}}
    return line, roof, cpu


def test_final_line_is_small_and_carries_the_contract():
    for pad in (10, 300, 5000):
        line, roof, cpu = fake_record(pad)
        s = bench.compact_line(line, roof, cpu, 'gpurun_out/bench_detail.json')
        assert len(s) <= bench.LINE_LIMIT < 6000 and '\n' not in s
        rec = json.loads(s)
        for k in CONTRACT:
            assert k in rec
        assert rec['value'] == 1.2e6 and rec['config']['parallelism'] == 'trajectory-sharded x1'
        assert set(rec['roofline']) == set(bench.ROOF_KEYS) and rec['roofline']['frac'] == 0.117 and rec['roofline']['bound'] == 'issue' and rec['roofline']['mfma_busy_frac'] == 0.069
        assert set(rec['cpu_baseline']) == set(bench.CPU_KEYS) and rec['cpu_baseline']['cores'] == 16 and rec['cpu_baseline']['kind'] == 'port'
        assert rec['rank_kernel_ms_per_step_min_max'] == [3.3, 3.5]
        assert rec['detail'] == 'gpurun_out/bench_detail.json'


def test_no_records_still_a_line():
    line, _, _ = fake_record(10)
    rec = json.loads(bench.compact_line(line, None, None))
    assert rec['roofline'] is None and rec['cpu_baseline'] is None


def test_emit_prints_the_small_line_last(capsys, tmp_path, monkeypatch):
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    line, roof, cpu = fake_record(300)
    bench.emit(line, dict(line, roofline=roof, cpu_baseline=cpu, parity={'others': ['z' * 20000]}), roof, cpu)
    out = capsys.readouterr().out.strip().split('\n')
    assert len(out) == 2 and out[0].startswith(bench.DETAIL_PREFIX) and len(out[0]) > 20000
    assert len(out[1]) <= bench.LINE_LIMIT and json.loads(out[1])['detail'] == 'bench_detail.json'
    assert json.load(open(tmp_path / 'bench_detail.json'))['parity']['others'][0].startswith('zzz')
