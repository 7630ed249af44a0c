# coding=utf-8
"""bench.py's output contract: exactly ONE line on stdout, a JSON object with the driver's keys, BASELINE.json's metric,
the `roofline` and `cpu_baseline` objects — single-GPU path and the replicated multi-GPU pipeline (world size 1 over RCCL,
whose version banner must not reach stdout)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ['metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
        'dtype', 'data', 'config', 'roofline', 'cpu_baseline']


def run(*flags):
    from conftest import free_port
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(free_port()))        # (a fresh port per bench process)
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--steps', '20', '--warmup', '5', '--users', '30000', '--items',
                        '9000'] + list(flags), cwd=REPO, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = r.stdout.decode().splitlines()
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_single_gpu_line():
    d = run('--cpu_steps', '2')
    for k in KEYS:
        assert k in d, k
    base = json.load(open(os.path.join(REPO, 'BASELINE.json')))
    assert d['metric'].split(',')[0] in base['metric'] and d['unit'] == 'pairs/s'
    assert d['n_gpus'] == 1 and d['steps'] == 20 and d['warmup'] == 5 and d['higher_is_better'] is True
    assert d['scaling'] == 'weak' and d['vs_baseline'] is None and d['dtype'] == 'f32' and d['data'] == 'synthetic'
    assert 'workload' in d['config'] and 'model' not in d['config']
    assert d['value'] == pytest.approx(128 / (d['ms_per_step'] * 1e-3), rel=1e-3)
    r = d['roofline']
    assert r['bound'] in ('hbm', 'mfma') and r['unit'] in ('GB/s', 'TFLOP/s')
    assert r['frac'] == pytest.approx(r['achieved'] / r['peak'], rel=1e-3) and 0 < r['frac'] < 1
    c = d['cpu_baseline']
    assert c['kind'] in ('port', 'reference') and c['value'] > 0 and c['cores'] >= 1 and c['unit'] == 'pairs/s' and c['sample']


def test_replicated_pipeline_line():
    d = run('--cpu_baseline', '0', '--force_replicated', '1')
    for k in KEYS:
        assert k in d, k
    assert d['n_gpus'] == 1 and d['config']['replicas_bit_identical'] is True
    r = d['roofline']            # the dominant kernel of the timed launch structure, measured in the run (not the dense pass)
    assert r['kernel'] == 'k_bwd' and r['bound'] == 'mfma' and 0 < r['frac'] < 1
    assert r['frac'] == pytest.approx(r['achieved'] / r['peak'], rel=1e-3) and 0 < r['noise_fwd']['frac'] < 1
    assert r['dense_adam_whole_pass']['bound'] == 'hbm' and 0 < r['dense_adam_whole_pass']['frac'] < 1.2


def test_single_gpu_line_is_honest_about_caches_and_mfma():
    """the HBM figure beyond the Infinity Cache, the MFMA fractions of forward / backward computed in the run,
    the launch structure that was timed, and `traffic` labelled as coming from the committed PMC passes."""
    d = run('--cpu_baseline', '0')
    r = d['roofline']
    assert 0 < r['frac_beyond_llc'] < 1 and r['beyond_llc']['params'] * 12 > 1 << 30          # p + m + v > 1 GiB
    assert 'dccf_train_step' in r['launch_structure'] and r['whole_pass']['avg_launch_ms'] > 0
    assert r['traffic'] is None or 'NOT measured in this run' in r['traffic_unit']
    m = d['roofline_mfma']
    for k in ('noise_fwd', 'k_bwd'):
        assert 0 < m[k]['frac'] < 1 and m[k]['flops'] > 0 and m[k]['avg_us'] > 0
    assert 'layout' not in d['config'] or d['config']['layout'] == 'single'


def test_sharded_pipeline_line():
    """--mp sharded (the all-to-all layout BASELINE.json's north_star names) at N = 1: the line names the layout, carries a
    roofline object (the dense Adam pass over the local shard) and the per-phase timings of a step."""
    d = run('--cpu_baseline', '0', '--mp', 'sharded')
    for k in KEYS:
        assert k in d, k
    assert d['n_gpus'] == 1 and d['config']['layout'] == 'sharded' and 'layout=sharded' in d['config']['workload']
    assert d['roofline']['bound'] == 'hbm' and 0 < d['roofline']['frac'] < 1
    assert set(d['phase_us']) >= {'pack', 'a2a_rows', 'unpack', 'fwd_bwd', 'adam'} and d['host_us_per_step'] > 0
    r = run('--cpu_baseline', '0', '--mp', 'replicated')
    assert r['config']['layout'] == 'replicated' and 'layout=replicated' in r['config']['workload']
