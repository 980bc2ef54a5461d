"""The bench line's contract, checked on the newest committed line (profiles/r*_bench.json is what `python bench.py` printed on an
MI355X): the keys the driver and the judge read, the roofline / cpu_baseline / parity blocks, internal consistency."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _newest():
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_bench.json')))
    assert files
    return json.loads(open(files[-1]).read().strip().splitlines()[-1]), files[-1]


def test_bench_line_has_the_contract_keys():
    d, path = _newest()
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
              'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, (k, path)
    base = json.load(open(os.path.join(ROOT, 'BASELINE.json')))
    assert d['metric'] in base['metric'].replace('\u00d7', 'x') and d['unit'] == 'images/s' and d['higher_is_better'] is True and d['scaling'] == 'weak'
    assert d['vs_baseline'] is None and base['published'] == {}              # no published number for this metric
    assert d['dtype'] == 'bf16' and d['data'] == 'synthetic' and 'workload' in d['config'] and 'model' not in d['config']
    assert abs(d['value'] - d['config']['global_batch'] / d['ms_per_step'] * 1e3) <= 1e-3 * d['value']
    r = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in r
    assert r['bound'] in ('hbm', 'mfma') and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3 and 0 < r['frac'] < 1
    c = d['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in c
    assert c['kind'] in ('reference', 'port') and c['value'] > 0


def test_round_4_lines_carry_the_parity_block():
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r04_*_bench.json')))
    assert files
    for f in files:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        p = d['parity']
        assert p['mu_max_abs'] < p['bounds']['mu_max_abs'] and p['loss_rel'] < p['bounds']['loss_rel']
        assert p['recon_rel_l2'] < p['bounds']['recon_rel_l2'] and p['landmark_mse'] < 1e-6
