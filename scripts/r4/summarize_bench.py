"""prints the figures of a bench.py JSON line (development aid)"""
import json
import sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d['roofline']
print('headline', round(d['value'], 1), 'samples/s, it', round(d['config']['mean_minres_iterations'], 2), 'storage', d['config']['precond_storage'],
      '| K5 us', round(r['avg_kernel_ms'] * 1e3, 2), 'frac', round(r['frac'], 3), 'traffic', r['traffic'], '| iso', round(r['isolated']['frac'], 3),
      'nb1', round(r['spmv_nb1']['frac'], 3), '| solver', round(r['solver']['frac'], 3), round(r['solver']['bytes_per_iteration'] / 1e6, 1), 'MB/it')
print('cpu', d.get('cpu_baseline'))
e = d.get('extra', {})
for k, v in e.items():
    if isinstance(v, dict) and 'error' in v:
        print('ERROR', k, v['error'])
if 'fp64_storage' in e:
    f = e['fp64_storage']; print('fp64_storage', round(f['value'], 1), 'K5 frac', round(f['roofline']['frac'], 3), 'solver', round(f['roofline']['solver']['frac'], 3))
if 'dropin_nb1' in e:
    print('dropin c2', {k: round(v, 3) if isinstance(v, float) else v for k, v in e['dropin_nb1']['config2'].items()})
    print('dropin c2 graph', round(e['dropin_nb1']['config2_hipgraph']['samples_per_s'], 1), round(e['dropin_nb1']['config2_hipgraph']['ms_per_Eval'], 3))
if 'dropin_nb1_config3' in e:
    x = e['dropin_nb1_config3']; print('dropin c3 round', round(x['round_64_256_1024_realizations_per_s'], 1), [(l['level'], round(l['realizations_per_s'], 1), {k: round(v, 2) for k, v in l['ms_per_call'].items()}) for l in x['levels']])
if 'mlmc_config3' in e:
    m = e['mlmc_config3']; ro = m['roofline']
    print('c3', round(m['realizations_per_s'], 1), 'launches', m['kernel_launches_per_round'], 'op us', round(ro['avg_kernel_ms'] * 1e3, 1), 'frac', round(ro['frac'], 3),
          'poly us', round(ro['m_block_polynomial']['avg_kernel_ms'] * 1e3, 1), 'frac', round(ro['m_block_polynomial']['frac'], 3), round(ro['m_block_polynomial']['bytes_per_launch'] / 1e6, 1), 'MB',
          'cpu', round(m.get('cpu_baseline', {}).get('realizations_per_s', 0), 1), 'per level s', [round(x, 5) for x in m['seconds_per_sample_per_level']])
    print('   phase ms', [{k: round(v, 1) for k, v in p.items()} for p in m['phase_timers_ms']])
if 'hex64' in e:
    h = e['hex64']; print('hex64', round(h['value'], 1), 'it', round(h['mean_minres_iterations'], 1), 'K5 us', round(h['roofline']['avg_kernel_ms'] * 1e3, 1), 'frac', round(h['roofline']['frac'], 3),
                          'solver', round(h['roofline']['solver']['frac'], 3), 'cpu', h.get('cpu_baseline', {}).get('value'), h.get('cpu_baseline', {}).get('sample'))
if 'r6' in e:
    x = e['r6']; print('r6', round(x['value'], 1), 'it', round(x['mean_minres_iterations'], 1), 'K5 us', round(x['roofline']['avg_kernel_ms'] * 1e3, 1), 'frac', round(x['roofline']['frac'], 3),
                       'solver', round(x['roofline']['solver']['frac'], 3), 'cpu', x.get('cpu_baseline', {}).get('value'))
for k in ('c4', 'c5'):
    if k in e and 'levels' in e[k]:
        print(k, [(l['level'], round(l['realizations_per_s'], 1), round(l.get('sampler_iterations_mean', 0), 1), round(l.get('darcy_iterations_mean', 0), 1)) for l in e[k]['levels']],
              'mlmc', e[k].get('mlmc_round', {}).get('realizations_per_s'))
print('walls', {k: round(v.get('wall_s', 0), 1) for k, v in e.items() if isinstance(v, dict) and 'wall_s' in v}, e.get('setup_seconds_in_worker_processes'))
