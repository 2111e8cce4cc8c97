"""prints the figures of a bench.py JSON line (development aid)"""
import json
import sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])


def roof(r):
    if 'operator' in r:     # hybridized solver
        o = r['operator']
        return (f"post us {r['avg_kernel_ms'] * 1e3:.1f} frac {r['frac']:.3f} traffic {r['traffic']} | K5(H) us {o['avg_kernel_ms'] * 1e3:.1f} "
                f"frac {o['frac']:.3f} iso {o['isolated']['frac']:.3f} | solver {r['solver']['frac']:.3f} {r['solver']['bytes_per_iteration'] / 1e6:.1f} MB/it")
    if 'spmv_nb1' not in r:
        return f"K5(H) us {r['avg_kernel_ms'] * 1e3:.2f} frac {r['frac']:.3f} | solver {r['solver']['frac']:.3f}"
    return (f"K5 us {r['avg_kernel_ms'] * 1e3:.2f} frac {r['frac']:.3f} traffic {r['traffic']} | iso {r['isolated']['frac']:.3f} nb1 {r['spmv_nb1']['frac']:.3f} "
            f"| solver {r['solver']['frac']:.3f} {r['solver']['bytes_per_iteration'] / 1e6:.1f} MB/it")


def point(name, x):
    if 'error' in x:
        print('ERROR', name, x['error'])
        return
    print(name, round(x['value'], 1), 'it', round(x['mean_minres_iterations'], 1), x.get('solver', 'saddle'), '|', roof(x['roofline']) if 'roofline' in x else '',
          '| cpu', x.get('cpu_baseline', {}).get('value'))
    if 'other_solver' in x:
        point('   other solver:', x['other_solver'])


print('headline', round(d['value'], 1), 'samples/s, it', round(d['config']['mean_minres_iterations'], 2), d['config'].get('solver'), 'storage',
      d['config']['precond_storage'], '|', roof(d['roofline']))
print('cpu', d.get('cpu_baseline'))
e = d.get('extra', {})
for k, v in e.items():
    if isinstance(v, dict) and 'error' in v:
        print('ERROR', k, v['error'])
for k in ('saddle_point_minres', 'hybridization', 'fp64_storage', 'hex64', 'r6'):
    if k in e:
        point(k, e[k])
if 'dropin_nb1' in e and 'config2' in e['dropin_nb1']:
    for k, v in e['dropin_nb1'].items():
        if isinstance(v, dict):
            print('dropin', k, round(v['samples_per_s'], 1), 'samples/s', round(v['ms_per_Eval'], 3), 'ms per Eval')
if 'dropin_nb1_config3' in e and 'levels' in e['dropin_nb1_config3']:
    x = e['dropin_nb1_config3']; print('dropin c3 round', round(x['round_64_256_1024_realizations_per_s'], 1), [(l['level'], round(l['realizations_per_s'], 1), {k: round(v, 2) for k, v in l['ms_per_call'].items()}) for l in x['levels']])
if 'mlmc_config3' in e and 'roofline' in e['mlmc_config3']:
    m = e['mlmc_config3']; ro = m['roofline']
    print('c3', round(m['realizations_per_s'], 1), 'launches', m['kernel_launches_per_round'], 'op us', round(ro['avg_kernel_ms'] * 1e3, 1), 'frac', round(ro['frac'], 3),
          'poly us', round(ro['m_block_polynomial']['avg_kernel_ms'] * 1e3, 1), 'frac', round(ro['m_block_polynomial']['frac'], 3), round(ro['m_block_polynomial']['bytes_per_launch'] / 1e6, 1), 'MB',
          'cpu', round(m.get('cpu_baseline', {}).get('realizations_per_s', 0), 1), 'per level s', [round(x, 5) for x in m['seconds_per_sample_per_level']])
for k in ('c4', 'c5'):
    if k in e and 'levels' in e[k]:
        def lv(x):
            return [(l['level'], round(l['realizations_per_s'], 1), round(l.get('sampler_iterations_mean', 0), 1), round(l.get('darcy_iterations_mean', 0), 1)) for l in x['levels']]
        print(k, e[k].get('solver', ''), lv(e[k]), 'mlmc', e[k].get('mlmc_round', {}).get('realizations_per_s'))
        if 'other_solver' in e[k] and 'levels' in e[k]['other_solver']:
            print('   other solver', e[k]['other_solver'].get('solver'), lv(e[k]['other_solver']))
print('walls', {k: round(v.get('wall_s', 0), 1) for k, v in e.items() if isinstance(v, dict) and 'wall_s' in v}, e.get('setup_seconds_in_worker_processes'))
