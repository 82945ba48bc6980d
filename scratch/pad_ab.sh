for W in 32_122 9_36_55x55; do
for V in 1 0; do
  DBA_SHEAR_PAD=$V python bench.py --window $W --steps 40 --warmup 10 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['extra']; print('$W', 'pad=$V', 'value', d['value'], 'step ms', d['ms_per_step'], 'lookup us', round(d['roofline']['avg_launch_ms']*1e3,2), 'frac', d['roofline']['frac'], 'kernel', d['roofline']['kernel'][:28], 'build us/edge', e.get('build_us_per_edge'), e.get('build_frac_of_hbm_peak'), 'motion', e.get('motion_filter_us'), 'graph', e.get('step_graph_replay_us'))"
done; done
