# what the two timing events attached to the lookup's dispatch cost the timed step (bench.py, headline window), alternating
for r in 1 2 3; do for V in 0 1; do
  DBA_BENCH_NO_LOOKUP_EVENTS=$V python bench.py --no-extras --steps 100 --warmup 20 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no_events=$V', 'value', d['value'], 'step ms', d['ms_per_step'])"
done; done
