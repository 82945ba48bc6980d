# the headline and 64/512 lookups with a library whose lookup file (corr_sheared.hip) is the one of the tree before the padded-grid work against the current one, alternating
for r in 1 2 3; do for V in new old; do for W in 25_96 64_512; do
  if [ $V = new ]; then unset DBA_HIP_LIB; else export DBA_HIP_LIB=$PWD/scratch/abl/libdba_hip_oldlookup.so; fi
  python bench.py --window $W --no-extras --steps 40 --warmup 10 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W', '$V', 'lookup us', round(d['roofline']['avg_launch_ms']*1e3,2), 'step ms', d['ms_per_step'])"
done; done; done
