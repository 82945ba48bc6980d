for W in 9_36_55x55 32_122; do
for V in default lcap96 lcap104 lcap112; do
  if [ $V = default ]; then unset DBA_HIP_LIB; else export DBA_HIP_LIB=$PWD/scratch/abl/libdba_hip_$V.so; fi
  python bench.py --window $W --no-extras --steps 40 --warmup 10 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W', '$V', 'lookup us', round(d['roofline']['avg_launch_ms']*1e3,2), 'frac', d['roofline']['frac'], 'value', d['value'])"
done; done
