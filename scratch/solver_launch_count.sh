cd /tmp && export TMPDIR=/tmp
for w in 32_122 64_512; do
rm -rf /tmp/tr_$w
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$w -- python $GRAFT_REPO_ROOT/bench.py --window $w --steps 20 --warmup 4 --no-cpu-baseline --no-extras > /dev/null 2>&1
f=$(ls -t $(find /tmp/tr_$w -name "*kernel_stats.csv") | head -1)
echo "== $w"; grep -i "ba_solve" $f | cut -c1-160
done
