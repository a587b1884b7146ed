for n in 983040 1000000 1048576 917504; do
python3 bench.py --rows $n --steps 20 --warmup 3 --repeats 2 --data device --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());n=$n;print('n=%d: %.1f it/s  step %.3f ms  row %.3f ms (%.3f ns/row, %.2f WG rounds)  col %.3f ms (%.3f ns/row)'%(n,d['value'],d['ms_per_step'],d['roofline']['avg_launch_ms'],d['roofline']['avg_launch_ms']*1e6/n, n/256/256,[v for kk, v in d['kernels'].items() if kk.startswith('k_colpass')][0]['avg_launch_ms'],[v for kk, v in d['kernels'].items() if kk.startswith('k_colpass')][0]['avg_launch_ms']*1e6/n))"
done
