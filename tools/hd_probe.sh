set -e
for args in "--batch 256" "--batch 64" "--workload hd --batch 64" "--workload hd --batch 16" "--workload hd --batch 128"; do
  python bench.py $args --steps 10 --warmup 3 --no-cpu --no-frontend --no-host-api --no-warp 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; p=d['roofline_pyr_grad_pass']
print('$args', 'l0 %.4f ms frac %.3f | pass %.4f ms frac %.3f | lk %.3f ms | step %.3f' % (r['avg_launch_ms'], r['frac'], p['avg_pass_ms'], p['frac'], d['lk_kernel']['avg_launch_ms'], d['ms_per_step']))"
done
