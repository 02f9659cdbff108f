show() { python -c "
import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']; c=d.get('cfg3_pipeline',{})
print(sys.argv[1], 'cfg2 %.1f ms  resident %.1f  fused launch %.4f  full-width %.4f | cfg3 %.1f ms' % (d['ms_per_step'], d['resident_path']['ms_per_step'], r['avg_launch_ms'], d['resident_path']['roofline_full_width_launches']['avg_launch_ms'], c.get('ms_per_step', 0)))" $1; }
for rep in 1 2; do
  (cd _old && python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null > ../gpurun_out/ab_old$rep.json); show gpurun_out/ab_old$rep.json
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null > gpurun_out/ab_new$rep.json; show gpurun_out/ab_new$rep.json
done
