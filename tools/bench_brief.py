"""One-line digest of a bench.py JSON line (development aid): python tools/bench_brief.py file.json"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r, e = d['roofline'], d.get('roofline_eval') or {}
s, f, bt = d.get('single_record') or {}, d.get('roofline_eval_fused') or {}, d.get('batched_records') or {}
print('%s: %s %.1f %s, %.1f ms/step, fit roofline %.3f, eval roofline %.3f | single record %.1f ms (max %.1f) | call with hull '
      '%.3f ms = %.3g points/s | batched %.0f records/s | cpu %.4g %s'
      % (sys.argv[1], d['config'].get('workload', '')[:24], d['value'], d['unit'], d['ms_per_step'], r.get('frac', float('nan')),
         e.get('frac', float('nan')), s.get('ms_per_step', float('nan')), (s.get('step_ms') or {}).get('max', float('nan')),
         f.get('call_ms_hull_on', float('nan')), f.get('points_per_sec_hull_on', float('nan')),
         bt.get('records_per_sec', float('nan')), (d.get('cpu_baseline') or {}).get('value', float('nan')),
         (d.get('cpu_baseline') or {}).get('unit', '')))
