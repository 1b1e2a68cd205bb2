"""One-line digest of a bench.py JSON line (development aid): python tools/bench_brief.py file.json"""
import json,sys
d=json.load(open(sys.argv[1]))
b=d['breakdown_ms']; r=d['roofline']; bt=d.get('batched_records') or {}
print(sys.argv[1],'ms/step %.1f fit %.1f solves/step %s brent %s consistent %s redone %s | launches/step %.0f avg %.2f ms rounds/sys %.0f | batched rec/s %.0f redone %s'%(d['ms_per_step'],b['fit'],b['fit_solves_per_step'],b['brent_iterations'],b['consistent'],b['redone_cold'],r['launches_per_step'],r['avg_launch_ms'],r['rounds_per_system'],bt.get('records_per_sec_fit',0),bt.get('redone_cold')))
