"""Print the top kernels of a rocprofv3 --stats CSV:  python tools/show_stats.py <kernel_stats.csv> <steps> [rows]"""
import csv,re,sys
f=sys.argv[1]; div=float(sys.argv[2]); n=int(sys.argv[3]) if len(sys.argv)>3 else 12
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total ms',tot/1e6, 'per step', tot/1e6/div)
for r in rows[:n]:
    nm=re.sub(r'\(.*','',r['Name'].replace('(anonymous namespace)::',''))[:62]
    print(f"{nm:62s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6/div:9.3f} ms/step {float(r['Percentage']):5.1f}%  avg {float(r['AverageNs'])/1e3:8.1f} us")
