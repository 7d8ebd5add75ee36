"""Sum the per-dispatch counters of the two passes written by tools/pmc_bench.sh / pmc_bwd.sh per kernel.

    python tools/pmc_show.py k_lift_b_mfma k_cgp_rate
"""
import csv,collections,sys
pat=sys.argv[1:]
for f in ['gpurun_out/pmc_w1/w1_counter_collection.csv','gpurun_out/pmc_w2/w2_counter_collection.csv']:
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open('/root/repo/'+f)):
        k=r['Kernel_Name'][:70]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
    for k,v in agg.items():
        if any(p in k for p in pat):
            print(k)
            for c,x in sorted(v.items()): print('   ',c,f'{x:.4g}')
