import csv,sys,collections,glob
for d in sys.argv[1:]:
    for f in sorted(glob.glob(d+'/*/r_counter_collection.csv')):
        rows=list(csv.DictReader(open(f)))
        agg=collections.defaultdict(lambda: collections.defaultdict(float))
        for r in rows:
            if 'misp_compile' in r['Kernel_Name']: agg[int(r['Dispatch_Id'])][r['Counter_Name']]+=float(r['Counter_Value'])
        ids=sorted(agg)
        last=ids[-2:]   # the two timed launches
        tot=collections.defaultdict(float)
        for i in last:
            for k,v in agg[i].items(): tot[k]+=v
        print(f, {k: f"{v:.4g}" for k,v in tot.items()})
