import csv,glob,collections,sys,json
def load(d):
    pmc=collections.defaultdict(dict)
    for f in glob.glob(d+"/**/*_counter_collection.csv",recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "leon::k_recon" not in r["Kernel_Name"]: continue
            agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append((int(r["Grid_Size"]),float(r["Counter_Value"])))
        for k,cs in agg.items():
            for c,vals in cs.items():
                g=max(x for x,_ in vals); big=[v for x,v in vals if x==g]
                pmc[k][c]=sum(big)/len(big)
    return pmc
for d in sys.argv[1:]:
    p=load(d)
    for k in sorted(p):
        w=p[k].get("SQ_WAVES",1)
        print(d.split('/')[-1],k.replace('void leon::',''))
        print('   ', {c:round(v/w,1) for c,v in sorted(p[k].items())})
