import csv,glob,sys,collections
for pat,name in (("gpurun_out/xcd/pmcF/**/*counter_collection.csv","FETCH_SIZE"),("gpurun_out/xcd/pmcW/**/*counter_collection.csv","WRITE_SIZE")):
    acc=collections.defaultdict(list)
    for f in glob.glob(pat,recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"]==name: acc[r["Kernel_Name"][:90]].append(float(r["Counter_Value"]))
    for k,v in acc.items(): print(name,k,"n=%d avg=%.0f"%(len(v),sum(v)/len(v)))
