import csv, sys, collections, re
def load(path):
    d=collections.defaultdict(lambda: collections.defaultdict(float))
    n=collections.defaultdict(int)
    dur=collections.defaultdict(float)
    seen=set()
    for r in csv.DictReader(open(path)):
        k=r['Kernel_Name']
        k=re.sub(r'^void ','',k); k=re.sub(r'\(.*$','',k)
        d[k][r['Counter_Name']]+=float(r['Counter_Value'])
        key=(r['Dispatch_Id'])
        if key not in seen:
            seen.add(key); n[k]+=1; dur[k]+=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    return d,n,dur
tabs=[load(p) for p in sys.argv[1:]]
kernels=sorted(tabs[0][2], key=lambda k:-tabs[0][2][k])[:int(14)]
for k in kernels:
    print(f"== {k}  n={tabs[0][1][k]} time={tabs[0][2][k]:.0f} us")
    for d,n,dur in tabs:
        for c,v in sorted(d[k].items()):
            print(f"     {c:28s} {v:16.0f}")
