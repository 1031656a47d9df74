import sys
from collections import OrderedDict
rows=[l.rstrip('\n').split('\t') for l in open(sys.argv[1])][1:]
top=int(sys.argv[2]) if len(sys.argv)>2 else 40
agg=OrderedDict()
for k,l,gf,mb,ms,tf,gbs in rows:
    a=agg.setdefault(l,[k,0,0,0,0]); a[1]+=float(gf); a[2]+=float(mb); a[3]+=float(ms); a[4]+=1
tot=sum(a[3] for a in agg.values())
print("total ms %.3f  launches %d"%(tot,len(rows)))
kagg={}
for k,l,gf,mb,ms,tf,gbs in rows:
    a=kagg.setdefault(k,[0,0,0,0]); a[0]+=float(gf); a[1]+=float(mb); a[2]+=float(ms); a[3]+=1
for k,a in sorted(kagg.items(), key=lambda kv:-kv[1][2]):
    print(f"  {k:50s} n={a[3]:3d} {a[2]:7.3f} ms {a[0]/a[2] if a[2] else 0:7.1f} TF {a[1]/a[2]:8.1f} GB/s")
for l,a in sorted(agg.items(), key=lambda kv:-kv[1][3])[:top]:
    kn=a[0].replace('dffw::','')
    print(f"{l.replace('DFF_net.',''):48s} {kn:38s} n={a[4]} {a[1]:7.1f} GF {a[2]:7.1f} MB {a[3]:7.3f} ms {a[1]/a[3] if a[3] else 0:6.1f} TF {a[2]/a[3]:7.1f} GB/s")
