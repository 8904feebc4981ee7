import csv,glob,sys,collections
d=sys.argv[1]
f=glob.glob(d+'/*/*_kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
agg=collections.defaultdict(lambda:[0,0.0])
for r in rows:
    n=r['Kernel_Name'].split('(')[0][:44]
    key=(n, r['Grid_Size_X'],r['Grid_Size_Y'],r['Grid_Size_Z'])
    dd=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    agg[key][0]+=1; agg[key][1]+=dd
tot=sum(v[1] for v in agg.values())
items=sorted(agg.items(), key=lambda kv:-kv[1][1])
n=int(sys.argv[2]) if len(sys.argv)>2 else 30
for k,v in items[:n]:
    print('%-46s grid %8s %6s %3s calls %4d total %9.1f us avg %9.1f us %5.1f%%'%(k[0],k[1],k[2],k[3],v[0],v[1],v[1]/v[0],100*v[1]/tot))
print('total GPU ms',tot/1e3)
