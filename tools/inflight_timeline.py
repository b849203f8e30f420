import csv, sys, collections
rows=[]
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Kernel_Name"]
    short = "parse" if "lz4_chunks_kernel" in n else "transp" if "bitswap1_u16" in n else "key" if "dedupe_key" in n else "clear" if "dedupe_clear" in n else "tail" if "inplace_tail_fused" in n else "scan" if "frame_scan" in n else "stash" if "stash" in n else "gather" if "frame_gather" in n else "finish" if "inplace_finish" in n else "marks" if "tail_marks" in n else "other:"+n[:30]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, int(r["Queue_Id"])))
rows.sort()
parses=[r for r in rows if r[2]=="parse"]
# longest run of parses with start gaps < 1.2 ms
best=(0,0); i=0
while i < len(parses):
    j=i
    while j+1 < len(parses) and parses[j+1][0]-parses[j][0] < 1_200_000: j+=1
    if j-i > best[1]-best[0]: best=(i,j)
    i=j+1
a,b=best
t0=parses[a+8][0]; t1=parses[b-8][0]
ncalls=b-8-(a+8)
print("window %.2f ms, %d parses -> %.4f ms/step"%((t1-t0)/1e6, ncalls, (t1-t0)/1e6/ncalls))
win=[r for r in rows if r[0]>=t0 and r[1]<=t1]
dur=collections.defaultdict(list)
for s,e,n,q in win: dur[n].append((e-s)/1e3)
for n,v in sorted(dur.items(), key=lambda kv:-sum(kv[1])): print("  %-10s n %4d mean %8.1f us  sum/step %.3f ms"%(n,len(v),sum(v)/len(v),sum(v)/1e3/ncalls))
# per-queue chains: gaps
byq=collections.defaultdict(list)
for r in win: byq[r[3]].append(r)
gaps=collections.defaultdict(list)
for q,v in byq.items():
    v.sort()
    for x,y in zip(v,v[1:]): gaps[(x[2],y[2])].append((y[0]-x[1])/1e3)
print("gaps (end->start) per queue:")
for k,v in sorted(gaps.items(), key=lambda kv:-sum(kv[1])): print("  %-8s -> %-8s n %4d mean %8.1f us  sum/step %.3f ms"%(k[0],k[1],len(v),sum(v)/len(v),sum(v)/1e3/ncalls))
# call latency: clear start -> finish end per queue
lat=[]
for q,v in byq.items():
    cur=None
    for s,e,n,qq in v:
        if n=="clear": cur=s
        if n in ("finish","tail") and cur is not None: lat.append((e-cur)/1e3); cur=None
print("call latency clear->finish: n %d mean %.1f us"%(len(lat), sum(lat)/max(1,len(lat))))
# concurrency
ev=[]
for s,e,n,q in win:
    if n in("parse","transp"): ev.append((s,1,n)); ev.append((e,-1,n))
ev.sort()
cnt={"parse":0,"transp":0}; last=t0; hist=collections.Counter()
for t,d,n in ev:
    hist[(cnt["transp"],cnt["parse"])]+=t-last; last=t; cnt[n]+=d
tot=sum(hist.values())
print("concurrency (transposes, parses): share")
for k,v in sorted(hist.items(), key=lambda kv:-kv[1])[:12]: print("   T=%d P=%d : %5.1f %%"%(k[0],k[1],100*v/tot))
