import csv, collections, re, sys
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(lambda: collections.defaultdict(list)); dur={}
for r in rows:
    m=re.search(r"(k_\w+(?:<[^>]*>)?)", r['Kernel_Name'])
    if not m or 'gemm' not in m.group(1): continue
    k=m.group(1); agg[k][r['Counter_Name']].append(float(r['Counter_Value'])); dur[(k,r['Dispatch_Id'])]=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,c in agg.items():
    ds=[v for (kk,d),v in dur.items() if kk==k]; t=sum(ds)/len(ds)
    g=sum(c['GRBM_GUI_ACTIVE'])/len(c['GRBM_GUI_ACTIVE']); mb=sum(c['SQ_VALU_MFMA_BUSY_CYCLES'])/len(c['SQ_VALU_MFMA_BUSY_CYCLES'])
    print(k.ljust(36), "avg_us %.1f clock %.3f GHz mfma_busy %.3f" % (t/1e3, g/8/t, mb/(g/8*1024)))
