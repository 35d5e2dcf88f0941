import csv,glob,sys
f=sorted(glob.glob(sys.argv[1]+'/*/*_kernel_trace.csv'))[-1]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
prev=None
for r in rows[-16:]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print(r['Kernel_Name'][:60].ljust(60), 'dur %.3f ms'%((e-s)/1e6), 'gap %.3f ms'%(((s-prev)/1e6) if prev else 0))
    prev=e
