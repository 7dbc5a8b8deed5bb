import csv,sys,glob,collections
for f in glob.glob(sys.argv[1]+'/**/*kernel_trace.csv', recursive=True):
    d=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        d[r['Kernel_Name'][:28]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000)
    for k,v in d.items():
        if 'k_' in k: print(f"   {k:30s} n={len(v):3d} median {sorted(v)[len(v)//2]:9.1f} us")
