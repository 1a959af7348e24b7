"""Per (kernel, grid) durations of the k_ra_* launches in rocprofv3 --kernel-trace csv output: python tools/sum_ra_trace.py <dir> ..."""
import csv,sys,glob,collections
for d in sys.argv[1:]:
    fs=glob.glob(d+'/**/*kernel_trace.csv',recursive=True)
    rows=[]
    for f in fs: rows+=list(csv.DictReader(open(f)))
    agg=collections.defaultdict(list)
    for r in rows:
        n=r['Kernel_Name']
        if 'k_ra_' not in n: continue
        import re; m=re.search(r'k_\w+(<[^>]*>)?',n); short=m.group(0)[:40] if m else n[:40]
        key=(short,int(r['Grid_Size_X'])*int(r.get('Grid_Size_Y',1) or 1))
        agg[key].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
    print(d)
    for k,v in sorted(agg.items(), key=lambda kv:(-kv[0][1],kv[0][0])):
        v=sorted(v); print(f"  {k[0]:42s} grid {k[1]:9d}  n {len(v):3d}  median {v[len(v)//2]:8.1f} us  min {v[0]:8.1f}")
