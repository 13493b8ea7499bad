import csv, glob, sys
O=sys.argv[1]
rows=[r for r in csv.DictReader(open(glob.glob(O+"/kt/**/*kernel_trace.csv", recursive=True)[0])) if "dlm" in r["Kernel_Name"] and "simulate" not in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last iteration: find last occurrence of k_stats_pool and print the kernels between the previous k_stats_pool and it
idx=[i for i,r in enumerate(rows) if "k_stats_pool" in r["Kernel_Name"]]
lo=idx[-2]+1 if len(idx)>=2 else 0; hi=idx[-1]+1
t0=int(rows[lo]["Start_Timestamp"])
for r in rows[lo:hi]:
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} -> {(int(r['End_Timestamp'])-t0)/1e3:9.1f} us q{r['Queue_Id']:>2} grid {r['Grid_Size_X']:>8} {r['Kernel_Name'].replace('void dlm::','')[:60]}")
