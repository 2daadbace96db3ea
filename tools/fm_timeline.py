import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'fm_gather' in r['Kernel_Name']][-1]
t0=int(rows[idx]['Start_Timestamp'])
for r in rows[idx:idx+14]:
    n=r['Kernel_Name'].split('(anonymous namespace)::')[-1].split('(')[0]
    print(f"{n:22s} start {(int(r['Start_Timestamp'])-t0)/1e3:8.1f} us  dur {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:7.1f}  grid {r['Grid_Size_X']},{r['Grid_Size_Y']},{r['Grid_Size_Z']}")
