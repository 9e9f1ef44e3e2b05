"""The last step of a kernel trace as a timeline: tools/timeline.py <kernel_trace.csv> [steps back]
A step starts at the first rz_begin_kernel after a gap of more than 100 us... simply: prints the last N kernels."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = rows[-n:]
t0 = int(rows[0]['Start_Timestamp'])
for r in rows:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    print(f"{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:8.1f}  q{r['Queue_Id']:>2s} s{r['Stream_Id']:>2s}  grid {r['Grid_Size_X']:>8s}x{r['Grid_Size_Y']:>5s}x{r['Grid_Size_Z']:>4s} wg {r['Workgroup_Size_X']:>4s}  {r['Kernel_Name'][:58]}")
