"""Durations of the first update-kernel launches in a rocprofv3 kernel trace (n_live is still ~N there whatever the values).
usage: python scratch/first_updates.py <kernel_trace.csv> [count]"""
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith("ward_update_batch2_kernel")]
rows.sort()
k = int(sys.argv[2]) if len(sys.argv) > 2 else 6
print("launches", len(rows), "first durations us:", [round((e - s) / 1e3, 1) for s, e in rows[:k]])
