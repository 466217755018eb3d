#!/usr/bin/env python3
"""timeline of one device step from a rocprofv3 kernel trace (kt_kernel_trace.csv): start offset, duration, stream of every kernel between
two launches of text_index_kernel; argv[1] = the csv, argv[2] = which step (default 5)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/prof_kt/kt_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'text_index' in r['Kernel_Name']]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 5
i0, i1 = idx[k], idx[k + 1]
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i1]:
    s = int(r['Start_Timestamp']); e = int(r['End_Timestamp'])
    print(r['Kernel_Name'][:44].ljust(44), 'start', round((s - t0) / 1000, 1), 'dur', round((e - s) / 1000, 1), 'stream', r.get('Stream_Id'))
print('step', round((int(rows[i1]['Start_Timestamp']) - t0) / 1000, 1), 'us between two index launches')
