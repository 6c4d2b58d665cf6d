"""tools/kt_timeline.py <kernel_trace.csv> [rows] -- the last rows of a rocprofv3 kernel trace as a timeline (queue, start, duration)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    name = r["Kernel_Name"]
    name = name.split("fir_")[1][:24] if "fir_" in name else name[:24]
    print("%-26s q%-3s start %9.1f us  dur %8.1f us  grid %s" % (name, r["Queue_Id"], (int(r["Start_Timestamp"]) - t0) / 1e3,
                                                              (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Grid_Size_X"]))
