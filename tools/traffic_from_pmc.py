"""HBM-side bytes per launch of one kernel from two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE).

    python tools/traffic_from_pmc.py <fetch counter_collection.csv> <write counter_collection.csv> <kernel substring> > out.json

Units and the gfx950 correction follow MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are in KB
per dispatch; FETCH_SIZE reports half of the bytes of wide (16 B / lane) coalesced reads on gfx950 and is
doubled; WRITE_SIZE is exact for 16-B-per-lane stores.  Infinity-Cache hits are included in both.
"""
import csv
import json
import sys


def total(path, counter, needle):
    s, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and needle in r["Kernel_Name"]:
            s += float(r["Counter_Value"])
            n += 1
    return s, n


fetch_kb, nf = total(sys.argv[1], "FETCH_SIZE", sys.argv[3])
write_kb, nw = total(sys.argv[2], "WRITE_SIZE", sys.argv[3])
assert nf > 0 and nf == nw, (nf, nw)
out = {
    "kernel": sys.argv[3], "launches": nf,
    "fetch_bytes_per_launch_raw": fetch_kb * 1024 / nf,
    "fetch_bytes_per_launch": 2 * fetch_kb * 1024 / nf,
    "write_bytes_per_launch": write_kb * 1024 / nw,
    "bytes_per_launch": (2 * fetch_kb + write_kb) * 1024 / nf,
    "corrections": "FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B), WRITE_SIZE as read; KB = 1024 B",
}
json.dump(out, sys.stdout)
print()
