"""Summarise rocprofv3 --pmc CSVs (one dir per pass) for the trace kernel."""
import csv, glob, os, sys
out = sys.argv[1]
rows = {}
for f in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "trace_kernel" not in r["Kernel_Name"]:
            continue
        key = r["Counter_Name"]
        rows.setdefault(key, []).append(float(r["Counter_Value"]))
summary = {k: sum(v) / len(v) for k, v in rows.items()}
with open(os.path.join(out, "summary.txt"), "w") as fo:
    for k in sorted(summary):
        line = f"{k:32s} {summary[k]:.6g}   (per launch, mean of {len(rows[k])})"
        print(line); fo.write(line + "\n")
    s = summary
    def d(a, b): return s[a] / s[b] if a in s and b in s and s[b] else float("nan")
    extra = [
        ("VALU lane utilisation  SQ_THREAD_CYCLES_VALU/(SQ_ACTIVE_INST_VALU*64)?", d("SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU") / 64),
        ("VALU insts per wave", d("SQ_INSTS_VALU", "SQ_WAVES")),
        ("wave cycles busy share: ACTIVE_INST_ANY/WAVE_CYCLES", d("SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES")),
        ("WAIT_ANY/WAVE_CYCLES", d("SQ_WAIT_ANY", "SQ_WAVE_CYCLES")),
        ("WAIT_INST_ANY/WAVE_CYCLES", d("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES")),
        ("L2 hit rate", s.get("TCC_HIT_sum", 0) / max(1.0, s.get("TCC_HIT_sum", 0) + s.get("TCC_MISS_sum", 0))),
        ("L1 miss share: TCP_TCC_READ_REQ/TCP_TOTAL_CACHE_ACCESSES", d("TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum")),
    ]
    for k, v in extra:
        line = f"{k:72s} {v:.4f}"
        print(line); fo.write(line + "\n")
