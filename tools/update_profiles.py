"""Copies the summaries of a tools/final_measure.sh run (gpurun_out/<tag>) into profiles/r01_final
and refreshes profiles/traffic.json, which bench.py reads for roofline.traffic / valu_issue.
usage: python tools/update_profiles.py <tag> [dest-dir-name]"""
import glob, json, os, re, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
dest = os.path.join(root, "profiles", sys.argv[2] if len(sys.argv) > 2 else "r01_final")
src = os.path.join(root, "gpurun_out", tag)
os.makedirs(dest, exist_ok=True)
shutil.copy(glob.glob(os.path.join(src, "stats", "**", "*_kernel_stats.csv"), recursive=True)[0], os.path.join(dest, "kernel_stats.csv"))
for a, b in (("pmc_c2/summary.txt", "pmc_c2_summary.txt"), ("pmc_c3/summary.txt", "pmc_c3_summary.txt"),
             ("bench_under_rocprof.json", "bench_under_rocprof.json"), ("perf.log", "perf_survey.txt"),
             ("divergence.log", "divergence.txt"), ("full_parity.txt", "full_size_parity.txt")):
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dest, b))
pmc = {}
for line in open(os.path.join(src, "pmc_c2", "summary.txt")):
    m = re.match(r"(\S+)\s+([0-9.e+]+)\s+\(per launch", line)
    if m:
        pmc[m.group(1)] = float(m.group(2))
util = pmc["SQ_THREAD_CYCLES_VALU"] / pmc["SQ_ACTIVE_INST_VALU"] / 64
tpath = os.path.join(root, "profiles", "traffic.json")
t = json.load(open(tpath))
t.update({"fetch_size_kib": pmc["FETCH_SIZE"], "write_size_kib": pmc["WRITE_SIZE"],
          "traffic_bytes_per_launch": int((pmc["FETCH_SIZE"] * t.get("fetch_factor", 1.0) + pmc["WRITE_SIZE"]) * 1024),
          "valu_insts_per_launch": pmc["SQ_INSTS_VALU"], "valu_lane_utilisation": round(util, 4)})
t["source"] = re.sub(r"profiles/[^/]+/", os.path.relpath(dest, root) + "/", t["source"])
json.dump(t, open(tpath, "w"), indent=1)
# the bench line of that run, with traffic / valu_issue recomputed from ITS OWN pmc passes
b = json.load(open(os.path.join(src, "bench.json")))
k = b["roofline"]["kernel_ms"]
b["roofline"]["traffic"] = t["traffic_bytes_per_launch"]
b["roofline"]["traffic_source"] = t["source"]
rate = pmc["SQ_INSTS_VALU"] / (k * 1e-3) / 1e9
b["roofline"]["valu_issue"] = {"achieved_ginst_per_s": round(rate, 1), "peak_ginst_per_s": 614.4, "frac": round(rate / 614.4, 3),
                               "lane_utilisation": round(util, 4),
                               "insts_per_ray": round(pmc["SQ_INSTS_VALU"] * 64 / b["config"]["rays_per_step"], 1)}
json.dump(b, open(os.path.join(dest, "bench.json"), "w"))
print("value", b["value"], "ms/step", b["ms_per_step"], "kernel ms", k, "valu_issue", b["roofline"]["valu_issue"])
