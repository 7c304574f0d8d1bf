"""Copies the summaries of a tools/final_measure.sh run (gpurun_out/<tag>) into profiles/<dest> and
refreshes profiles/traffic.json, which bench.py reads for roofline.traffic / valu_issue / hbm.
usage: python tools/update_profiles.py <tag> <dest-dir-name>"""
import glob, json, os, re, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, dest_name = sys.argv[1], sys.argv[2]
dest = os.path.join(root, "profiles", dest_name)
src = os.path.join(root, "gpurun_out", tag)
os.makedirs(dest, exist_ok=True)
for f in glob.glob(os.path.join(src, "stats", "**", "*_kernel_stats.csv"), recursive=True)[:1]:
    shutil.copy(f, os.path.join(dest, "kernel_stats.csv"))
for f in glob.glob(os.path.join(src, "stats_also", "**", "*_kernel_stats.csv"), recursive=True)[:1]:
    shutil.copy(f, os.path.join(dest, "kernel_stats_with_also.csv"))
for a, b in (("bench.json", "bench.json"), ("bench_under_rocprof_also.json", "bench_under_rocprof_also.json"), ("bench_under_rocprof.json", "bench_under_rocprof.json"), ("perf.log", "perf_survey.txt"),
             ("divergence.log", "divergence.txt"), ("calib.log", "fetch_calibration.txt"), ("valu_issue.txt", "ubench_valu_issue.txt"),
             ("hbm_stream.txt", "ubench_hbm_stream.txt"), ("host_path.txt", "host_path.txt"), ("pytest_gpu.log", "pytest_gpu.txt")):
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dest, b))

def pmc_of(name):
    path = os.path.join(src, f"pmc_{name}", "summary.txt")
    if not os.path.exists(path):
        return None
    shutil.copy(path, os.path.join(dest, f"pmc_{name}_summary.txt"))
    out = {}
    for line in open(path):
        m = re.match(r"(\S+)\s+([0-9.e+]+)\s+\(per launch", line)
        if m:
            out[m.group(1)] = float(m.group(2))
    return out

WORKLOADS = {"c2": ("C2", "S0", 1920, 1080, 16, 8), "c3": ("C3", "S1 level 8", 1920, 1080, 64, 8), "c4": ("C4 frame", "S0", 3840, 2160, 256, 8)}
PMC_DIR = {"c2": "c2", "c3": "c3full", "c4": "c4"}
tpath = os.path.join(root, "profiles", "traffic.json")
doc = json.load(open(tpath))
for key, (label, scene, w, h, spp, depth) in WORKLOADS.items():
    pmc = pmc_of(PMC_DIR[key])
    if not pmc or "FETCH_SIZE" not in pmc or "WRITE_SIZE" not in pmc:
        continue
    util = pmc["SQ_THREAD_CYCLES_VALU"] / pmc["SQ_ACTIVE_INST_VALU"] / 64 if "SQ_THREAD_CYCLES_VALU" in pmc else None
    doc["workloads"][key] = {
        "name": label, "scene": scene, "width": w, "height": h, "spp": spp, "depth": depth,
        "fetch_size_kib": pmc["FETCH_SIZE"], "write_size_kib": pmc["WRITE_SIZE"], "fetch_factor": 1.0,
        "traffic_bytes_per_launch": int((pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024),
        "source": f"profiles/{dest_name}/pmc_{PMC_DIR[key]}_summary.txt (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate runs, "
                  "mean of 2 launches; KiB->B; FETCH factor 1.0 for this kernel's divergent 16-B-per-lane record fetches per "
                  f"profiles/{dest_name}/fetch_calibration.txt, the guide's x2 applies to wide streaming reads only; WRITE_SIZE exact for 16-B/lane stores)",
        "valu_insts_per_launch": pmc.get("SQ_INSTS_VALU"), "valu_lane_utilisation": round(util, 4) if util else None,
    }
# ubench peaks measured in the same run
ub = doc.setdefault("ubench", {})
vi = os.path.join(src, "valu_issue.txt")
if os.path.exists(vi):
    best = 0.0
    for line in open(vi):
        m = re.match(r"(fma|pk_fma)\s+\d+\s+[\d.]+\s+[\d.]+\s+([\d.]+)", line)
        if m and m.group(1) == "fma":
            best = max(best, float(m.group(2)))
    if best:
        ub["valu_issue_peak_ginst_per_s"] = round(best * 2.4 * 1024, 1)
hs = os.path.join(src, "hbm_stream.txt")
if os.path.exists(hs):
    reads = [float(m.group(1)) for m in re.finditer(r"read\s+\d+ blocks/CU:\s+[\d.]+ ms\s+(\d+) GB/s", open(hs).read())]
    if reads:
        ub["hbm_read_gbs"] = max(reads)
ub["source"] = (f"profiles/{dest_name}/ubench_valu_issue.txt (independent v_fma_f32 stream on every CU, best wall rate over 1-8 waves "
                f"per SIMD: inst/cycle/SIMD at the nominal 2.4 GHz x 1024 SIMDs x 2.4) and profiles/{dest_name}/ubench_hbm_stream.txt (best read rate)")
json.dump(doc, open(tpath, "w"), indent=1)
print(json.dumps({k: {kk: v[kk] for kk in ("traffic_bytes_per_launch", "valu_insts_per_launch", "valu_lane_utilisation")} for k, v in doc["workloads"].items()}, indent=1))
print(ub)
