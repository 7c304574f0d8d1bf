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
             ("divergence.log", "divergence.txt"), ("fetch_calibration.txt", "fetch_calibration.txt"), ("valu_issue.txt", "ubench_valu_issue.txt"),
             ("node_fetch.txt", "ubench_vmem_records.txt"), ("perf_rays2.log", "perf_survey_rays2.txt"),
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

WORKLOADS = {"c2": ("C2", "S0", 1920, 1080, 16, 8), "c3": ("C3", "S1 level 8", 1920, 1080, 64, 8), "c4": ("C4 frame", "S0", 3840, 2160, 256, 8),
             "c5": ("C5 frame", "S2 level 8", 1920, 1080, 16, 8)}
PMC_DIR = {"c2": "c2", "c3": "c3full", "c4": "c4", "c5": "c5full16"}
for extra in ("c2_lds", "c2_rays2"):   # the LDS-pipeline pass and the two-rays-per-lane experiment: kept as evidence, not read by bench.py
    pmc_of(extra)
FETCH_FACTOR = 2.0   # profiles/<dest>/fetch_calibration.txt: every L2 miss moves a 128-byte line that FETCH_SIZE tallies at 64 bytes
tpath = os.path.join(root, "profiles", "traffic.json")
doc = json.load(open(tpath))
for key, (label, scene, w, h, spp, depth) in WORKLOADS.items():
    pmc = pmc_of(PMC_DIR[key])
    if not pmc or "FETCH_SIZE" not in pmc or "WRITE_SIZE" not in pmc:
        continue
    util = pmc["SQ_THREAD_CYCLES_VALU"] / pmc["SQ_ACTIVE_INST_VALU"] / 64 if "SQ_THREAD_CYCLES_VALU" in pmc else None
    doc["workloads"][key] = {
        "name": label, "scene": scene, "width": w, "height": h, "spp": spp, "depth": depth,
        "fetch_size_kib": pmc["FETCH_SIZE"], "write_size_kib": pmc["WRITE_SIZE"], "fetch_factor": FETCH_FACTOR,
        "traffic_bytes_per_launch": int((pmc["FETCH_SIZE"] * FETCH_FACTOR + pmc["WRITE_SIZE"]) * 1024),
        "source": f"profiles/{dest_name}/pmc_{PMC_DIR[key]}_summary.txt (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate runs, "
                  f"mean of 2 launches; KiB->B; FETCH_SIZE x {FETCH_FACTOR}: the guide's gfx950 correction, confirmed for this kernel's record "
                  f"shapes (64 / 96 / 48-byte records, 16 B per lane) in profiles/{dest_name}/fetch_calibration.txt; WRITE_SIZE exact for 16-B/lane stores)",
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
nf = os.path.join(src, "node_fetch.txt")
if os.path.exists(nf):
    # records per cycle per CU of the kernel's two record shapes, 16 waves per CU, with a step's worth of arithmetic between
    # fetches: tables that fit one L2 (2 MB) and tables served by the Infinity Cache (32 MB)
    vm = {}
    for line in open(nf):
        m = re.match(r"lane/record\s+(\d)x16B of\s+(\d+)B fill\s+(\d+) table\s+(\d+) KB waves/CU (\d+):\s+[\d.]+ ms\s+([\d.]+) records/cycle/CU", line)
        if m and int(m.group(3)) > 0 and m.group(5) == "16":
            shape = {"96": "node96", "48": "tri48", "64": "node64"}.get(m.group(2))
            kb = int(m.group(4))
            level = "l1" if kb <= 16 else "l2" if kb <= 2048 else "l2x8" if kb <= 8192 else "mall" if kb <= 32768 else "hbm"
            if shape and m.group(1) in ("6", "3", "4") and not (m.group(2) == "128"):
                vm[f"{shape}_{level}"] = float(m.group(6))
    if vm:
        ub["vmem_records_per_cycle_per_cu"] = vm
        ub["vmem_source"] = (f"profiles/{dest_name}/ubench_vmem_records.txt (tools/ubench/node_fetch.hip: every lane follows its own chain of random "
                             "records, L x 16 B per record, ~100 / 40 dependent fmas between fetches; the rate does not depend on the waves per CU from 4 up)")
ub["source"] = (f"profiles/{dest_name}/ubench_valu_issue.txt (independent v_fma_f32 stream on every CU, best wall rate over 1-8 waves "
                f"per SIMD: inst/cycle/SIMD at the nominal 2.4 GHz x 1024 SIMDs x 2.4) and profiles/{dest_name}/ubench_hbm_stream.txt (best read rate)")
json.dump(doc, open(tpath, "w"), indent=1)
print(json.dumps({k: {kk: v[kk] for kk in ("traffic_bytes_per_launch", "valu_insts_per_launch", "valu_lane_utilisation")} for k, v in doc["workloads"].items()}, indent=1))
print(ub)
