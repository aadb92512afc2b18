"""Turns gpurun_out/<tag>/ (scripts/profile_round.sh) into the committed profiles/<tag>_* files."""
import collections, csv, glob, json, shutil, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from bench import csrc_hash
tag = sys.argv[1]
src = Path("gpurun_out") / tag
dst = Path("profiles")
dst.mkdir(exist_ok=True)
shutil.copy(src / "bench.json", dst / f"{tag}_bench.json")
for f in glob.glob(str(src / "stats" / "*" / "*kernel_stats.csv")):
    shutil.copy(f, dst / f"{tag}_kernel_stats.csv")
for f in glob.glob(str(src / "stats_refvehicle" / "*" / "*kernel_stats.csv")):
    shutil.copy(f, dst / f"{tag}_refvehicle_kernel_stats.csv")
if (src / "bench_refvehicle.json").exists():
    shutil.copy(src / "bench_refvehicle.json", dst / f"{tag}_refvehicle_bench.json")
out = {}
for f in glob.glob(str(src / "pmc_*" / "*" / "*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "ftmpc" in k:
            agg[(k.split("(")[0].replace("void ", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        out.setdefault(k, {})[c] = sum(v) / len(v)
(dst / f"{tag}_pmc_summary.json").write_text(json.dumps(out, indent=1))
outrv = {}
for f in glob.glob(str(src / "pmcrv_*" / "*" / "*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "ftmpc" in k:
            agg[(k.split("(")[0].replace("void ", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        outrv.setdefault(k, {})[c] = sum(v) / len(v)
if outrv:
    (dst / f"{tag}_refvehicle_pmc_summary.json").write_text(json.dumps(outrv, indent=1))
bench = json.loads((src / "bench.json").read_text())
dom = bench["roofline"]["kernel"]
pm = out.get("ftmpc::" + dom, {})
if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
    # rocprofv3 reports KiB; gfx950 FETCH_SIZE counts half of wide coalesced reads (MI355X_MICROARCH.md HBM section)
    b = (2.0 * pm["FETCH_SIZE"] + pm["WRITE_SIZE"]) * 1024.0
    cfg = bench["config"]
    (dst / "traffic_latest.json").write_text(json.dumps({
        "kernel": dom, "batch": cfg["batch_per_gpu"], "horizon": cfg["horizon"], "thrusters": cfg["thrusters"],
        "csrc_sha": csrc_hash(),   # bench.py quotes this figure only for the kernel sources it was measured on
        "bytes_per_launch": b, "fetch_size_kib": pm["FETCH_SIZE"], "write_size_kib": pm["WRITE_SIZE"],
        "source": f"profiles/{tag}_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; (2*FETCH+WRITE)*1024)"}, indent=1))
print(json.dumps({dom: pm}, indent=1))
