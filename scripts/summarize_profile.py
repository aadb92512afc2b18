"""Turns gpurun_out/<tag>/<shape>/ (scripts/profile_round.sh) into the committed profiles/<tag>_<shape>_* files:
bench line, rocprofv3 kernel stats, and a PMC summary (mean counter values per ftmpc kernel)."""
import collections, csv, glob, json, shutil, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from bench import csrc_hash
tag = sys.argv[1]
src = Path("gpurun_out") / tag
dst = Path("profiles")
dst.mkdir(exist_ok=True)
index = {}
for d in sorted(p for p in src.iterdir() if p.is_dir()):
    name = d.name
    pre = f"{tag}_{name}"
    if (d / "bench.json").exists() and (d / "bench.json").stat().st_size:
        shutil.copy(d / "bench.json", dst / f"{pre}_bench.json")
    for f in glob.glob(str(d / "stats" / "*" / "*kernel_stats.csv")):
        shutil.copy(f, dst / f"{pre}_kernel_stats.csv")
    out = {}
    for f in glob.glob(str(d / "pmc_*" / "*" / "*counter_collection.csv")):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "ftmpc" in k:
                agg[(k.split("(")[0].replace("void ", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(agg.items()):
            out.setdefault(k, {})[c] = sum(v) / len(v)
    if out:
        (dst / f"{pre}_pmc_summary.json").write_text(json.dumps(out, indent=1))
    try:
        bench = json.loads((d / "bench.json").read_text())
    except Exception:
        continue
    dom = bench["roofline"]["kernel"]
    key = "ftmpc::" + dom.split(" |")[0]
    pm = out.get(key) or next((v for k, v in out.items() if k.startswith(key) and v.get("FETCH_SIZE", 0) > 1000), {})
    row = {"value": bench["value"], "kernel": dom, "kernel_ms": bench["roofline"]["kernel_ms"], "frac": bench["roofline"]["frac"]}
    if "executed" in bench["roofline"]:
        row["executed_frac"] = bench["roofline"]["executed"]["frac"]
    if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
        # rocprofv3 reports KiB; gfx950 FETCH_SIZE counts half of wide coalesced reads (MI355X_MICROARCH.md HBM section)
        b = (2.0 * pm["FETCH_SIZE"] + pm["WRITE_SIZE"]) * 1024.0
        cfg = bench["config"]
        row["fabric_bytes_per_launch"] = b
        row["fabric_TBps"] = b / (bench["roofline"]["kernel_ms"] * 1e-3) / 1e12
        if name == "headline":
            (dst / "traffic_latest.json").write_text(json.dumps({
                "kernel": dom, "batch": cfg["batch_per_gpu"], "horizon": cfg["horizon"], "thrusters": cfg["thrusters"],
                "csrc_sha": csrc_hash(),   # bench.py quotes this figure only for the kernel sources it was measured on
                "bytes_per_launch": b, "fetch_size_kib": pm["FETCH_SIZE"], "write_size_kib": pm["WRITE_SIZE"],
                "source": f"profiles/{pre}_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; (2*FETCH+WRITE)*1024)"}, indent=1))
    index[name] = row
(dst / f"{tag}_index.json").write_text(json.dumps(index, indent=1))
print(json.dumps(index, indent=1))
