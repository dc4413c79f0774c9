#!/bin/bash
# HBM traffic of the bench's kernels from the PMC counters: two separate rocprofv3 passes (FETCH_SIZE, WRITE_SIZE) with
# kernel-trace only, hipGraphs off so every dispatch is a plain kernel launch.  Output: gpurun_out/pmc_traffic_raw.json
mkdir -p gpurun_out && rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
export LIDK_GRAPHS=0
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --cavg-steps 0 --fit-epochs 0 --resident 1 > gpurun_out/pmc_fetch.log 2>&1; echo "fetch rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --cavg-steps 0 --fit-epochs 0 --resident 1 > gpurun_out/pmc_write.log 2>&1; echo "write rc=$?"
python3 - <<'PY'
import csv, glob, json, collections
out = {}
for tag, cname in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmc_{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == cname:
                agg[r["Kernel_Name"].split("(")[0][:60]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out.setdefault(k, {})[cname + "_KB_per_launch_raw"] = round(sum(v) / len(v), 1)
        out[k]["launches_" + tag] = len(v)
json.dump(out, open("gpurun_out/pmc_traffic_raw.json", "w"), indent=1, sort_keys=True)
tot_f = sum(v.get("FETCH_SIZE_KB_per_launch_raw", 0) * v.get("launches_fetch", 0) for v in out.values())
tot_w = sum(v.get("WRITE_SIZE_KB_per_launch_raw", 0) * v.get("launches_write", 0) for v in out.values())
print("total raw FETCH_SIZE KB", tot_f, "WRITE_SIZE KB", tot_w)
for k in sorted(out, key=lambda k: -out[k].get("WRITE_SIZE_KB_per_launch_raw", 0) * out[k].get("launches_write", 0))[:14]:
    print(k[:50], out[k])
PY
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
