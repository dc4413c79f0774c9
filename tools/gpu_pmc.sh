#!/bin/bash
# PMC passes over one GEMM shape: CMD is the program after "--" (python3 tools/gemm_one.py ...); results in gpurun_out/pmc_<tag>/
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
TAG=${TAG:-x}
mkdir -p gpurun_out/pmc_$TAG
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|TCC|TCP|GRBM|TA|TD)_[A-Z0-9_a-z]+" | sort -u > gpurun_out/counters_avail.txt; wc -l gpurun_out/counters_avail.txt
i=0
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -k 10 180 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d gpurun_out/pmc_$TAG/p$i -- $CMD > gpurun_out/pmc_$TAG/p$i.log 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then echo "pass $i failed rc=$rc"; tail -3 gpurun_out/pmc_$TAG/p$i.log; fi
done
python3 - <<'PY'
import csv, glob, os, collections
tag=os.environ.get("TAG","x")
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"gpurun_out/pmc_{tag}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    print(k)
    for c,vals in sorted(v.items()):
        vals=vals[2:] or vals
        print(f"   {c:34s} {sum(vals)/len(vals):16.1f}  (n={len(vals)})")
PY
