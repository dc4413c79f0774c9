#!/bin/bash
# PMC passes over the feature path (tools/feat_prof.py, eager launches): results in gpurun_out/pmc_feat/
mkdir -p /root/repo/gpurun_out/pmc_feat; cd /tmp && export TMPDIR=/tmp
i=0
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d /root/repo/gpurun_out/pmc_feat/p$i -o p -- python3 /root/repo/tools/feat_prof.py > /root/repo/gpurun_out/pmc_feat/p$i.log 2>&1
  echo "pass $i rc $?"
done
python3 - <<'PY'
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/root/repo/gpurun_out/pmc_feat/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0]
        if "stft" in k or "floor" in k or "stats" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    print(k)
    for c,vals in sorted(v.items()):
        vals=vals[2:] or vals
        print(f"   {c:28s} {sum(vals)/len(vals):16.1f}  (n={len(vals)})")
PY
