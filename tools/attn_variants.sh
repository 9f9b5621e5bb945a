#!/bin/bash
# spatial-attention A/B on the GPU box: timing of every kernel form, then SQ counters of the chosen ones (separate --pmc passes)
cd /root/repo && mkdir -p gpurun_out
export TMPDIR=/tmp
tools/micro/attn_bench 20 > gpurun_out/attn_variants.txt 2>&1
cat gpurun_out/attn_variants.txt
for v in ${PMC_VARIANTS:-0}; do
  for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE"; do
    rm -rf /tmp/pmc; (cd /tmp && rocprofv3 --pmc $set --output-format csv -d /tmp/pmc -- /root/repo/tools/micro/attn_bench 20 $v > /tmp/pmc.log 2>&1)
    f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1)
    echo "== variant $v: $set" | tee -a gpurun_out/attn_variants.txt
    python3 - "$f" <<'PY' | tee -a gpurun_out/attn_variants.txt
import csv,sys,collections
acc=collections.defaultdict(float); n=collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    if 'attn_spatial' in r['Kernel_Name']: acc[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
print({k:(v/n[k]) for k,v in acc.items()}, "per launch over", max(n.values()) if n else 0, "launches")
PY
  done
done
