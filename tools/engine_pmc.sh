#!/bin/bash
# SQ counters of the decode engine kernel (one scene, decode steps 1101..1104 of one frame: the mean KV length), one rocprofv3 --pmc pass per counter set.
# Per launch: 256 workgroups x 8 waves; the kernel lasts ~440 us = ~1.06 M cycles at 2.4 GHz, i.e. ~1.08 G SIMD-cycles on 1024 SIMDs.
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_MISC" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
rm -rf /tmp/pmc; UMGEN_DEBUG_OAR_STEPS=1101:1105 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc -- python /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /tmp/pmc.log 2>&1
f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1)
echo "== $set"; [ -n "$f" ] && python - "$f" <<'PY' || tail -3 /tmp/pmc.log
import csv,sys,collections
acc=collections.defaultdict(float); n=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if 'oar_engine_kernel' in r['Kernel_Name']: acc[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
print({k: round(v/n[k],1) for k,v in acc.items()}, "launches", dict(n))
PY
done
