cd /tmp && export TMPDIR=/tmp
cat > /tmp/run_attn.py <<'PY'
import sys; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from gpu_util import lib, vp, bf16_bits, check
F,S,H=20,2207,16; E=H*48
rng=np.random.default_rng(0)
qk=bf16_bits(rng.standard_normal((F*S,2*E),dtype=np.float32)); v=bf16_bits(rng.standard_normal((F*S,E),dtype=np.float32))
y=np.zeros((F*S,E),np.uint16)
check(lib().umgen_dbg_attn_spatial(1, vp(qk), vp(v), F, S, H, vp(y)))
PY
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE"; do
rm -rf /tmp/pmc; rocprofv3 --pmc $set --output-format csv -d /tmp/pmc -- python /tmp/run_attn.py > /tmp/pmc.log 2>&1
f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1)
echo "== $set"; python - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    if 'attn_spatial' in r['Kernel_Name']: acc[r['Counter_Name']]+=float(r['Counter_Value'])
print(dict(acc))
PY
done
