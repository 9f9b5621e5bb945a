cd /tmp && export TMPDIR=/tmp
cat > /tmp/run_gemm.py <<'PY'
import ctypes as C, sys
sys.path.insert(0, "/root/repo")
from umgen_amd import _lib
lib = _lib.load_library(); ms = C.c_float()
lib.umgen_dbg_gemm_bench(353120, 3072, 768, 0, 3, C.byref(ms)); print(ms.value)
PY
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" "SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum" "TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum"; do
rm -rf /tmp/pmc; rocprofv3 --pmc $set --output-format csv -d /tmp/pmc -- python /tmp/run_gemm.py > /tmp/pmc.log 2>&1
f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1)
echo "== $set"; [ -n "$f" ] && python - "$f" <<'PY' || tail -3 /tmp/pmc.log
import csv,sys,collections
acc=collections.defaultdict(float); n=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm" in r["Kernel_Name"]: acc[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
print({k:(v/n[k]) for k,v in acc.items()}, dict(n))
PY
done
