cd /root/repo; export TMPDIR=/tmp
timeout 100 tools/micro/attn_bench 20 455 967 1479 1991 455
for v in 455 967 1479; do
  rm -rf /tmp/pmc; (cd /tmp && rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --output-format csv -d /tmp/pmc -- /root/repo/tools/micro/attn_bench 20 $v > /tmp/pmc.log 2>&1)
  f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1)
  echo "== variant $v"
  python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(float); n=collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    if 'attn_spatial' in r['Kernel_Name']: acc[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
print({k:(v/n[k]) for k,v in acc.items()})
PY
done
