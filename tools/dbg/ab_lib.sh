for i in 1 2; do
for which in base new; do
  if [ $which = base ]; then export UMGEN_LIB_PATH=/root/repo/umgen_amd/libumgen_hip_base.so   # a build of the commit to compare with, copied there by hand; else unset UMGEN_LIB_PATH; fi
  UMGEN_DEBUG_TIMING=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>gpurun_out/ab.err >gpurun_out/ab.json
  grep "prologue" gpurun_out/ab.err | tail -1 | sed 's/.*launches): //'
  python -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('$which', round(d['value'],1), round(d['ms_per_step'],1), round(d['roofline']['avg_launch_us'],1), round(d['roofline']['step']['avg_step_us'],1))"
done; done
