# forward-level A/B of an environment switch: bash scripts/r5_ab.sh VAR v1 v2 ...   (extra env via EXTRA="A=1 B=2")
export TMPDIR=/tmp
var=$1; shift
mkdir -p gpurun_out/ab
for v in "$@"; do
  env $EXTRA $var="$v" timeout 600 python bench.py --no-cpu-baseline --no-single-pair --no-rccl-selftest --steps ${STEPS:-8} > gpurun_out/ab/${var}_$(echo $v | tr ',. ' '___').log 2>&1
  tail -1 gpurun_out/ab/${var}_$(echo $v | tr ',. ' '___').log | python -c "
import json,sys
o=json.loads(sys.stdin.read()); k=o.get('kernel_ms_per_step',{})
print('$var=$v', 'pairs/s', round(o['value'],1), 'ms/step', round(o['ms_per_step'],2), {a:round(b,2) for a,b in k.items()})"
done
