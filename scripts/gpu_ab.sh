# A/B of environment switches inside ONE gpurun call (boxes differ by several percent): bench value per setting, twice
# usage: gpu_ab.sh "VAR=a" "VAR=b" ...
for rep in 1 2; do
  for setting in "$@"; do
    v=$(env $setting python bench.py --no-termination --no-cpu-baseline --no-dense-ldlt $BENCH_ARGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print('%.0f QP/s  %.1f ms/step  factor %.3f s  solve %.3f s  sweeps %d  ipm %d fac %d' % (d['value'], d['ms_per_step'], r.get('factor_seconds',0), r.get('solve_seconds',0), c['sweeps'], c['ipm_iterations'], c['kkt_factorisations']))")
    echo "rep $rep  [$setting]  $v"
  done
done
