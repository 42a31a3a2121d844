for cfg in "--steps 30 --warmup 5" "--steps 100 --warmup 5" "--steps 20 --warmup 5"; do
  for np in 0 1; do
    SSAK_BENCH_NO_PROF=$np python bench.py $cfg --no-secondary --no-cpu-baseline --long-steps 50 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$cfg noprof=$np', d['value'], d['ms_per_step'], 'long', d['long_run']['value'], d['long_run']['ms_per_step'], 'kept', d['config']['kept_layers_per_step'])"
  done
done
