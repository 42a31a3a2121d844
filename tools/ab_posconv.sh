show='import json,sys
d=json.loads(sys.stdin.read()); r=d["roofline"]
pc=[k for k in r["kernels"] if "posconv" in k["kernel"] or "gemm_dma_kernel<256, 64" in k["kernel"] or "gemm_dma_kernel<128, 64, 4, 1" in k["kernel"]]
print(sys.argv[1], d["value"], d["ms_per_step"], " | ".join("%s %.0f us (%.2f)" % (k["kernel"][:40], k["us_per_step"], k["frac"]) for k in pc))'
for i in 1 2 3; do
  python bench.py --steps 60 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$show" "direct"
  SSAK_BENCH_POSCONV_GEMM=1 python bench.py --steps 60 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$show" "gemm  "
done
