cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_ab; mkdir -p $O
for i in 1 2; do for w in 0 1; do
  CTRLORA_GEMM_R06=$w timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ddim > $O/bench_vae_r06_${w}_$i.log 2>> $O/err.log
done; done
for f in $O/bench_vae_*.log; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "ms_per_step", d["ms_per_step"], "vae_encode_ms", d["end_to_end"]["vae_encode"]["ms"], "e2e img/s", d["end_to_end"]["images_per_s_per_gpu"])
PY
done
timeout 900 python bench.py > $O/bench_default_second_box.log 2>> $O/err.log; tail -1 $O/bench_default_second_box.log | cut -c1-200
