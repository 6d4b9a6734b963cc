# Run ON THE GPU BOX: the whole GPU suite, then the default bench line with a short digest (tools/round_check.sh)
cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu -x 2>&1 | tail -8
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_mid.json 2> gpurun_out/r04_bench_mid.err
tail -c 600 gpurun_out/r04_bench_mid.err
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r04_bench_mid.json").read().strip().splitlines()[-1])
print({k: r[k] for k in ("value", "warmup", "ms_per_step", "value_host_buffers", "value_dropin_api", "refine_cca_s")})
print(r["roofline"]["kernel"][:60], r["roofline"]["frac"],
      [(t["kernel"][:40], round(t["frac"], 3)) for t in r["roofline"]["largest_two_symbols"]])
for k, v in r["secondary"].items():
    print(k, {a: v.get(a) for a in ("ms_per_step", "ms", "ms_device_resident", "ms_stateless_call", "error")})
print(r["recall_trained_weights"])
k = r["kernels"]; tot = sum(k.values())
for a, b in sorted(k.items(), key=lambda kv: -kv[1])[:14]:
    print("%-28s %.4f ms  %.1f%%" % (a, b, 100 * b / tot))
PY
