#!/bin/bash
# On the GPU box: the per-GPU share of the strong-scaling job on ONE GPU (bench.py --series N), default kernels, and the batch sizes
# around the number of SIMDs of the device (1024): one wave per series -- at 1024 series every SIMD carries one wave, at 1250 226 of them
# carry two and the launch ends when they do, at 2048 all carry two.  -> gpurun_out/r04_small_batch.json
python3 - <<'PY'
import json, subprocess, sys
res = {}
for n in (10000, 5000, 2500, 2048, 1250, 1025, 1024, 512):
    best = None
    for rep in range(2):
        out = subprocess.run([sys.executable, "bench.py", "--series", str(n), "--steps", "20", "--warmup", "3", "--no-cpu-baseline"], capture_output=True, text=True).stdout
        d = json.loads(out.strip().splitlines()[-1])
        if best is None or d["ms_per_step"] < best["ms_per_step"]:
            best = d
    res[str(n)] = {"ms_per_call": best["ms_per_step"], "forward_ms": best["roofline"]["forward_ms"], "backward_ms": best["roofline"]["backward_ms"],
                   "series_steps_per_s": best["value"]}
    print(n, res[str(n)], flush=True)
t1 = res["10000"]["ms_per_call"]
out = {"note": "python bench.py --series N --steps 20 --warmup 3 --no-cpu-baseline on ONE MI355X, best of 2 (tools/sweep_small4.sh): the share of one GPU of the 10 000-series strong-scaling job at 1 / 2 / 4 / 8 GPUs; projected speed-up = t(10 000) / t(10 000 / G).  No 8-GPU node was available to the build.  1024 / 1025 / 2048 / 512: one wave per series on 1024 SIMDs.",
       "runs": res,
       "projected_speedup": {"2": t1 / res["5000"]["ms_per_call"], "4": t1 / res["2500"]["ms_per_call"], "8": t1 / res["1250"]["ms_per_call"]}}
json.dump(out, open("gpurun_out/r04_small_batch.json", "w"), indent=1)
print(json.dumps(out["projected_speedup"]))
PY
