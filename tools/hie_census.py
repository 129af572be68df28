"""ms per launch of HieCoAtten's streaming / affinity kernels in a config-4 step (bench.py's census) for the library VQF_LIB selects:
    VQF_LIB=variants/libvqf_X.so python tools/hie_census.py"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--model", "hieCoAtten", "--no-cpu-baseline", "--no-secondary",
                "--steps", "8", "--warmup", "3"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
d = json.load(open(os.path.join(root, "gpurun_out", "bench_census.json")))
ks = d["kernels_ms_per_step"]
print(os.environ.get("VQF_LIB", "product build"), "step %.3f ms |" % d["ms_per_step"],
      "  ".join("%s %.1f" % (k, 1e3 * v["ms_per_step"] / v["launches_per_step"]) for k, v in ks.items() if k.startswith("hie_")))
