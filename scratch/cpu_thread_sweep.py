"""The cpu_baseline leg of bench.py (the oracle on the host's cores, BASELINE configs[0]: 4 images x 5 captions, fwd + bwd + clip + Adam) at
4 / 8 / 16 / 32 / 64 / 128 / 256 torch threads: the sweep behind bench.py's default thread count.  Each setting runs in its own process (torch's
intra-op pool is sized once).  The whole sweep takes ~40 minutes on the 256-thread host of the GPU box: past 32 threads the oracle's small
CPU kernels only pay for the pool (256 threads: 0.05 captions/s, minutes per step); `--max-threads N` stops earlier."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[1] == "--max-threads":
    os.environ["ORTK_SWEEP_MAX"] = sys.argv[2]; del sys.argv[1:3]
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    import bench
    from sparse_image_captioning_amd.utils.config import ORT_DEFAULTS
    bench.CPU_THREADS = int(sys.argv[1])
    r = bench.cpu_baseline(sys.argv[2], dict(ORT_DEFAULTS), seconds=6.0)
    print(json.dumps({"threads": int(sys.argv[1]), "kind": sys.argv[2], "value": r["value"], "unit": r["unit"], "cores": r["cores"], "host_threads": r["host_threads"]}))
else:
    print(f"host threads: {os.cpu_count()}")
    cap = min(os.cpu_count() or 1, int(os.environ.get("ORTK_SWEEP_MAX", "1000000")))
    for kind in ("xe", "decode", "scst"):
        for th in (4, 8, 16, 32, 64, 128, 256):
            if th > cap: continue
            out = subprocess.run([sys.executable, os.path.abspath(__file__), str(th), kind], capture_output=True, text=True)
            print(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else f"threads {th} {kind}: failed {out.stderr[-200:]}", flush=True)
