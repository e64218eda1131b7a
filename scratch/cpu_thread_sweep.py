"""The cpu_baseline leg of bench.py (the oracle on the host's cores, BASELINE configs[0]: 4 images x 5 captions, fwd + bwd + clip + Adam) at
4 / 8 / 16 / 32 / 64 / 128 / 256 torch threads: the sweep behind bench.py's default thread count.  Each setting runs in its own process (torch's
intra-op pool is sized once)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    import bench
    from sparse_image_captioning_amd.utils.config import ORT_DEFAULTS
    bench.CPU_THREADS = int(sys.argv[1])
    r = bench.cpu_baseline(sys.argv[2], dict(ORT_DEFAULTS), seconds=6.0)
    print(json.dumps({"threads": int(sys.argv[1]), "kind": sys.argv[2], "value": r["value"], "unit": r["unit"], "cores": r["cores"], "host_threads": r["host_threads"]}))
else:
    print(f"host threads: {os.cpu_count()}")
    for kind in ("xe", "decode", "scst"):
        for th in (4, 8, 16, 32, 64, 128, 256):
            if th > (os.cpu_count() or 1): continue
            out = subprocess.run([sys.executable, os.path.abspath(__file__), str(th), kind], capture_output=True, text=True)
            print(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else f"threads {th} {kind}: failed {out.stderr[-200:]}", flush=True)
