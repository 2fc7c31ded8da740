"""A/B of a module-level switch of gpplus_amd.linalg through bench.py in ONE process (dev tool; used for the side-stream
experiment recorded in linalg.ExactMLLFunction).  Edit the attribute name / values below for another switch."""
import sys, json, io, contextlib
sys.path.insert(0, ".")
import gpplus_amd.linalg as L
import bench
for side_min in (6144, 10**9, 6144, 10**9):
    L.SIDE_STREAM_MIN_N = side_min
    sys.argv = ["bench.py", "--steps", "10", "--warmup", "2", "--no-cpu-baseline"]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    b = json.loads(buf.getvalue().strip().splitlines()[-1])
    print("side_min", side_min, "ms/step %.2f" % b["ms_per_step"], {k: round(v, 2) for k, v in b["stages"]["ms"].items()}, flush=True)
