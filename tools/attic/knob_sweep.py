"""Sweep the drivers' environment knobs (gpp_api.hip) with tools/bench_stages.py: kernel-only ms per evaluation.  Dev tool.
usage: python tools/attic/knob_sweep.py  [N ...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sizes = [int(a) for a in sys.argv[1:]] or [4096, 8192, 10000, 15000, 20000]
variants = [
    ("baseline", {}),
    ("BORDER_T128=320", {"GPP_BORDER_T128": "320"}), ("BORDER_T128=160", {"GPP_BORDER_T128": "160"}),
    ("BORDER_T128=1280", {"GPP_BORDER_T128": "1280"}),
    ("BLK_UPD_TILE=64", {"GPP_BLK_UPD_TILE": "64"}),
    ("TILE_T128=128", {"GPP_TILE_T128": "128"}), ("TILE_T128=448", {"GPP_TILE_T128": "448"}),
    ("TILE_T64=96", {"GPP_TILE_T64": "96"}), ("TILE_T64=384", {"GPP_TILE_T64": "384"}),
    ("PANEL_CUS=16", {"GPP_PANEL_CUS": "16"}), ("PANEL_CUS=8", {"GPP_PANEL_CUS": "8"}), ("PANEL_CUS=64", {"GPP_PANEL_CUS": "64"}),
    ("SPLIT_ELEMS=3e7", {"GPP_SPLIT_ELEMS": "30000000"}), ("SPLIT_ELEMS=1e8", {"GPP_SPLIT_ELEMS": "100000000"}),
    ("SPLIT_UPD=0", {"GPP_SPLIT_UPD": "0"}),
    ("NB=1024,512,8192", {"GPP_LOOKAHEAD_NB": "1024,512,8192"}), ("NB=1024,512,4096", {"GPP_LOOKAHEAD_NB": "1024,512,4096"}),
    ("NB=2048,512,10240", {"GPP_LOOKAHEAD_NB": "2048,512,10240"}), ("NB=1024,256,3072", {"GPP_LOOKAHEAD_NB": "1024,256,3072"}),
    ("BORDER_MAX=8192", {"GPP_BORDER_MAX": "8192"}), ("BORDER_MAX=16384", {"GPP_BORDER_MAX": "16384"}),
    ("BORDER_MAX=21000", {"GPP_BORDER_MAX": "21000"}),
    ("LAUUM_TILE=128", {"GPP_LAUUM_TILE": "128"}), ("LAUUM_TILE=64", {"GPP_LAUUM_TILE": "64"}),
    ("BLK_MAX=3072", {"GPP_BLK_MAX": "3072"}), ("BORDER_MIN=2048", {"GPP_BORDER_MIN": "2048"}),
]
print("%-22s" % "variant" + "".join("%10d" % n for n in sizes), flush=True)
for name, env in variants:
    row = "%-22s" % name
    for n in sizes:
        e = dict(os.environ, **env)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_stages.py"), str(n), "8", "3"], env=e,
                           capture_output=True, text=True)
        m = re.search(r"total\s+([0-9.]+) ms", p.stdout)
        row += "%10s" % (m.group(1) if m else "fail")
    print(row, flush=True)
