"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/attic/collect_r05_profiles.sh) into the per-stage traffic records that
bench.py reads: profiles/<ROUND>_{potrf,trtri,lauum}_pmc.json (ROUND from the environment, default r05).  Each record carries the signature of the kernel build it was collected
with (gpp_version()), and bench.py refuses a record whose signature differs from the library it runs.
Bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB — FETCH_SIZE counts 128-byte requests at 64 B on gfx950 (MI355X_MICROARCH.md, HBM) — per
evaluation (the passes run tools/bench_stages.py N 8 1 = two evaluations).
usage: python tools/pmc_to_json.py <out-dir> <fetch P> <write P> <fetch PT> <write PT> <fetch ALL> <write ALL>
       P = build+potrf only, PT = build+potrf+trtri, ALL = whole evaluation (LAUUM is the separately named instantiation)."""
import collections, csv, glob, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
EVALS = 2
ROUND = os.environ.get("ROUND", "r05")


def sums(d, counter, only=None):
    tot = 0.0
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            if only is not None and only not in r["Kernel_Name"]:
                continue
            tot += float(r["Counter_Value"])
    return tot / EVALS  # KB per evaluation


def main():
    out, fP, wP, fPT, wPT, fA, wA = sys.argv[1:8]
    from gpplus_amd import _lib
    sig = _lib.load().gpp_version().decode().split("src ")[-1]
    how = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of tools/bench_stages.py 20000 8 1 "
           "(STAGES_ONLY selects the stages; two evaluations per pass, values per evaluation; GPP_DAG_PHASED=1: the factorisation's ticket list as a sequence of launches of the same kernel); bytes = (2*FETCH_SIZE + WRITE_SIZE) KB "
           "(gfx950 correction); fabric-side counters, Infinity-Cache hits included")
    fp, wp = sums(fP, "FETCH_SIZE"), sums(wP, "WRITE_SIZE")
    fpt, wpt = sums(fPT, "FETCH_SIZE"), sums(wPT, "WRITE_SIZE")
    lau = "gpp_gemm_f64<2, 64, 64, 1, 16, 2"  # (prefix: later template parameters follow)
    fl, wl = sums(fA, "FETCH_SIZE", lau), sums(wA, "WRITE_SIZE", lau)
    recs = {
        "potrf": ("gpp_potrf_ws stage (all launches of one factorisation at N=20000, incl. the covariance build's 1.6 GB)", fp, wp,
                  "build + potrf"),
        "trtri": ("gpp_trtri stage (batched pair merges at N=20000)", fpt - fp, wpt - wp, "(build + potrf + trtri) - (build + potrf)"),
        "lauum": (lau, fl, wl, "the LAUUM launch of the whole-evaluation pass"),
    }
    for name, (kernel, f, w, what) in recs.items():
        rec = {"kernel": kernel, "lib_signature": sig, "fetch_bytes_per_launch": 2 * f * 1024, "write_bytes_per_launch": w * 1024,
               "traffic_bytes_per_launch": (2 * f + w) * 1024, "note": f"{how}; {what}; kernel build {sig}; profiles/{ROUND}_pmc_fetch_write.txt"}
        with open(os.path.join(out, f"{ROUND}_{name}_pmc.json"), "w") as fh:
            json.dump(rec, fh, indent=1)
        print(f"{name:6s}: fetch {2 * f * 1024 / 1e9:8.2f} GB  write {w * 1024 / 1e9:7.2f} GB  total {(2 * f + w) * 1024 / 1e9:8.2f} GB per evaluation"
              f"   (algorithmic: 3.2 GB = read + write one triangle)")


if __name__ == "__main__":
    main()
